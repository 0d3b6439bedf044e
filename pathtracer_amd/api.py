"""Python driver over the C ABI: used by the tests, the smoke check and bench.py.

`HostAlgebra` wraps the GPU-free affine helpers; `Context` owns one GPU context; `DeviceScene` is an
uploaded scene.  Everything here forwards to libmipt.so; nothing is computed in Python.
"""
import ctypes as C

import numpy as np

from . import capi
from .capi import MiptError, check, f32, fp   # noqa: F401


STAGES = ("generate", "extend", "sdf_march", "shade", "shade_sdf", "accumulate")


class HostAlgebra:
    """rotation / affine_mul / affine_inverse / noise_transform in the reference's operation order."""

    def rotation(self, axis, angle):
        return self.rotation_batch(axis, np.array([angle], np.float32))[0]

    def rotation_batch(self, axis, angles):
        angles = f32(angles)
        out = np.zeros((angles.shape[0], 9), np.float32)
        if angles.shape[0]:
            check(capi.dll().mipt_rotation(axis.encode(), angles.shape[0], fp(angles), fp(out)), "mipt_rotation")
        return out

    def affine_mul(self, a, b):
        a, b = f32(a).reshape(-1, 12), f32(b).reshape(-1, 12)
        out = np.zeros_like(a)
        check(capi.dll().mipt_affine_mul(a.shape[0], fp(a), fp(b), fp(out)), "mipt_affine_mul")
        return out[0] if out.shape[0] == 1 else out

    def affine_inverse(self, fwd):
        try:
            return self.affine_inverse_batch(f32(fwd).reshape(1, 12))[0]
        except MiptError as e:
            if e.status == -4:
                return None
            raise

    def affine_inverse_batch(self, fwd):
        fwd = f32(fwd).reshape(-1, 12)
        out = np.zeros_like(fwd)
        if fwd.shape[0]:
            check(capi.dll().mipt_affine_inverse(fwd.shape[0], fp(fwd), fp(out)), "mipt_affine_inverse")
        return out

    def noise_transform(self, t3, r3):
        t3, r3 = f32(t3), f32(r3)
        out = np.zeros(12, np.float32)
        check(capi.dll().mipt_noise_transform(fp(t3), fp(r3), fp(out)), "mipt_noise_transform")
        return out

    def world_bounds(self, kind, fwd, params):
        fwd, params = f32(fwd), f32(params)
        out = np.zeros(6, np.float32)
        check(capi.dll().mipt_world_bounds(int(kind), fp(fwd), fp(params), fp(out)), "mipt_world_bounds")
        return out


UPLOAD_DEVICE_BVH = 1


def host_callbacks(host_hit, host_deflect):
    """mipt_host_callbacks around two Python functions: host_hit(prim, pos3, dir3) -> hit position or None,
    host_deflect(prim, pos3, dir3) -> 9 floats (emitted, attenuation, world direction)."""
    def _hit(_user, prim, pos, dirs, out):
        r = host_hit(int(prim), np.ctypeslib.as_array(pos, (3,)).copy(), np.ctypeslib.as_array(dirs, (3,)).copy())
        if r is None:
            return 0
        out[0], out[1], out[2] = float(r[0]), float(r[1]), float(r[2])
        return 1

    def _deflect(_user, prim, pos, dirs, out):
        r = host_deflect(int(prim), np.ctypeslib.as_array(pos, (3,)).copy(), np.ctypeslib.as_array(dirs, (3,)).copy())
        for k in range(9):
            out[k] = float(r[k])

    return capi.HostCallbacks(capi.HIT_FN(_hit), capi.DEFLECT_FN(_deflect), None)


class DeviceScene:
    def __init__(self, ctx, flat, flags=None, host_hit=None, host_deflect=None):
        """host_hit(prim, pos3, dir3) -> hit position (3 floats) or None, host_deflect(prim, pos3, dir3) -> 9 floats
        (emitted, attenuation, world direction): the callbacks of the scene's MIPT_HOST_OBJECT primitives (kind 8)."""
        self.ctx = ctx
        self._keep = [np.ascontiguousarray(flat.kind, np.uint32), f32(flat.fwd), f32(flat.inv), f32(flat.params),
                      np.ascontiguousarray(flat.material, np.uint32), f32(flat.materials), f32(flat.noise)]
        k, fwd, inv, par, mat, mats, noise = self._keep
        d = capi.SceneDesc(k.shape[0], k.ctypes.data_as(capi._u32p), fp(fwd), fp(inv), fp(par),
                           mat.ctypes.data_as(capi._u32p), mats.shape[0], fp(mats), fp(noise))
        h = C.c_void_p()
        if host_hit is not None:
            self._cb = host_callbacks(host_hit, host_deflect)   # kept alive with the scene
            check(capi.dll().mipt_scene_upload_host(ctx.h, C.byref(d), int(flags or 0), C.byref(self._cb), C.byref(h)), "mipt_scene_upload_host")
        elif flags is None:
            check(capi.dll().mipt_scene_upload(ctx.h, C.byref(d), C.byref(h)), "mipt_scene_upload")
        else:
            check(capi.dll().mipt_scene_upload_ex(ctx.h, C.byref(d), int(flags), C.byref(h)), "mipt_scene_upload_ex")
        self.h = h

    def info(self):
        out = np.zeros(4, np.uint32)
        check(capi.dll().mipt_scene_info(self.h, out.ctypes.data_as(capi._u32p)), "mipt_scene_info")
        return dict(nodes=int(out[0]), depth=int(out[1]), analytic=int(out[2]), sdf=int(out[3]))

    def free(self):
        if self.h:
            capi.dll().mipt_scene_free(self.h)
            self.h = None

    def __del__(self):
        try:
            self.free()
        except Exception:   # noqa: BLE001
            pass


class Context:
    def __init__(self, device=-1, stream=None):
        h = C.c_void_p()
        check(capi.dll().mipt_init(device, C.c_void_p(stream) if stream else None, C.byref(h)), "mipt_init")
        self.h = h

    def close(self):
        if self.h:
            capi.dll().mipt_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass

    def upload(self, flat, device_bvh=None, host_hit=None, host_deflect=None):
        """device_bvh: None = the library's default (host build unless MIPT_DEVICE_BVH=1), True/False = explicit."""
        return DeviceScene(self, flat, None if device_bvh is None else (UPLOAD_DEVICE_BVH if device_bvh else 0), host_hit, host_deflect)

    def set_profiling(self, on):
        check(capi.dll().mipt_set_profiling(self.h, 1 if on else 0), "mipt_set_profiling")

    def set_counting(self, on):
        """Counting build of the closest-hit kernels: stats gain node_visits and prim_tests."""
        check(capi.dll().mipt_set_counting(self.h, 1 if on else 0), "mipt_set_counting")

    def trim(self):
        """Frees the working set the context keeps on the GPU between calls."""
        check(capi.dll().mipt_trim(self.h), "mipt_trim")

    def cancel(self):
        check(capi.dll().mipt_cancel(self.h), "mipt_cancel")

    @staticmethod
    def _settings(w, h, bounces, spp, seed, first_sample, flags):
        return capi.Settings(w, h, bounces, spp, seed, first_sample, flags)

    @staticmethod
    def _stats(st):
        return dict(paths=int(st.paths), rays=int(st.rays), sdf_rays=int(st.sdf_rays), sdf_shades=int(st.sdf_shades),
                    sdf_far_evals=int(st.sdf_far_evals), sdf_near_evals=int(st.sdf_near_evals),
                    sdf_noise_calls=int(st.sdf_noise_calls), shade_evals=int(st.shade_evals),
                    shade_noise_calls=int(st.shade_noise_calls), ms_device=float(st.ms_device),
                    stage_ms=dict(zip(STAGES, (float(x) for x in st.stage_ms))),
                    stage_launches=dict(zip(STAGES, (int(x) for x in st.stage_launches))),
                    primary_traced=int(st.primary_traced), samples_done=int(st.samples_done),
                    node_visits=int(st.node_visits), prim_tests=int(st.prim_tests),
                    host_rays=int(st.host_rays), host_shades=int(st.host_shades),
                    trav_rounds=int(st.trav_rounds), trav_leaf_steps=int(st.trav_leaf_steps), trav_restarts=int(st.trav_restarts),
                    marches_carried=int(st.marches_carried))

    def render(self, scene, cam16, w, h, bounces, spp, seed=1, first_sample=0, flags=0):
        """Renderer::render: (h, w, 3) float32 mean image and the call's statistics."""
        cam = capi.make_camera(cam16)
        st = self._settings(w, h, bounces, spp, seed, first_sample, flags)
        out = np.zeros((h, w, 3), np.float32)
        stats = capi.Stats()
        check(capi.dll().mipt_render(self.h, scene.h, C.byref(cam), C.byref(st), fp(out), C.byref(stats)), "mipt_render")
        return out, self._stats(stats)

    def render_accumulate(self, scene, cam16, w, h, bounces, spp, device_ptr, seed=1, first_sample=0, flags=0):
        """Adds this call's samples into the device buffer at `device_ptr` (w*h*3 floats)."""
        cam = capi.make_camera(cam16)
        st = self._settings(w, h, bounces, spp, seed, first_sample, flags)
        stats = capi.Stats()
        check(capi.dll().mipt_render_accumulate(self.h, scene.h, C.byref(cam), C.byref(st), C.c_void_p(device_ptr),
                                                C.byref(stats)), "mipt_render_accumulate")
        return self._stats(stats)

    def scale_device(self, in_ptr, k, n_floats, out_ptr):
        check(capi.dll().mipt_scale_device(self.h, C.c_void_p(in_ptr), C.c_float(k), n_floats, C.c_void_p(out_ptr)),
              "mipt_scale_device")

    def primary_hits(self, scene, cam16, w, h):
        cam = capi.make_camera(cam16)
        prim = np.zeros((h, w), np.int32)
        t = np.zeros((h, w), np.float32)
        check(capi.dll().mipt_primary_hits(self.h, scene.h, C.byref(cam), w, h, prim.ctypes.data_as(capi._i32p), fp(t)),
              "mipt_primary_hits")
        return prim, t

    def trace_rays(self, scene, pos, dirs, bounces, seed=1, first_hit=False):
        pos, dirs = f32(pos), f32(dirs)
        n = pos.shape[0]
        out = np.zeros((n, 3), np.float32)
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        check(capi.dll().mipt_trace_rays(self.h, scene.h, n, fp(pos), fp(dirs), bounces, seed, fp(out),
                                         prim.ctypes.data_as(capi._i32p) if first_hit else None,
                                         fp(t) if first_hit else None), "mipt_trace_rays")
        return (out, prim, t) if first_hit else out

    def sdf_eval(self, scene, prim, pos):
        """Signed distance of SDF primitive `prim` at (n, 3) points of its local frame."""
        pos = f32(pos)
        out = np.zeros(pos.shape[0], np.float32)
        check(capi.dll().mipt_sdf_eval(self.h, scene.h, int(prim), pos.shape[0], fp(pos), fp(out)), "mipt_sdf_eval")
        return out

    def sdf_material_eval(self, scene, prim, pos):
        """The material hits at (n, 3) local points of SDF primitive `prim` are shaded with: (n, 9) floats
        {albedo rgb, emitted rgb, diffuse, specular, sharpness}."""
        pos = f32(pos)
        out = np.zeros((pos.shape[0], 9), np.float32)
        check(capi.dll().mipt_sdf_material_eval(self.h, scene.h, int(prim), pos.shape[0], fp(pos), fp(out)), "mipt_sdf_material_eval")
        return out

    def tonemap(self, kind, img):
        img = f32(img)
        h, w = img.shape[:2]
        out = np.zeros_like(img)
        check(capi.dll().mipt_tonemap(self.h, kind, fp(img), fp(out), w, h), "mipt_tonemap")
        return out

    def present_rgba8(self, sum_ptr, scale, tonemap, w, h):
        """The demo's display step for a device accumulator: (h, w, 4) uint8."""
        out = np.zeros((h, w, 4), np.uint8)
        check(capi.dll().mipt_present_rgba8(self.h, C.c_void_p(sum_ptr), C.c_float(scale), int(tonemap),
                                            out.ctypes.data_as(C.POINTER(C.c_uint8)), w, h), "mipt_present_rgba8")
        return out

    def tonemap_device(self, kind, in_ptr, out_ptr, w, h):
        check(capi.dll().mipt_tonemap_device(self.h, kind, C.c_void_p(in_ptr), C.c_void_p(out_ptr), w, h),
              "mipt_tonemap_device")


NODE_SIMULATE = 1


class Node:
    """Several GPUs driven by one process (mipt_init_all): samples sharded over the devices, one RCCL all-reduce per frame.
    devices=None: the first n devices (n = 0: all).  simulate=True: no RCCL, entries of `devices` may repeat (k shards on one GPU)."""

    def __init__(self, n=0, devices=None, simulate=False):
        h = C.c_void_p()
        arr = None
        if devices is not None:
            arr = (C.c_int * len(devices))(*devices)
            n = len(devices)
        check(capi.dll().mipt_init_all(n, arr, NODE_SIMULATE if simulate else 0, C.byref(h)), "mipt_init_all")
        self.h = h
        self._scenes = []
        self._keep = []

    def size(self):
        return int(capi.dll().mipt_node_size(self.h))

    def upload(self, flat, host_hit=None, host_deflect=None):
        keep = [np.ascontiguousarray(flat.kind, np.uint32), f32(flat.fwd), f32(flat.inv), f32(flat.params),
                np.ascontiguousarray(flat.material, np.uint32), f32(flat.materials), f32(flat.noise)]
        k, fwd, inv, par, mat, mats, noise = keep
        d = capi.SceneDesc(k.shape[0], k.ctypes.data_as(capi._u32p), fp(fwd), fp(inv), fp(par),
                           mat.ctypes.data_as(capi._u32p), mats.shape[0], fp(mats), fp(noise))
        s = C.c_void_p()
        if host_hit is not None:
            cb = host_callbacks(host_hit, host_deflect)
            self._keep.append(cb)   # the callbacks live as long as the node
            check(capi.dll().mipt_node_scene_upload_host(self.h, C.byref(d), C.byref(cb), C.byref(s)), "mipt_node_scene_upload_host")
        else:
            check(capi.dll().mipt_node_scene_upload(self.h, C.byref(d), C.byref(s)), "mipt_node_scene_upload")
        self._scenes.append(s)
        return s

    def ctx_handle(self, shard):
        return capi.dll().mipt_node_ctx(self.h, int(shard))

    def render(self, scene, cam16, w, h, bounces, spp, seed=1, tonemap=-1, first_sample=0):
        cam = capi.make_camera(cam16)
        st = capi.Settings(w, h, bounces, spp, seed, first_sample, 0)
        out = np.zeros((h, w, 3), np.float32)
        stats = capi.Stats()
        check(capi.dll().mipt_node_render(self.h, scene, C.byref(cam), C.byref(st), int(tonemap), fp(out), C.byref(stats)), "mipt_node_render")
        return out, Context._stats(stats)

    def close(self):
        if self.h:
            for s in self._scenes:
                capi.dll().mipt_node_scene_free(self.h, s)
            self._scenes = []
            capi.dll().mipt_node_shutdown(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:   # noqa: BLE001
            pass
