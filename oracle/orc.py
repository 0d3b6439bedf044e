"""TEST INFRASTRUCTURE -- ctypes bindings for liboracle.so (the CPU restatement) and, where it has
been built, oracle/_ref/libref.so (the reference's own sources behind the same calls).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
Scenes are passed as the flat arrays described in refcpu.hpp; any object with the attributes
kind, fwd, inv, params, material, materials, noise (numpy arrays) works.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)
_u8p = C.POINTER(C.c_uint8)
_u64p = C.POINTER(C.c_uint64)

RNG_MINSTD, RNG_COUNTER = 0, 1
SPHERE, BOX, TRIANGLE, CRATERBALL = 0, 1, 2, 3


def build(ref=True):
    """Compile liboracle.so and, when the reference checkout is present, _ref/libref.so."""
    subprocess.run(["make", "-C", _HERE, "liboracle.so", "liboracle_fast.so"], check=True, stdout=subprocess.DEVNULL)
    if ref and os.path.isdir("/root/reference/src"):
        subprocess.run(["make", "-C", _HERE, "ref"], check=True, stdout=subprocess.DEVNULL)


def _f(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _fp(a):
    return a.ctypes.data_as(_f32p)


def _load(path):
    if not os.path.exists(path):
        return None
    return C.CDLL(path)


_lib = None
_ref = None


def lib():
    global _lib
    if _lib is None:
        _lib = _load(os.path.join(_HERE, "liboracle.so"))
        if _lib is None:
            raise RuntimeError("oracle/liboracle.so missing: run `make -C oracle`")
        _lib.orc_scene_create.restype = C.c_void_p
        _lib.orc_counter_uniform01.restype = C.c_float
    return _lib


_fast = None


def fast_lib():
    """liboracle_fast.so (the restatement compiled with -ffast-math: a timing baseline, not an oracle), or None."""
    global _fast
    if _fast is None:
        _fast = _load(os.path.join(_HERE, "liboracle_fast.so"))
        if _fast is not None:
            _fast.orc_scene_create.restype = C.c_void_p
            _fast.orc_counter_uniform01.restype = C.c_float
    return _fast


_v3 = None


def v3_lib():
    """liboracle_v3.so (the restatement compiled for x86-64-v3, same results as liboracle.so: the timed CPU baseline),
    or None where it is missing or this host's cores lack AVX2."""
    global _v3
    if _v3 is None:
        try:
            flags = open("/proc/cpuinfo").read()
        except OSError:
            flags = ""
        if " avx2" in flags and " bmi2" in flags and " fma" in flags:
            _v3 = _load(os.path.join(_HERE, "liboracle_v3.so"))
            if _v3 is not None:
                _v3.orc_scene_create.restype = C.c_void_p
                _v3.orc_counter_uniform01.restype = C.c_float
    return _v3


def ref():
    """The compiled reference, or None where it was not built."""
    global _ref
    if _ref is None:
        _ref = _load(os.path.join(_HERE, "_ref", "libref.so"))
        if _ref is not None:
            _ref.ref_scene_create.restype = C.c_void_p
    return _ref


# ---------------------------------------------------------------------------------------------
# helpers shared by both libraries
# ---------------------------------------------------------------------------------------------
class _Api:
    def __init__(self, dll, prefix):
        self.dll = dll
        self.p = prefix

    def fn(self, name):
        return getattr(self.dll, self.p + name)

    # -- linear algebra -------------------------------------------------------------------------
    def rotation(self, axis, angle):
        out = np.zeros(9, np.float32)
        self.fn("rotation_" + axis)(C.c_float(angle), _fp(out))
        return out

    def affine_mul(self, a, b):
        a, b = _f(a), _f(b)
        out = np.zeros(12, np.float32)
        self.fn("affine_mul")(_fp(a), _fp(b), _fp(out))
        return out

    def affine_inverse(self, fwd):
        fwd = _f(fwd)
        out = np.zeros(12, np.float32)
        ok = self.fn("affine_inverse")(_fp(fwd), _fp(out))
        return out if ok else None

    def rotation_batch(self, axis, angles):
        return np.stack([self.rotation(axis, float(a)) for a in angles]) if len(angles) else np.zeros((0, 9), np.float32)

    def affine_inverse_batch(self, fwd):
        rows = [self.affine_inverse(f) for f in fwd]
        if any(r is None for r in rows):
            raise ValueError("transformation is not invertible by the reference's rule (det < 1e-6)")
        return np.stack(rows) if rows else np.zeros((0, 12), np.float32)

    def noise_transform(self, t3, r3):
        t3, r3 = _f(t3), _f(r3)
        out = np.zeros(12, np.float32)
        self.fn("noise_transform")(_fp(t3), _fp(r3), _fp(out))
        return out

    def world_bbox(self, kind, fwd, params):
        fwd, params = _f(fwd), _f(params)
        out = np.zeros(6, np.float32)
        self.fn("world_bbox")(C.c_uint32(kind), _fp(fwd), _fp(params), _fp(out))
        return out


class Oracle(_Api):
    """liboracle.so; with fast_math=True the -ffast-math build (for timing only)."""

    def __init__(self, fast_math=False, tuned=False):
        dll = fast_lib() if fast_math else (v3_lib() if tuned else lib())
        if dll is None:
            raise RuntimeError("oracle/liboracle_fast.so or liboracle_v3.so missing (or no AVX2 here): run `make -C oracle`")
        super().__init__(dll, "orc_")

    def scene(self, s):
        return OracleScene(self, s)

    def hit_local(self, kind, params, noise, pos, dirs):
        params, noise, pos, dirs = _f(params), _f(noise), _f(pos), _f(dirs)
        n = pos.shape[0]
        hit = np.zeros(n, np.uint8)
        out = np.zeros((n, 3), np.float32)
        self.dll.orc_hit_local(C.c_uint32(kind), _fp(params), _fp(noise), C.c_uint32(n), _fp(pos), _fp(dirs),
                               hit.ctypes.data_as(_u8p), _fp(out))
        return hit.astype(bool), out

    def normal_local(self, kind, params, noise, pos):
        params, noise, pos = _f(params), _f(noise), _f(pos)
        n = pos.shape[0]
        out = np.zeros((n, 3), np.float32)
        self.dll.orc_normal_local(C.c_uint32(kind), _fp(params), _fp(noise), C.c_uint32(n), _fp(pos), _fp(out))
        return out

    def sdf_craterball(self, offset, noise, pos):
        offset, noise, pos = _f(offset), _f(noise), _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        self.dll.orc_sdf_craterball(_fp(offset), _fp(noise), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out))
        return out

    def sdf_value(self, kind, params, noise, pos):
        """signedDistance of the demo's RoughBoxObject (5), RoughSphereObject (6), WeirdWallObject (7) at local points."""
        params, noise, pos = _f(params), _f(noise), _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        if not self.dll.orc_sdf_value(C.c_uint32(kind), _fp(params), _fp(noise), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out)):
            raise ValueError("no signed distance for kind %d" % kind)
        return out

    def noise(self, noise, pos):
        noise, pos = _f(noise), _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        self.dll.orc_noise(_fp(noise), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out))
        return out

    def camera_rays(self, cam, sx, sy, rng_mode=RNG_MINSTD, seed=1):
        cam, sx, sy = _f(cam), _f(sx), _f(sy)
        n = sx.shape[0]
        d = np.zeros((n, 3), np.float32)
        p = np.zeros((n, 3), np.float32)
        self.dll.orc_camera_rays(_fp(cam), C.c_uint32(n), _fp(sx), _fp(sy), C.c_uint32(rng_mode), C.c_uint64(seed),
                                 _fp(d), _fp(p))
        return d, p

    def deflect(self, mat14, inbound, normal, rng_mode=RNG_MINSTD, seed=1):
        mat14, inbound, normal = _f(mat14), _f(inbound), _f(normal)
        n = inbound.shape[0]
        out = np.zeros((n, 12), np.float32)
        self.dll.orc_deflect(_fp(mat14), C.c_uint32(n), _fp(inbound), _fp(normal), C.c_uint32(rng_mode),
                             C.c_uint64(seed), _fp(out))
        return out

    def tonemap(self, kind, img):
        img = _f(img)
        h, w = img.shape[:2]
        out = np.zeros_like(img)
        self.dll.orc_tonemap(C.c_uint32(kind), C.c_uint32(w), C.c_uint32(h), _fp(img), _fp(out))
        return out

    def image_bytes(self, img):
        img = _f(img)
        h, w = img.shape[:2]
        out = np.zeros(16 + 12 * w * h, np.uint8)
        self.dll.orc_image_bytes(C.c_uint32(w), C.c_uint32(h), _fp(img), out.ctypes.data_as(_u8p))
        return out.tobytes()

    def uniform01(self, seed, pixel, sample, stage, k):
        return float(self.dll.orc_counter_uniform01(C.c_uint64(seed), C.c_uint32(pixel), C.c_uint32(sample),
                                                    C.c_uint32(stage), C.c_uint32(k)))


class OracleScene:
    def __init__(self, api, s):
        self.api = api
        self.n = int(s.kind.shape[0])
        self._keep = [np.ascontiguousarray(s.kind, np.uint32), _f(s.fwd), _f(s.inv), _f(s.params),
                      np.ascontiguousarray(s.material, np.uint32), _f(s.materials), _f(s.noise)]
        k, fwd, inv, par, mat, mats, noise = self._keep
        self.h = C.c_void_p(api.dll.orc_scene_create(
            C.c_uint32(self.n), k.ctypes.data_as(_u32p), _fp(fwd), _fp(inv), _fp(par), mat.ctypes.data_as(_u32p),
            C.c_uint32(mats.shape[0]), _fp(mats), _fp(noise)))

    def __del__(self):
        if getattr(self, "h", None):
            self.api.dll.orc_scene_free(self.h)
            self.h = None

    def use_reference_tree_order(self):
        order = np.zeros(self.n, np.uint32)
        rc = self.api.dll.orc_scene_use_reference_tree_order(self.h, order.ctypes.data_as(_u32p))
        if rc != 0:
            raise RuntimeError("reference tree builder finds no split for this scene")
        return order

    def closest_hit(self, pos, dirs):
        pos, dirs = _f(pos), _f(dirs)
        n = pos.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        hp = np.zeros((n, 3), np.float32)
        self.api.dll.orc_closest_hit(self.h, C.c_uint32(n), _fp(pos), _fp(dirs), prim.ctypes.data_as(_i32p), _fp(t), _fp(hp))
        return prim, t, hp

    def trace(self, pos, dirs, depth, rng_mode=RNG_MINSTD, seed=1):
        pos, dirs = _f(pos), _f(dirs)
        n = pos.shape[0]
        out = np.zeros((n, 3), np.float32)
        self.api.dll.orc_trace(self.h, C.c_uint32(n), _fp(pos), _fp(dirs), C.c_uint32(depth), C.c_uint32(rng_mode),
                               C.c_uint64(seed), _fp(out))
        return out

    def primary_hits(self, cam, w, h):
        cam = _f(cam)
        prim = np.zeros((h, w), np.int32)
        t = np.zeros((h, w), np.float32)
        self.api.dll.orc_primary_hits(self.h, _fp(cam), C.c_uint32(w), C.c_uint32(h), prim.ctypes.data_as(_i32p), _fp(t))
        return prim, t

    def render(self, cam, w, h, bounces, spp, rng_mode=RNG_COUNTER, seed=1, first_sample=0, threads=0,
               full_width=True):
        cam = _f(cam)
        out = np.zeros((h, w, 3), np.float32)
        stats = np.zeros(4, np.uint64)
        rc = self.api.dll.orc_render(self.h, _fp(cam), C.c_uint32(w), C.c_uint32(h), C.c_uint32(bounces),
                                     C.c_uint32(spp), C.c_uint32(rng_mode), C.c_uint64(seed), C.c_uint32(first_sample),
                                     C.c_uint32(threads), C.c_uint32(1 if full_width else 0), _fp(out),
                                     stats.ctypes.data_as(_u64p))
        if rc != 0:
            raise ValueError("orc_render rejected the settings")
        return out, dict(paths=int(stats[0]), rays=int(stats[1]), sdf_evals=int(stats[2]), sdf_near_evals=int(stats[3]))


class Reference(_Api):
    """oracle/_ref/libref.so -- the reference's own code -- or, given a path, another build of the same
    harness (oracle/ref_harness.cpp compiled against this repository's host API headers)."""

    def __init__(self, path=None):
        if path is None:
            dll = ref()
            if dll is None:
                raise RuntimeError("oracle/_ref/libref.so has not been built (needs the reference checkout)")
        else:
            dll = C.CDLL(path)
            dll.ref_scene_create.restype = C.c_void_p
        super().__init__(dll, "ref_")

    def has(self, name):
        return hasattr(self.dll, name)

    def set_noise(self, noise12):
        """Harness builds without the demo's source block take the noise transform as data."""
        if self.has("ref_set_noise"):
            n = _f(noise12)
            self.dll.ref_set_noise(_fp(n))

    def init_noise(self, seed):
        out = np.zeros(12, np.float32)
        self.dll.ref_init_noise(C.c_uint64(seed), _fp(out))
        return out

    def scene(self, s, build_tree=True):
        return ReferenceScene(self, s, build_tree)

    def hit_local(self, kind, params, pos, dirs):
        params, pos, dirs = _f(params), _f(pos), _f(dirs)
        n = pos.shape[0]
        hit = np.zeros(n, np.uint8)
        out = np.zeros((n, 3), np.float32)
        self.dll.ref_hit_local(C.c_uint32(kind), _fp(params), C.c_uint32(n), _fp(pos), _fp(dirs),
                               hit.ctypes.data_as(_u8p), _fp(out))
        return hit.astype(bool), out

    def normal_local(self, kind, params, pos):
        params, pos = _f(params), _f(pos)
        out = np.zeros((pos.shape[0], 3), np.float32)
        self.dll.ref_normal_local(C.c_uint32(kind), _fp(params), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out))
        return out

    def sdf_craterball(self, offset, pos):
        offset, pos = _f(offset), _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        self.dll.ref_sdf_craterball(_fp(offset), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out))
        return out

    def sdf_value(self, kind, params, pos):
        params, pos = _f(params), _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        if not self.has("ref_sdf_value") or not self.dll.ref_sdf_value(C.c_uint32(kind), _fp(params), C.c_uint32(pos.shape[0]), _fp(pos), _fp(out)):
            return None
        return out

    def noise(self, pos):
        pos = _f(pos)
        out = np.zeros(pos.shape[0], np.float32)
        self.dll.ref_noise(C.c_uint32(pos.shape[0]), _fp(pos), _fp(out))
        return out

    def camera_rays(self, cam, sx, sy, seed=1):
        cam, sx, sy = _f(cam), _f(sx), _f(sy)
        n = sx.shape[0]
        d = np.zeros((n, 3), np.float32)
        p = np.zeros((n, 3), np.float32)
        self.dll.ref_camera_rays(_fp(cam), C.c_uint32(n), _fp(sx), _fp(sy), C.c_uint64(seed), _fp(d), _fp(p))
        return d, p

    def deflect(self, mat14, inbound, normal, seed=1):
        mat14, inbound, normal = _f(mat14), _f(inbound), _f(normal)
        n = inbound.shape[0]
        out = np.zeros((n, 12), np.float32)
        self.dll.ref_deflect(_fp(mat14), C.c_uint32(n), _fp(inbound), _fp(normal), C.c_uint64(seed), _fp(out))
        return out

    def tonemap(self, kind, img):
        img = _f(img)
        h, w = img.shape[:2]
        out = np.zeros_like(img)
        self.dll.ref_tonemap(C.c_uint32(kind), C.c_uint32(w), C.c_uint32(h), _fp(img), _fp(out))
        return out

    def image_bytes(self, img):
        img = _f(img)
        h, w = img.shape[:2]
        out = np.zeros(16 + 12 * w * h, np.uint8)
        rc = self.dll.ref_image_bytes(C.c_uint32(w), C.c_uint32(h), _fp(img), out.ctypes.data_as(_u8p))
        assert rc == 0
        return out.tobytes()


class ReferenceScene:
    def __init__(self, api, s, build_tree):
        self.api = api
        self.n = int(s.kind.shape[0])
        k = np.ascontiguousarray(s.kind, np.uint32)
        fwd, inv, par = _f(s.fwd), _f(s.inv), _f(s.params)
        mat = np.ascontiguousarray(s.material, np.uint32)
        mats = _f(s.materials)
        self.h = C.c_void_p(api.dll.ref_scene_create(
            C.c_uint32(self.n), k.ctypes.data_as(_u32p), _fp(fwd), _fp(inv), _fp(par), mat.ctypes.data_as(_u32p),
            C.c_uint32(mats.shape[0]), _fp(mats)))
        if build_tree:
            api.dll.ref_scene_build_tree(self.h)

    def __del__(self):
        if getattr(self, "h", None):
            self.api.dll.ref_scene_free(self.h)
            self.h = None

    def inverse(self, i):
        out = np.zeros(12, np.float32)
        self.api.dll.ref_scene_inverse(self.h, C.c_uint32(i), _fp(out))
        return out

    def closest_hit(self, pos, dirs):
        pos, dirs = _f(pos), _f(dirs)
        n = pos.shape[0]
        prim = np.zeros(n, np.int32)
        t = np.zeros(n, np.float32)
        hp = np.zeros((n, 3), np.float32)
        self.api.dll.ref_closest_hit(self.h, C.c_uint32(n), _fp(pos), _fp(dirs), prim.ctypes.data_as(_i32p), _fp(t), _fp(hp))
        return prim, t, hp

    def trace(self, pos, dirs, depth, seed=1):
        pos, dirs = _f(pos), _f(dirs)
        out = np.zeros((pos.shape[0], 3), np.float32)
        self.api.dll.ref_trace(self.h, C.c_uint32(pos.shape[0]), _fp(pos), _fp(dirs), C.c_uint32(depth),
                               C.c_uint64(seed), _fp(out))
        return out

    def primary_hits(self, cam, w, h):
        cam = _f(cam)
        prim = np.zeros((h, w), np.int32)
        t = np.zeros((h, w), np.float32)
        self.api.dll.ref_primary_hits(self.h, _fp(cam), C.c_uint32(w), C.c_uint32(h), prim.ctypes.data_as(_i32p), _fp(t))
        return prim, t

    def render(self, cam, w, h, bounces, spp, seed=1, threads=1):
        cam = _f(cam)
        out = np.zeros((h, w, 3), np.float32)
        self.api.dll.ref_render(self.h, _fp(cam), C.c_uint32(w), C.c_uint32(h), C.c_uint32(bounces), C.c_uint32(spp),
                                C.c_uint64(seed), C.c_uint32(threads), _fp(out))
        return out
