"""ctypes binding of libmipt.so (include/mipt.h): the only way Python code reaches the HIP path.

There is no fallback: if the shared library is missing or a call fails, an exception is raised.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libmipt.so")

_f32p = C.POINTER(C.c_float)
_u32p = C.POINTER(C.c_uint32)
_i32p = C.POINTER(C.c_int32)

OK = 0
ERR_CANCELLED = -5
FLAG_NO_JITTER = 1


class MiptError(RuntimeError):
    def __init__(self, status, what):
        self.status = status
        super().__init__(f"{what}: {_strerror(status)} (status {status})")


class SceneDesc(C.Structure):
    _fields_ = [("n_prims", C.c_uint32), ("kind", _u32p), ("fwd", _f32p), ("inv", _f32p), ("params", _f32p),
                ("material", _u32p), ("n_materials", C.c_uint32), ("materials", _f32p), ("noise_transform", _f32p)]


class Camera(C.Structure):
    _fields_ = [("transform", C.c_float * 12), ("aspect", C.c_float), ("fov_degrees", C.c_float),
                ("focal_distance", C.c_float), ("blur_radius", C.c_float)]


class Settings(C.Structure):
    _fields_ = [("width", C.c_uint32), ("height", C.c_uint32), ("bounces", C.c_uint32), ("spp", C.c_uint32),
                ("seed", C.c_uint64), ("first_sample", C.c_uint32), ("flags", C.c_uint32)]


class Stats(C.Structure):
    _fields_ = [("paths", C.c_uint64), ("rays", C.c_uint64), ("sdf_rays", C.c_uint64), ("sdf_shades", C.c_uint64),
                ("sdf_far_evals", C.c_uint64), ("sdf_near_evals", C.c_uint64), ("sdf_noise_calls", C.c_uint64),
                ("shade_evals", C.c_uint64), ("shade_noise_calls", C.c_uint64),
                ("ms_device", C.c_double), ("stage_ms", C.c_double * 6), ("stage_launches", C.c_uint64 * 6),
                ("primary_traced", C.c_uint64), ("samples_done", C.c_uint64),
                ("node_visits", C.c_uint64), ("prim_tests", C.c_uint64),
                ("host_rays", C.c_uint64), ("host_shades", C.c_uint64),
                ("trav_rounds", C.c_uint64), ("trav_leaf_steps", C.c_uint64), ("trav_restarts", C.c_uint64),
                ("marches_carried", C.c_uint64)]


HIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint32, _f32p, _f32p, _f32p)
DEFLECT_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_uint32, _f32p, _f32p, _f32p)


class HostCallbacks(C.Structure):
    _fields_ = [("hit", HIT_FN), ("deflect", DEFLECT_FN), ("user", C.c_void_p)]


EXPORTS = [
    "mipt_init", "mipt_shutdown", "mipt_cancel", "mipt_last_hip_error", "mipt_set_profiling", "mipt_strerror", "mipt_scene_upload",
    "mipt_scene_upload_ex", "mipt_device_floats_alloc", "mipt_device_floats_zero", "mipt_device_floats_read", "mipt_device_floats_free",
    "mipt_scene_free", "mipt_scene_info", "mipt_render", "mipt_render_accumulate", "mipt_scale_device",
    "mipt_primary_hits", "mipt_trace_rays", "mipt_tonemap", "mipt_tonemap_device", "mipt_affine_inverse",
    "mipt_rotation", "mipt_affine_mul", "mipt_noise_transform", "mipt_world_bounds", "mipt_present_rgba8",
    "mipt_tonemap_device_noalloc", "mipt_trim", "mipt_sdf_eval", "mipt_sdf_material_eval", "mipt_set_counting", "mipt_device_floats_write", "mipt_scene_upload_host", "mipt_add_device", "mipt_init_all", "mipt_node_shutdown", "mipt_node_size", "mipt_node_ctx",
    "mipt_node_cancel", "mipt_node_scene_upload", "mipt_node_scene_upload_host", "mipt_node_scene_free", "mipt_node_render",
]

_dll = None


def dll():
    global _dll
    if _dll is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(f"{LIB_PATH} is missing: build it with `make -C pathtracer_amd/csrc` "
                              "(or __graft_entry__.build()); there is no CPU fallback")
        d = C.CDLL(LIB_PATH)
        d.mipt_strerror.restype = C.c_char_p
        d.mipt_init.argtypes = [C.c_int, C.c_void_p, C.POINTER(C.c_void_p)]
        d.mipt_shutdown.argtypes = [C.c_void_p]
        d.mipt_shutdown.restype = None
        d.mipt_cancel.argtypes = [C.c_void_p]
        d.mipt_last_hip_error.argtypes = [C.c_void_p]
        d.mipt_set_profiling.argtypes = [C.c_void_p, C.c_int]
        d.mipt_trim.argtypes = [C.c_void_p]
        d.mipt_set_counting.argtypes = [C.c_void_p, C.c_int]
        d.mipt_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        d.mipt_scene_upload_ex.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.c_uint32, C.POINTER(C.c_void_p)]
        d.mipt_scene_upload_host.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.c_uint32, C.POINTER(HostCallbacks), C.POINTER(C.c_void_p)]
        d.mipt_add_device.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_size_t]
        d.mipt_init_all.argtypes = [C.c_int, C.POINTER(C.c_int), C.c_uint32, C.POINTER(C.c_void_p)]
        d.mipt_node_shutdown.argtypes = [C.c_void_p]
        d.mipt_node_shutdown.restype = None
        d.mipt_node_size.argtypes = [C.c_void_p]
        d.mipt_node_ctx.argtypes = [C.c_void_p, C.c_int]
        d.mipt_node_ctx.restype = C.c_void_p
        d.mipt_node_cancel.argtypes = [C.c_void_p]
        d.mipt_node_scene_upload.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.POINTER(C.c_void_p)]
        d.mipt_node_scene_upload_host.argtypes = [C.c_void_p, C.POINTER(SceneDesc), C.c_void_p, C.POINTER(C.c_void_p)]
        d.mipt_node_scene_free.argtypes = [C.c_void_p, C.c_void_p]
        d.mipt_node_scene_free.restype = None
        d.mipt_node_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Camera), C.POINTER(Settings), C.c_int, _f32p, C.POINTER(Stats)]
        d.mipt_device_floats_alloc.argtypes = [C.c_void_p, C.c_size_t, C.POINTER(C.c_void_p)]
        d.mipt_device_floats_zero.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t]
        d.mipt_device_floats_read.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_size_t, _f32p]
        d.mipt_device_floats_write.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, _f32p]
        d.mipt_device_floats_free.argtypes = [C.c_void_p, C.c_void_p]
        d.mipt_device_floats_free.restype = None
        d.mipt_scene_free.argtypes = [C.c_void_p]
        d.mipt_scene_free.restype = None
        d.mipt_scene_info.argtypes = [C.c_void_p, _u32p]
        d.mipt_render.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Camera), C.POINTER(Settings), _f32p, C.POINTER(Stats)]
        d.mipt_render_accumulate.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Camera), C.POINTER(Settings), C.c_void_p, C.POINTER(Stats)]
        d.mipt_scale_device.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_size_t, C.c_void_p]
        d.mipt_primary_hits.argtypes = [C.c_void_p, C.c_void_p, C.POINTER(Camera), C.c_uint32, C.c_uint32, _i32p, _f32p]
        d.mipt_trace_rays.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, _f32p, _f32p, C.c_uint32, C.c_uint64, _f32p, _i32p, _f32p]
        d.mipt_sdf_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, _f32p]
        d.mipt_sdf_material_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint32, _f32p, _f32p]
        d.mipt_tonemap.argtypes = [C.c_void_p, C.c_int, _f32p, _f32p, C.c_size_t, C.c_size_t]
        d.mipt_tonemap_device.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        d.mipt_tonemap_device_noalloc.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_size_t, C.c_size_t]
        d.mipt_affine_inverse.argtypes = [C.c_size_t, _f32p, _f32p]
        d.mipt_rotation.argtypes = [C.c_char, C.c_size_t, _f32p, _f32p]
        d.mipt_affine_mul.argtypes = [C.c_size_t, _f32p, _f32p, _f32p]
        d.mipt_noise_transform.argtypes = [_f32p, _f32p, _f32p]
        d.mipt_world_bounds.argtypes = [C.c_uint32, _f32p, _f32p, _f32p]
        d.mipt_present_rgba8.argtypes = [C.c_void_p, C.c_void_p, C.c_float, C.c_int, C.POINTER(C.c_uint8), C.c_size_t, C.c_size_t]
        _dll = d
    return _dll


def _strerror(status):
    try:
        return dll().mipt_strerror(status).decode()
    except Exception:   # noqa: BLE001
        return "?"


def check(status, what):
    if status != OK:
        raise MiptError(status, what)


def f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def fp(a):
    return a.ctypes.data_as(_f32p)


def make_camera(cam16):
    cam16 = f32(cam16)
    c = Camera()
    for i in range(12):
        c.transform[i] = float(cam16[i])
    c.aspect, c.fov_degrees, c.focal_distance, c.blur_radius = (float(x) for x in cam16[12:16])
    return c
