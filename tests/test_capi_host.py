"""GPU-free checks of the C ABI: the shared library loads, exports every symbol include/mipt.h
declares, and its host-side helpers return the same bits as the CPU restatement."""
import ctypes as C
import os
import re

import numpy as np
import pytest

from pathtracer_amd import api, capi, scene as S

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_library_exports_every_declared_symbol():
    header = open(os.path.join(ROOT, "include", "mipt.h")).read()
    declared = sorted(set(re.findall(r"\b(mipt_[a-z_0-9]+)\s*\(", header)))
    assert declared == sorted(capi.EXPORTS)
    d = capi.dll()
    for name in declared:
        assert hasattr(d, name), name


def test_init_without_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = C.c_void_p()
    rc = capi.dll().mipt_init(-1, None, C.byref(h))
    assert rc == -2 and not h.value
    with pytest.raises(api.MiptError):
        api.Context()
    assert capi.dll().mipt_strerror(-2).decode() == "no usable HIP device"


def test_host_helpers_match_oracle_bitwise(oracle):
    alg = api.HostAlgebra()
    rs = np.random.RandomState(0)
    angles = rs.uniform(-4, 4, 50).astype(np.float32)
    for axis in "xyz":
        got = alg.rotation_batch(axis, angles)
        want = np.stack([oracle.rotation(axis, float(a)) for a in angles])
        assert np.array_equal(got, want)
    a = rs.normal(size=(40, 12)).astype(np.float32)
    b = rs.normal(size=(40, 12)).astype(np.float32)
    assert np.array_equal(alg.affine_mul(a, b), np.stack([oracle.affine_mul(x, y) for x, y in zip(a, b)]))
    for f in a:
        want = oracle.affine_inverse(f)
        got = alg.affine_inverse(f)
        assert (want is None) == (got is None)
        if want is not None:
            assert np.array_equal(want, got)
    assert alg.affine_inverse(S.scale(0.009)) is None    # det < 1e-6 is rejected like the reference does
    t, r = rs.uniform(0, 1, 3), rs.uniform(0, 1, 3)
    assert np.array_equal(alg.noise_transform(t, r), oracle.noise_transform(t, r))


def test_world_bounds_match_oracle(oracle):
    alg = api.HostAlgebra()
    rs = np.random.RandomState(1)
    for kind in range(4):
        for _ in range(10):
            params = rs.uniform(0.1, 2.0, 9).astype(np.float32)
            rot = alg.affine_mul(S.linear_to_affine(alg.rotation("x", float(rs.uniform(-3, 3)))),
                                 S.linear_to_affine(alg.rotation("z", float(rs.uniform(-3, 3)))))
            fwd = alg.affine_mul(S.translation(*rs.uniform(-5, 5, 3)), alg.affine_mul(rot, S.scale(float(rs.uniform(0.5, 2)))))
            got = alg.world_bounds(kind, fwd, params)
            want = oracle.world_bbox(kind, fwd, params)     # centre, half (AxisAlignedBox(Pos, Pos))
            lo, hi = got[:3], got[3:]
            centre = lo + np.float32(0.5) * (hi - lo)
            half = np.float32(0.5) * np.abs(lo - hi)
            assert np.array_equal(np.concatenate([centre, half]).astype(np.float32), want)


def test_scene_builders_agree_between_algebras(oracle):
    """Scenes built with the product's helpers equal scenes built with the oracle's."""
    alg = api.HostAlgebra()
    for mk in (S.demo_scene, S.cornell_scene):
        a, b = mk(alg), mk(oracle)
        for f in ("kind", "fwd", "inv", "params", "material", "materials", "noise"):
            assert np.array_equal(getattr(a, f), getattr(b, f)), f
    a, b = S.random_scene(alg, 64, seed=5), S.random_scene(oracle, 64, seed=5)
    assert np.array_equal(a.inv, b.inv) and np.array_equal(a.fwd, b.fwd)


def test_no_entry_point_takes_the_context_lock_twice():
    """The context's mutex is not recursive: an entry point that holds it must not call another entry point that takes
    it (mipt_tonemap once did, and hung).  Checked on the source: every extern "C" body with a lock_guard is free of
    calls to the other locking entry points."""
    src = open(os.path.join(os.path.dirname(capi.LIB_PATH), "..", "csrc", "api.cpp")).read()
    bodies = {}
    for m in re.finditer(r"^(?:int|void|const char\*)\s+(mipt_\w+)\s*\([^)]*\)\s*\{", src, re.M):
        depth, i = 1, m.end()
        while depth:
            depth += {"{": 1, "}": -1}.get(src[i], 0)
            i += 1
        bodies[m.group(1)] = src[m.end():i]
    assert len(bodies) >= 20
    locking = {name for name, body in bodies.items() if "lock_guard" in body}
    assert {"mipt_render", "mipt_tonemap", "mipt_tonemap_device", "mipt_scene_upload_host"} <= locking
    for name in locking:
        called = set(re.findall(r"\b(mipt_\w+)\s*\(", bodies[name])) & locking
        assert not called, f"{name} holds the context lock and calls {sorted(called)}"
