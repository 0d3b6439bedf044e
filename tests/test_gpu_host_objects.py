"""Objects without a device form (MIPT_HOST_OBJECT: any user subclass of Object, include/Object.hpp:45-52): rays that
reach their box are evaluated by the caller's callbacks on the host, between the device stages.

The callbacks here answer for a sphere and a box exactly as the CPU restatement's Object::hitRay does, so the frame must
show what the same scene shows with those two objects as device primitives: primary hits bit for bit (closest of ALL
objects, ties to the lower index), images statistically (the callbacks' deflect draws its own random numbers).
"""
import dataclasses
import os
import subprocess

import numpy as np
import pytest

import golden_cases as G
from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu

LIB = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "pathtracer_amd", "lib")


@pytest.fixture(scope="module")
def ctx():
    c = api.Context()
    yield c
    c.close()


def _as_host_objects(alg, flat, which, oracle):
    """`flat` with primitives `which` turned into host objects answered by the restatement's single-object scenes."""
    kind = flat.kind.copy()
    params = flat.params.copy()
    singles = {}
    for i in which:
        one = S.FlatScene(flat.kind[i:i + 1], flat.fwd[i:i + 1], flat.inv[i:i + 1], flat.params[i:i + 1], np.zeros(1, np.uint32),
                          flat.materials[flat.material[i]:flat.material[i] + 1], flat.noise)
        singles[i] = (oracle.scene(one), flat.materials[flat.material[i]])
        box = alg.world_bounds(int(flat.kind[i]), flat.fwd[i], flat.params[i])
        params[i, :3] = 0.5 * (box[:3] + box[3:])
        params[i, 3:6] = 0.5 * (box[3:] - box[:3])
        kind[i] = 8
    hosted = dataclasses.replace(flat, kind=kind, params=params)
    calls = {"hit": 0, "deflect": 0}

    def hit(prim, pos, d):
        calls["hit"] += 1
        p, _, hp = singles[prim][0].closest_hit(pos[None, :], d[None, :])
        return hp[0] if p[0] >= 0 else None

    def deflect(prim, pos, d):
        calls["deflect"] += 1
        _, mat = singles[prim]
        # Object::deflectRay (src/Object.cpp:35-44) with a purely diffuse material: the local normal at the hit, a
        # cosine-weighted direction about it, the direction (only) taken back to world space
        inv, fwd = flat.inv[prim], flat.fwd[prim]
        local = inv[:9].reshape(3, 3) @ pos + inv[9:]
        n = oracle.normal_local(int(flat.kind[prim]), flat.params[prim], flat.noise, local[None, :].astype(np.float32))[0].astype(np.float64)
        if (inv[:9].reshape(3, 3) @ d) @ n > 0:
            n = -n   # struck from inside: leave through the surface
        r = np.random.default_rng(calls["deflect"])
        v = r.normal(size=3)
        v = n + v / np.linalg.norm(v)
        v /= max(np.linalg.norm(v), 1e-9)
        return np.concatenate([mat[6:9], mat[3:6], fwd[:9].reshape(3, 3) @ v]).astype(np.float32)

    return hosted, hit, deflect, calls


def test_host_objects_take_part_in_the_closest_hit_exactly(ctx, oracle):
    alg = api.HostAlgebra()
    flat = G.sphere_box_scene(alg)
    which = [int(np.flatnonzero(flat.kind == S.SPHERE)[0]), int(np.flatnonzero(flat.kind == S.BOX)[0])]
    hosted, hit, deflect, calls = _as_host_objects(alg, flat, which, oracle)
    cam = G.parity_camera(alg)
    w = h = 40
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, w, h)
    ds = ctx.upload(hosted, host_hit=hit, host_deflect=deflect)
    assert ds.info()["analytic"] == flat.n_prims - 2
    got_prim, got_t = ctx.primary_hits(ds, cam, w, h)
    assert np.array_equal(want_prim, got_prim), f"{(want_prim != got_prim).sum()} pixels see another object"
    assert np.array_equal(want_t, got_t)
    assert np.isin(got_prim, which).sum() > 10 and calls["hit"] > 0
    # a frame: the same light arrives, within the noise of 16 samples per pixel
    ref_img, _ = ctx.render(ctx.upload(flat), cam, 24, 24, 4, 16, seed=3)
    img, st = ctx.render(ds, cam, 24, 24, 4, 16, seed=3)
    assert st["host_rays"] > 0 and st["host_shades"] > 0 and calls["deflect"] == st["host_shades"]
    assert abs(img.mean() - ref_img.mean()) <= 0.1 * ref_img.mean(), (img.mean(), ref_img.mean())
    # explicit rays: first hits agree with the restatement's scan over all objects
    r = np.random.RandomState(4)
    o = r.uniform(-6, 6, (300, 3)).astype(np.float32)
    d = (r.uniform(-4, 4, (300, 3)) - o)
    d = (d / np.linalg.norm(d, axis=1, keepdims=True)).astype(np.float32)
    _, prim, t = ctx.trace_rays(ds, o, d, 3, seed=2, first_hit=True)
    wp, wt, _ = oracle.scene(flat).closest_hit(o, d)
    assert np.array_equal(wp, prim) and np.array_equal(wt[wp >= 0], t[wp >= 0])


def test_host_objects_on_the_several_gpus_path(oracle):
    """mipt_node_scene_upload_host: every shard of a node hands the rays that reach a host object to the same callbacks.
    Three simulated shards on one GPU render the frame of the single context (the callbacks' deflect draws its own random
    numbers, so the comparison is statistical) and report the summed host work."""
    alg = api.HostAlgebra()
    flat = G.sphere_box_scene(alg)
    which = [int(np.flatnonzero(flat.kind == S.SPHERE)[0]), int(np.flatnonzero(flat.kind == S.BOX)[0])]
    hosted, hit, deflect, calls = _as_host_objects(alg, flat, which, oracle)
    cam = G.parity_camera(alg)
    c = api.Context()
    try:
        want, _ = c.render(c.upload(flat), cam, 24, 24, 4, 24, seed=3)
    finally:
        c.close()
    node = api.Node(devices=[0, 0, 0], simulate=True)
    try:
        ns = node.upload(hosted, host_hit=hit, host_deflect=deflect)
        got, st = node.render(ns, cam, 24, 24, 4, 24, seed=3)
        assert st["samples_done"] == 24 and st["host_rays"] > 0 and st["host_shades"] == calls["deflect"]
        assert abs(got.mean() - want.mean()) <= 0.1 * want.mean(), (got.mean(), want.mean())
    finally:
        node.close()


def test_sdf_probes_refuse_primitives_without_a_device_sdf(ctx, oracle):
    """mipt_sdf_eval / mipt_sdf_material_eval: analytic primitives and host objects have no device SDF to ask."""
    alg = api.HostAlgebra()
    flat = G.sphere_box_scene(alg)
    which = [int(np.flatnonzero(flat.kind == S.SPHERE)[0])]
    hosted, hit, deflect, _ = _as_host_objects(alg, flat, which, oracle)
    ds = ctx.upload(hosted, host_hit=hit, host_deflect=deflect)
    pts = np.zeros((4, 3), np.float32)
    for prim in (which[0], (which[0] + 1) % flat.n_prims, flat.n_prims):
        for probe in (ctx.sdf_eval, ctx.sdf_material_eval):
            with pytest.raises(api.MiptError) as e:
                probe(ds, prim, pts)
            assert e.value.status == -1


def test_upload_refuses_host_objects_without_callbacks(ctx):
    alg = api.HostAlgebra()
    flat = S.cornell_scene(alg)
    bad = dataclasses.replace(flat, kind=flat.kind.copy())
    bad.kind[0] = 8
    with pytest.raises(api.MiptError) as e:
        ctx.upload(bad)
    assert e.value.status == -1


def _load_image(path):
    raw = open(path, "rb").read()
    w, h = np.frombuffer(raw[:16], np.uint64)
    return np.frombuffer(raw[16:16 + int(w) * int(h) * 12], np.float32).reshape(int(h), int(w), 3)


def test_user_object_subclass_renders_through_the_cpp_api(ctx, tmp_path):
    """examples/demo_headless.cpp with the light as a class of its own (UserSlab : Object, three virtuals, no device
    description): Renderer::render hands its rays to the object's virtuals and the frame shows the same ball in the same
    light as with the library's BoxObject."""
    exe = os.path.join(LIB, "demo_headless")
    if not os.path.exists(exe):
        pytest.skip("demo_headless not built (make -C pathtracer_amd/host example)")
    out_user, out_box = tmp_path / "user.img", tmp_path / "box.img"
    r = subprocess.run([exe, "96", "64", "32", "8", "5", str(out_user), "1", "0", "user"], capture_output=True, text=True, timeout=200)
    assert r.returncode == 0, r.stderr
    assert "no device form" in r.stderr and "UserSlab" in r.stderr
    r2 = subprocess.run([exe, "96", "64", "32", "8", "5", str(out_box)], capture_output=True, text=True, timeout=100)
    assert r2.returncode == 0, r2.stderr
    user, box = _load_image(out_user), _load_image(out_box)
    assert box.mean() > 1e-3 and abs(user.mean() - box.mean()) <= 0.05 * box.mean(), (user.mean(), box.mean())
    lit_user, lit_box = user.sum(axis=2) > 0, box.sum(axis=2) > 0
    assert (lit_user != lit_box).mean() < 0.02   # the same pixels see the lit ball
