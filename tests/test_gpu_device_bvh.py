"""The tree built on the GPU (mipt_scene_upload_ex with MIPT_UPLOAD_DEVICE_BVH) against the host-built one and
the oracle's scan over every primitive.  The reference's tree never culls (src/ObjectTree.cpp:22-63), so the
closest hit does not depend on the topology: indices, distances and whole frames must be bit-identical."""
import os
import time

import numpy as np
import pytest

import golden_cases as G
from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "reference_cases.npz")


@pytest.fixture(scope="module")
def ctx():
    c = api.Context()
    yield c
    c.close()


@pytest.fixture(scope="module")
def alg():
    return api.HostAlgebra()


SCENES = {
    "sphere_box": (lambda a: G.sphere_box_scene(a), lambda a: G.parity_camera(a), 96, 96),
    "mixed": (lambda a: G.mixed_scene(a), lambda a: G.parity_camera(a, 4.0 / 3.0), 96, 72),
    "cornell": (lambda a: S.cornell_scene(a), lambda a: S.cornell_camera(a), 64, 64),
    "random2000": (lambda a: S.random_scene(a, 2000, seed=3, radius_scale=4.0), lambda a: S.random_scene_camera(a, aspect=1.0), 192, 192),
}


@pytest.mark.parametrize("name", sorted(SCENES))
def test_primary_hits_do_not_depend_on_the_builder(name, ctx, alg, oracle):
    mk_scene, mk_cam, w, h = SCENES[name]
    flat, cam = mk_scene(alg), mk_cam(alg)
    dev = ctx.upload(flat, device_bvh=True)
    info = dev.info()
    assert info["nodes"] == max(1, info["analytic"] - 1) and 1 <= info["depth"] <= 60
    got_prim, got_t = ctx.primary_hits(dev, cam, w, h)
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, w, h)
    assert np.array_equal(want_prim, got_prim), f"{np.sum(want_prim != got_prim)} primitive indices differ"
    assert np.array_equal(want_t, got_t)
    host_prim, host_t = ctx.primary_hits(ctx.upload(flat, device_bvh=False), cam, w, h)
    assert np.array_equal(host_prim, got_prim) and np.array_equal(host_t, got_t)
    assert (want_prim >= 0).mean() > 0.2


def test_frames_do_not_depend_on_the_builder(ctx, alg):
    """Whole frames with bounces, glass and mirrors, and with an SDF object beside the analytic ones: identical
    bit for bit."""
    noise = np.load(FIXTURE, allow_pickle=False)["out_noise12"]
    for flat, cam, w, h, bounces, spp in (
        (S.random_scene(alg, 900, seed=9, radius_scale=7.0, glass_mirror_mix=True), S.random_scene_camera(alg, aspect=1.0), 64, 64, 16, 8),
        (S.demo_scene(alg, noise), G.demo_closeup_camera(alg), 48, 48, 6, 4),
    ):
        a, sa = ctx.render(ctx.upload(flat, device_bvh=False), cam, w, h, bounces, spp, seed=4)
        b, sb = ctx.render(ctx.upload(flat, device_bvh=True), cam, w, h, bounces, spp, seed=4)
        assert np.array_equal(a, b)
        assert sa["rays"] == sb["rays"] and sa["rays"] > sa["paths"]


def test_coincident_and_degenerate_boxes(ctx, alg, oracle):
    """Many objects at one place (equal Morton codes), and a scene flat in one axis (zero extent)."""
    flat = S.random_scene(alg, 300, seed=5, radius_scale=4.0)
    flat.fwd[:200] = flat.fwd[0]          # 200 objects share one transform: same centre, same code
    flat.fwd[200:, 10] = flat.fwd[200, 10]   # the rest in one horizontal plane
    flat.inv[:] = alg.affine_inverse_batch(flat.fwd)
    cam = S.random_scene_camera(alg, aspect=1.0)
    dev = ctx.upload(flat, device_bvh=True)
    assert dev.info()["depth"] <= 60
    got_prim, got_t = ctx.primary_hits(dev, cam, 128, 128)
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, 128, 128)
    assert np.array_equal(want_prim, got_prim) and np.array_equal(want_t, got_t)
    assert (want_prim >= 0).any()


def test_deep_trees_walk_with_the_deep_stacks(ctx, alg, oracle):
    """Spheres whose centres and radii grow geometrically: the surface-area heuristic peels them off one or two at a time, the
    tree is a chain, and the traversal runs in its instantiations with the deepest stacks.  First hits and 6-bounce radiance
    against the oracle's scan, for both builders."""
    n = 58
    flat = S.random_scene(alg, n, seed=2, radius_scale=1.0)
    flat.kind[:] = S.SPHERE
    x = (1.5 ** np.arange(n)).astype(np.float32)
    flat.fwd[:] = 0.0
    flat.fwd[:, 0] = flat.fwd[:, 4] = flat.fwd[:, 8] = 1.0
    flat.fwd[:, 9] = x
    flat.params[:] = 0.0
    flat.params[:, 0] = 0.2 * x
    flat.inv[:] = alg.affine_inverse_batch(flat.fwd)
    r = np.random.RandomState(3)
    i = r.randint(0, n, 4096)
    c = np.zeros((4096, 3), np.float32)
    c[:, 0] = x[i]
    rad = (0.2 * x[i])[:, None].astype(np.float32)
    away = r.normal(size=(4096, 3)).astype(np.float32)
    away /= np.linalg.norm(away, axis=1, keepdims=True)
    o = (c + 3.0 * rad * away).astype(np.float32)
    target = (c + 0.7 * rad * r.uniform(-1, 1, (4096, 3))).astype(np.float32)
    d = (target - o).astype(np.float32)
    d /= np.linalg.norm(d, axis=1, keepdims=True)
    osc = oracle.scene(flat)
    wp, wt, _ = osc.closest_hit(o, d)
    want = osc.trace(o, d, 6, rng_mode=1, seed=8)
    depths = []
    for dev_build in (False, True):
        ds = ctx.upload(flat, device_bvh=dev_build)
        depths.append(ds.info()["depth"])
        got, prim, t = ctx.trace_rays(ds, o, d, 6, seed=8, first_hit=True)
        assert np.array_equal(wp, prim) and np.array_equal(wt[wp >= 0], t[wp >= 0])
        close = np.all(np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
        assert close.mean() >= 0.995, close.mean()
    print(f"chain of {n} spheres: tree depth {depths[0]} (host build), {depths[1]} (device build)")
    assert (wp >= 0).mean() > 0.8
    # every instantiation of the traversal kernels on one scene (one ray per lane: stacks of 8, 24, 32, 36, 48 and 60 entries;
    # two rays per lane: shared columns of 32, 36, 38, 40 and 63 rows): identical frames
    flat = S.random_scene(alg, 40, seed=4, radius_scale=12.0, glass_mirror_mix=True)
    cam = S.random_scene_camera(alg, aspect=1.0)
    frames = []
    for need in ("0", "20", "30", "34", "37", "40", "60"):
        os.environ["MIPT_MIN_STACK"] = need
        try:
            c = api.Context()
        finally:
            os.environ.pop("MIPT_MIN_STACK", None)
        try:
            img, st = c.render(c.upload(flat), cam, 96, 96, 8, 8, seed=6)
            frames.append((img, st["rays"]))
        finally:
            c.close()
    for img, rays in frames[1:]:
        assert np.array_equal(img, frames[0][0]) and rays == frames[0][1]
    assert frames[0][0].max() > 0


def test_100k_primitives_device_build_is_fast_and_exact(ctx, alg, oracle):
    flat = S.random_scene(alg, 100_000, seed=7)
    cam = S.random_scene_camera(alg)
    r = np.random.RandomState(12)
    sx, sy = r.uniform(0, 1, 2048).astype(np.float32), r.uniform(0, 1, 2048).astype(np.float32)
    d, o = oracle.camera_rays(cam, sx, sy, rng_mode=1, seed=1)
    ctx.upload(flat, device_bvh=True).free()   # first use pays for loading the kernels
    t0 = time.perf_counter()
    dev = ctx.upload(flat, device_bvh=True)
    t_dev = time.perf_counter() - t0
    t0 = time.perf_counter()
    host = ctx.upload(flat, device_bvh=False)
    t_host = time.perf_counter() - t0
    info = dev.info()
    assert info["analytic"] == 100_000 and info["nodes"] == 99_999 and 17 <= info["depth"] <= 60
    print(f"upload of 100,000 primitives: device build {t_dev * 1e3:.1f} ms, host build {t_host * 1e3:.1f} ms, depths {info['depth']} / {host.info()['depth']}")
    assert t_dev < t_host
    _, prim, t = ctx.trace_rays(dev, o, d, 1, seed=1, first_hit=True)
    wp, wt, _ = oracle.scene(flat).closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    a, _ = ctx.render(host, cam, 160, 90, 8, 2, seed=3)
    b, _ = ctx.render(dev, cam, 160, 90, 8, 2, seed=3)
    assert np.array_equal(a, b)


def test_unknown_upload_flags_are_refused_and_small_scenes_need_no_tree(ctx, alg):
    flat = G.sphere_box_scene(alg)
    with pytest.raises(api.MiptError) as e:
        api.DeviceScene(ctx, flat, flags=2)
    assert e.value.status == -1
    noise = np.load(FIXTURE, allow_pickle=False)["out_noise12"]
    one = ctx.upload(S.demo_scene(alg, noise), device_bvh=True)   # one analytic primitive beside the crater ball
    assert one.info()["analytic"] == 1 and one.info()["nodes"] == 1 and one.info()["depth"] == 1
