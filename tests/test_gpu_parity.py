"""HIP path (through the C ABI of libmipt.so) against the CPU restatement on the same inputs.

Bars, per stage:
  * analytic primitives (sphere, box, triangle): primary-hit primitive indices identical, hit distances
    bit-identical (the exact part of the device code evaluates the reference's expressions in the
    reference's order, without fused multiply-add);
  * crater-ball SDF: the device evaluates the SDF with FMA contraction and hardware exp2/log2/rcp/rsq
    (values within 5.5e-6 of the reference's, tests/test_gpu_configs.py), so marches may stop one step apart:
    at most one pixel of a primary-hit map may disagree on hit / miss (none was measured), distances
    agree to 2e-4 (the march's own stop tolerance is 1e-4);
  * traced paths / rendered images with the shared counter-based random numbers: identical ray counts
    on analytic scenes, radiance equal to 1e-4 relative on at least 99.5 % of paths (a path whose
    rejection-sampling decision flips on a last-bit difference legitimately diverges).
"""
import os

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


@pytest.fixture(scope="module")
def golden():
    z = np.load(FIXTURE, allow_pickle=False)
    return {k[4:]: z[k] for k in z.files if k.startswith("out_")}


def _unit(v):
    return (v / np.linalg.norm(v, axis=-1, keepdims=True)).astype(np.float32)


ANALYTIC = {
    "sphere_box": (lambda a: G.sphere_box_scene(a), lambda a: G.parity_camera(a), 96, 96),
    "mixed": (lambda a: G.mixed_scene(a), lambda a: G.parity_camera(a, 4.0 / 3.0), 96, 72),
    "cornell": (lambda a: S.cornell_scene(a), lambda a: S.cornell_camera(a), 64, 64),
}


@pytest.mark.parametrize("name", sorted(ANALYTIC))
def test_primary_hits_analytic_exact(name, ctx, alg, oracle):
    mk_scene, mk_cam, w, h = ANALYTIC[name]
    flat, cam = mk_scene(alg), mk_cam(alg)
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, w, h)
    got_prim, got_t = ctx.primary_hits(ctx.upload(flat), cam, w, h)
    assert np.array_equal(want_prim, got_prim), f"{np.sum(want_prim != got_prim)} primitive indices differ"
    assert np.array_equal(want_t, got_t), f"max |dt| = {np.abs(want_t - got_t).max()}"
    assert (want_prim >= 0).sum() > 0.2 * w * h


@pytest.mark.parametrize("case", ["sb", "mx", "cb"])
def test_primary_hits_match_reference_fixture(case, ctx, alg, golden):
    """The same maps as the reference's own Tree::hit produced them (committed fixture)."""
    mk_scene, mk_cam, w, h = {"sb": ANALYTIC["sphere_box"], "mx": ANALYTIC["mixed"], "cb": ANALYTIC["cornell"]}[case]
    flat, cam = mk_scene(alg), mk_cam(alg)
    got_prim, got_t = ctx.primary_hits(ctx.upload(flat), cam, w, h)
    want_prim, want_t = golden[case + "_primary_prim"], golden[case + "_primary_t"]
    # Distances are identical everywhere.  Indices may differ only on exact ties (two primitives at
    # bit-identical t, e.g. the overlapping wall slabs in the Cornell box corners): the reference breaks
    # them by its tree's leaf order, the device by the lowest primitive index.
    assert np.array_equal(want_t, got_t)
    differ = want_prim != got_prim
    assert differ.mean() <= (0.02 if case == "cb" else 0.0)
    assert np.all((want_prim >= 0)[differ] & (got_prim >= 0)[differ] & (got_prim < want_prim)[differ])


def test_primary_hits_large_analytic(ctx, alg, oracle):
    """2,000 primitives at 256 x 256: exercises a deep BVH against the oracle's non-culling scan."""
    flat = S.random_scene(alg, 2000, seed=3, radius_scale=4.0)
    cam = S.random_scene_camera(alg, aspect=1.0)
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, 256, 256)
    ds = ctx.upload(flat)
    got_prim, got_t = ctx.primary_hits(ds, cam, 256, 256)
    assert ds.info()["depth"] >= 8
    assert np.array_equal(want_prim, got_prim)
    assert np.array_equal(want_t, got_t)
    assert (want_prim >= 0).mean() > 0.3


@pytest.mark.parametrize("close", [False, True])
def test_primary_hits_demo_scene(close, ctx, alg, oracle, golden):
    noise = golden["noise12"]
    flat = S.demo_scene(alg, noise)
    cam = G.demo_closeup_camera(alg) if close else S.demo_camera(alg)
    w = h = 48 if close else 64
    got_prim, got_t = ctx.primary_hits(ctx.upload(flat), cam, w, h)
    want_prim = golden["demo_close_prim" if close else "demo_primary_prim"]
    want_t = golden["demo_close_t" if close else "demo_primary_t"]
    mismatch = int((want_prim != got_prim).sum())
    both = (want_prim >= 0) & (got_prim >= 0)
    dt = np.abs(want_t - got_t)[both]
    print(f"reference fixture, {'close-up' if close else 'demo camera'}: {mismatch} of {w * h} pixels disagree on hit / miss, max |dt| = {dt.max():.3e}")
    # measured: no pixel disagrees at either size (tests/test_gpu_configs.py holds the 256 x 256 maps and their figures)
    assert mismatch <= 1, f"{mismatch} pixels disagree on hit / miss"
    assert both.sum() > (400 if close else 20)
    assert dt.max() <= 2e-4


def test_two_crater_balls_in_one_scene(alg, oracle, golden):
    """Two CraterBallObjects with different offsets (each has its own crater table and cell masks) side by side,
    so that one wave marches rays of both: hits against the oracle within the crater-ball bars, and the frame is
    the same whichever evaluator serves a point."""
    import os
    b = S.SceneBuilder(alg, golden["noise12"])
    m = S.material(diffuse=1.0, specular=0.0, albedo=(0.5, 0.5, 0.5))
    b.craterball((-8.0, 15.0, -6.5), m, S.translation(-0.55, 0.0, 0.0))
    b.craterball((3.25, -1.5, 7.75), m, alg.affine_mul(S.translation(0.6, 0.1, 0.3), alg.affine_mul(S.linear_to_affine(alg.rotation("z", 0.8)), S.scale(1.2))))
    b.box((6.0, 0.1, 6.0), S.material(diffuse=1.0, specular=0.0, albedo=(0.7, 0.7, 0.7)), S.translation(0.0, 1.0, 0.0))
    b.box((2.0, 0.05, 2.0), S.material(diffuse=1.0, specular=0.0, emitted=(12.0, 11.0, 10.0)), S.translation(0.0, -4.0, 0.0))
    flat = b.build()
    cam = S.camera(S.translation(0.0, 0.0, -7.0), aspect=1.0, fov=6.0, focal=10.0, blur=0.0)
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, 96, 96)
    assert (want_prim == 0).sum() > 500 and (want_prim == 1).sum() > 500
    frames = []
    old = os.environ.get("MIPT_LANE_THRESHOLD")
    try:
        for threshold in ("1", "65", "14"):
            os.environ["MIPT_LANE_THRESHOLD"] = threshold
            c = api.Context()
            ds = c.upload(flat)
            got_prim, got_t = c.primary_hits(ds, cam, 96, 96)
            assert (want_prim != got_prim).mean() <= 0.005
            both = (want_prim == got_prim) & (want_prim >= 0)
            assert np.abs(want_t - got_t)[both].max() <= 3e-4   # the second ball is scaled by 1.2
            frames.append(c.render(ds, cam, 64, 64, 6, 8, seed=3))
            c.close()
    finally:
        if old is None:
            os.environ.pop("MIPT_LANE_THRESHOLD", None)
        else:
            os.environ["MIPT_LANE_THRESHOLD"] = old
    for img, st in frames[1:]:
        assert np.array_equal(img, frames[0][0])
        assert st["sdf_near_evals"] == frames[0][1]["sdf_near_evals"]
    assert frames[0][1]["sdf_shades"] > 2000


def _random_rays(n, seed, extent, inner):
    r = np.random.RandomState(seed)
    o = r.uniform(-extent, extent, (n, 3)).astype(np.float32)
    d = _unit(r.uniform(-inner, inner, (n, 3)) - o)
    return o, d


@pytest.mark.parametrize("name", sorted(ANALYTIC))
def test_trace_rays_analytic(name, ctx, alg, oracle):
    mk_scene = ANALYTIC[name][0]
    flat = mk_scene(alg)
    o, d = _random_rays(4000, 5, 6.0, 4.0)
    want = oracle.scene(flat).trace(o, d, 8, rng_mode=1, seed=99)
    got, prim, t = ctx.trace_rays(ctx.upload(flat), o, d, 8, seed=99, first_hit=True)
    wp, wt, _ = oracle.scene(flat).closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    close = np.all(np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    assert close.mean() >= 0.995, f"only {close.mean():.4f} of paths agree"
    assert want.max() > 0


def test_trace_rays_demo_scene(ctx, alg, oracle, golden):
    noise = golden["noise12"]
    flat = S.demo_scene(alg, noise)
    r = np.random.RandomState(8)
    o = (np.array([0.0, 0.5, -2.0]) + _unit(r.normal(size=(1500, 3))) * 3.0).astype(np.float32)
    d = _unit(np.array([0.0, 0.5, -2.0]) + r.uniform(-0.6, 0.6, (1500, 3)) - o)
    osc = oracle.scene(flat)
    want = osc.trace(o, d, 8, rng_mode=1, seed=4)
    got, prim, t = ctx.trace_rays(ctx.upload(flat), o, d, 8, seed=4, first_hit=True)
    wp, wt, _ = osc.closest_hit(o, d)
    assert (wp != prim).mean() <= 0.01
    both = (wp >= 0) & (prim >= 0)
    assert both.sum() > 500
    assert np.abs(wt - t)[both].max() <= 2e-4
    # radiance: same random numbers, slightly different normals -> compare in aggregate and per path loosely
    assert abs(want.mean() - got.mean()) <= 0.03 * want.mean()
    close = np.all(np.abs(want - got) <= 0.02 + 0.02 * np.abs(want), axis=1)
    assert close.mean() >= 0.97


@pytest.mark.parametrize("name", sorted(ANALYTIC))
def test_render_matches_oracle_with_shared_random_numbers(name, ctx, alg, oracle):
    mk_scene, mk_cam, w, h = ANALYTIC[name]
    flat, cam = mk_scene(alg), mk_cam(alg)
    w, h = w // 2, h // 2
    want, wstats = oracle.scene(flat).render(cam, w, h, 6, 8, rng_mode=1, seed=7, threads=0, full_width=True)
    got, gstats = ctx.render(ctx.upload(flat), cam, w, h, 6, 8, seed=7)
    assert gstats["paths"] == wstats["paths"] == w * h * 8
    assert abs(gstats["rays"] - wstats["rays"]) <= 2e-3 * wstats["rays"]
    close = np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.99
    assert abs(want.mean() - got.mean()) <= 2e-3 * want.mean()


@pytest.mark.parametrize("aspect,w,h", [(1.0, 56, 56), (16.0 / 9.0, 96, 54), (0.6, 36, 60)])
def test_render_with_lens_blur_matches_oracle(aspect, w, h, ctx, alg, oracle):
    """Camera::getViewRay with a lens (src/Camera.cpp:65-81: a random point of the aperture per sample, focus at
    focalDistance), at several aspect ratios and with a rotated, scaled camera: the device draws the same
    aperture samples as the restatement in counter mode, so the images agree pixel by pixel; nothing is
    culled on screen because a lens has no single centre of projection."""
    flat = G.mixed_scene(alg)
    t = alg.affine_mul(S.translation(0.4, -0.3, -14.0), alg.affine_mul(S.linear_to_affine(alg.rotation("y", 0.05)), S.scale(1.3)))
    cam = S.camera(t, aspect=aspect, fov=20.0, focal=9.0, blur=0.35)
    want, wstats = oracle.scene(flat).render(cam, w, h, 5, 8, rng_mode=1, seed=21, threads=0, full_width=True)
    got, gstats = ctx.render(ctx.upload(flat), cam, w, h, 5, 8, seed=21)
    assert gstats["primary_traced"] == w * h * 8
    assert abs(gstats["rays"] - wstats["rays"]) <= 2e-3 * wstats["rays"]
    close = np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.99
    assert want.mean() > 0.01 and abs(want.mean() - got.mean()) <= 2e-3 * want.mean()
    sharp = S.camera(t, aspect=aspect, fov=20.0, focal=9.0, blur=0.0)
    crisp, _ = ctx.render(ctx.upload(flat), sharp, w, h, 5, 8, seed=21)
    assert np.abs(crisp - got).mean() > 1e-3 * got.mean()   # the lens does change the picture


def test_render_sample_sharding_is_consistent(ctx, alg):
    """Samples [0,8) rendered at once equal samples [0,4) + [4,8) rendered separately (what two GPUs
    would compute), up to the order of float additions."""
    flat, cam = S.cornell_scene(alg), S.cornell_camera(alg)
    ds = ctx.upload(flat)
    full, _ = ctx.render(ds, cam, 40, 40, 8, 8, seed=3)
    a, _ = ctx.render(ds, cam, 40, 40, 8, 4, seed=3, first_sample=0)
    b, _ = ctx.render(ds, cam, 40, 40, 8, 4, seed=3, first_sample=4)
    assert np.allclose(full, 0.5 * (a + b), rtol=1e-5, atol=1e-6)
    again, _ = ctx.render(ds, cam, 40, 40, 8, 8, seed=3)
    assert np.array_equal(full, again), "renders must be bitwise reproducible"


def test_render_demo_scene_matches_oracle(ctx, alg, oracle, golden):
    noise = golden["noise12"]
    flat = S.demo_scene(alg, noise)
    cam = G.demo_closeup_camera(alg)
    want, wstats = oracle.scene(flat).render(cam, 32, 32, 8, 4, rng_mode=1, seed=11, threads=0, full_width=True)
    got, gstats = ctx.render(ctx.upload(flat), cam, 32, 32, 8, 4, seed=11)
    assert gstats["paths"] == wstats["paths"]
    assert abs(gstats["rays"] - wstats["rays"]) <= 0.01 * wstats["rays"]
    assert abs(want.mean() - got.mean()) <= 0.03 * want.mean()
    rmse = np.sqrt(np.mean((want - got) ** 2))
    assert rmse <= 0.05 * max(want.mean(), 1e-3) * 4, rmse


@pytest.mark.parametrize("kind", [0, 1])
def test_tonemap(kind, ctx, oracle, golden):
    z = np.load(FIXTURE, allow_pickle=False)
    hdr = z["in_hdr"]
    got = ctx.tonemap(kind, hdr)
    want = golden["tonemap_reinhard" if kind == 0 else "tonemap_filmic"]
    if kind == 1:
        assert np.array_equal(want, got)
    else:   # the log-average is a parallel reduction: rounding differs from the sequential float sum
        assert np.allclose(want, got, rtol=2e-5, atol=1e-7)
    big = np.random.RandomState(2).gamma(0.5, 2.0, (270, 480, 3)).astype(np.float32)
    assert np.allclose(oracle.tonemap(kind, big), ctx.tonemap(kind, big), rtol=5e-5, atol=1e-7)


def test_cancel_and_errors(ctx, alg):
    flat, cam = S.cornell_scene(alg), S.cornell_camera(alg)
    ds = ctx.upload(flat)
    with pytest.raises(api.MiptError):
        ctx.render(ds, cam, 1, 40, 8, 1)          # W - 1 == 0 divides by zero in the reference
    with pytest.raises(api.MiptError):
        ctx.render(ds, cam, 40, 40, 0, 1)
    ctx.cancel()                                   # a cancel before the call is cleared by the call
    img, st = ctx.render(ds, cam, 16, 16, 4, 2)
    assert st["paths"] == 16 * 16 * 2 and img.max() > 0


def test_upload_and_render_refuse_bad_input(ctx, alg, oracle):
    """Every refusal is a status (include/mipt.h), nothing is uploaded or rendered, and the context keeps working."""
    import dataclasses
    good = S.cornell_scene(alg)
    bad_kind = dataclasses.replace(good, kind=good.kind.copy())
    bad_kind.kind[1] = 9                                            # no such Object subclass
    bad_material = dataclasses.replace(good, material=good.material.copy())
    bad_material.material[0] = good.materials.shape[0]              # index past the material table
    singular = dataclasses.replace(good, fwd=good.fwd.copy())
    singular.fwd[0, :9] = 0.0                                       # a primitive squashed to nothing still uploads ...
    for flat, status in ((bad_kind, -1), (bad_material, -1)):
        with pytest.raises(api.MiptError) as e:
            ctx.upload(flat)
        assert e.value.status == status
    ds = ctx.upload(good)
    cam = S.cornell_camera(alg)
    for args, status in (((40, 40, 65, 1), -6),      # more bounces than the per-batch counters hold
                         ((40, 40, 8, 0), -1),       # no samples
                         ((40, 1, 8, 1), -1)):       # H - 1 == 0
        with pytest.raises(api.MiptError) as e:
            ctx.render(ds, cam, *args)
        assert e.value.status == status
    # sizes that are not multiples of the 8 x 8 pixel tiles (ragged right and bottom edges)
    img, st = ctx.render(ds, cam, 37, 19, 6, 16, seed=2)
    assert img.shape == (19, 37, 3) and st["paths"] == 37 * 19 * 16 and np.isfinite(img).all()
    want, _ = oracle.scene(good).render(cam, 37, 19, 6, 16, rng_mode=1, seed=2, threads=0, full_width=True)
    close = np.abs(want - img) <= 1e-4 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.99
    assert (img[:, 32:] > 0).any() and (img[16:] > 0).any()   # the partial tiles at the right and bottom edges are rendered


def test_config3_100k_primitives_first_hits(ctx, alg, oracle):
    """Config 3's scene (100,000 alternating spheres / Y-rotated boxes): closest hits of 2,048 camera rays
    against a brute-force scan over every primitive (the reference's tree never culls, and cannot even be
    built at this size)."""
    flat = S.random_scene(alg, 100_000, seed=7)
    cam = S.random_scene_camera(alg)
    r = np.random.RandomState(12)
    sx, sy = r.uniform(0, 1, 2048).astype(np.float32), r.uniform(0, 1, 2048).astype(np.float32)
    d, o = oracle.camera_rays(cam, sx, sy, rng_mode=1, seed=1)
    ds = ctx.upload(flat)
    info = ds.info()
    assert info["analytic"] == 100_000 and info["nodes"] == 99_999 and 17 <= info["depth"] <= 60
    _, prim, t = ctx.trace_rays(ds, o, d, 1, seed=1, first_hit=True)
    wp, wt, _ = oracle.scene(flat).closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    assert (wp >= 0).mean() > 0.2


def test_config5_mirrors_glass_emitters_16_bounces(ctx, alg, oracle):
    """Config 5's material mix (mirrors, glass, diffuse + emitters) at 16 bounces on 900 primitives:
    every path is followed through reflection, refraction and inside hits with shared random numbers."""
    flat = S.random_scene(alg, 900, seed=9, radius_scale=7.0, glass_mirror_mix=True)
    o, d = _random_rays(6000, 21, 12.0, 8.0)
    osc = oracle.scene(flat)
    want = osc.trace(o, d, 16, rng_mode=1, seed=5)
    got, prim, t = ctx.trace_rays(ctx.upload(flat), o, d, 16, seed=5, first_hit=True)
    wp, wt, _ = osc.closest_hit(o, d)
    assert np.array_equal(wp, prim) and np.array_equal(wt[wp >= 0], t[wp >= 0])
    close = np.all(np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    assert close.mean() >= 0.995, close.mean()
    assert (want.max(axis=1) > 0).mean() > 0.05
    cam = S.random_scene_camera(alg, aspect=1.0)
    wimg, wst = osc.render(cam, 48, 48, 16, 4, rng_mode=1, seed=2, threads=0, full_width=True)
    gimg, gst = ctx.render(ctx.upload(flat), cam, 48, 48, 16, 4, seed=2)
    assert abs(gst["rays"] - wst["rays"]) <= 2e-3 * wst["rays"] and wst["rays"] > 1.5 * wst["paths"]
    assert (np.abs(wimg - gimg) <= 1e-4 * np.maximum(1.0, np.abs(wimg))).all(axis=2).mean() >= 0.99


def test_sdf_value_is_independent_of_evaluator():
    """A near-field SDF value may be computed by a whole wave for one point or by one lane per point,
    depending on how many lanes of a wave wait together (timing).  Both must agree bit for bit: the
    same frame with either evaluator forced is identical, march counts included."""
    import os
    alg = api.HostAlgebra()
    flat = S.demo_scene(alg)
    cam = G.demo_closeup_camera(alg)
    frames = []
    old = os.environ.get("MIPT_LANE_THRESHOLD")
    try:
        for threshold in ("1", "65", "14"):
            os.environ["MIPT_LANE_THRESHOLD"] = threshold
            c = api.Context()
            img, st = c.render(c.upload(flat), cam, 96, 96, 8, 8, seed=11)
            frames.append((img, st))
            c.close()
    finally:
        if old is None:
            os.environ.pop("MIPT_LANE_THRESHOLD", None)
        else:
            os.environ["MIPT_LANE_THRESHOLD"] = old
    for img, st in frames[1:]:
        assert np.array_equal(img, frames[0][0])
        for key in ("rays", "sdf_rays", "sdf_shades", "sdf_near_evals", "sdf_far_evals", "sdf_noise_calls", "shade_noise_calls"):
            assert st[key] == frames[0][1][key], key
    assert frames[0][1]["sdf_near_evals"] > 10000


def test_frame_does_not_depend_on_how_the_march_is_scheduled(golden):
    """Which launch finishes a march (MIPT_CARRY_LAG), in which order a launch takes its rays (MIPT_SORT_MARCHES), whether
    the small / tiny craters' cells are read from a ball's tables or computed (MIPT_CELL_TABLE_BALLS) and which kernel
    finds the analytic hits of a scene of a handful of primitives (MIPT_NO_FEW) are matters of speed: the frame and every
    work counter must be the same.  Two crater balls (two crater tables, two pairs of cell tables), a fractal in front of
    one of them (a first march that cannot be set up ahead), a rough box (the lane-evaluated kind), a mirror and a light."""
    import os
    alg = api.HostAlgebra()
    b = S.SceneBuilder(alg, golden["noise12"])
    grey = S.material(diffuse=1.0, specular=0.0, albedo=(0.6, 0.6, 0.6))
    b.craterball((-8.0, 15.0, -6.5), grey, S.translation(-0.7, 0.4, 0.0))
    b.craterball((3.25, -1.5, 7.75), grey, alg.affine_mul(S.translation(0.9, -0.2, 0.6), S.scale(0.8)))
    b.fractal(grey, alg.affine_mul(S.translation(-0.6, 0.3, -1.6), S.scale(0.35)))
    b.roughbox((0.3, 0.25, 0.3), grey, S.translation(0.2, 1.1, -0.3))
    b.sphere(0.4, S.material(diffuse=0.05, specular=0.95, sharpness=0.95), S.translation(0.1, -0.9, -0.4))
    b.box((30.0, 0.1, 30.0), S.material(diffuse=1.0, specular=0.0, emitted=(2.0, 2.0, 2.0)), S.translation(0.0, -12.0, 0.0))
    flat = b.build()
    cam = S.camera(S.translation(0.0, 0.0, -9.0), aspect=1.0, fov=22.0, focal=9.0, blur=0.0)
    switches = ("MIPT_CARRY_LAG", "MIPT_SORT_MARCHES", "MIPT_CELL_TABLE_BALLS", "MIPT_NO_FEW")
    old = {k: os.environ.get(k) for k in switches}
    frames = []
    try:
        for env in ({}, {"MIPT_CARRY_LAG": "0"}, {"MIPT_SORT_MARCHES": "0"}, {"MIPT_SORT_MARCHES": "8"}, {"MIPT_CELL_TABLE_BALLS": "0"},
                    {"MIPT_CELL_TABLE_BALLS": "1"}, {"MIPT_NO_FEW": "1"}, {"MIPT_CARRY_LAG": "8", "MIPT_CARRY_AFTER": "2"}):
            for k in switches + ("MIPT_CARRY_AFTER",):
                os.environ.pop(k, None)
            os.environ.update(env)
            c = api.Context()
            img, st = c.render(c.upload(flat), cam, 160, 160, 6, 12, seed=5)
            frames.append((env, img, st))
            c.close()
    finally:
        for k in switches + ("MIPT_CARRY_AFTER",):
            os.environ.pop(k, None)
        for k, v in old.items():
            if v is not None:
                os.environ[k] = v
    _, img0, st0 = frames[0]
    assert st0["sdf_near_evals"] > 100000 and st0["sdf_shades"] > 10000 and img0.max() > 0
    assert frames[-1][2]["marches_carried"] > 0 and frames[1][2]["marches_carried"] == 0
    for env, img, st in frames[1:]:
        assert np.array_equal(img, img0), env
        for key in ("rays", "sdf_rays", "sdf_shades", "sdf_near_evals", "sdf_far_evals", "sdf_noise_calls", "shade_noise_calls"):
            assert st[key] == st0[key], (env, key)


def test_marches_that_cannot_end_are_cut_short_without_changing_the_frame(golden):
    """A march whose state (p, d) stops changing, or comes back bit for bit, can only run out its 256 iterations
    and miss (include/Object.hpp:135-167), and a ray that stays farther than 0.6 from a crater ball's centre only
    ever sees the far field and walks out of the box.  The kernels return these misses at once; with
    MIPT_FULL_MARCH=1 they march everything that enters a box, every iteration, as the reference does.  Frames,
    ray counts and SDF hits must be identical, evaluations fewer."""
    import os
    alg = api.HostAlgebra()
    wide = S.camera(alg.affine_mul(S.translation(0.0, 0.0, -30.0), S.scale(0.01)), aspect=1.0, fov=3.5, focal=30.0, blur=0.0)   # ball and box corners in view
    cases = [(S.demo_scene(alg), G.demo_closeup_camera(alg)), (S.demo_scene(alg), wide),
             (G.roughbox_scene(alg, golden["noise12"]), G.roughbox_camera(alg))]
    old = os.environ.get("MIPT_FULL_MARCH")
    try:
        for flat, cam in cases:
            frames = []
            for full in ("1", "0"):
                os.environ["MIPT_FULL_MARCH"] = full
                c = api.Context()
                img, st = c.render(c.upload(flat), cam, 96, 96, 8, 8, seed=11)
                frames.append((img, st))
                c.close()
            (full_img, full_st), (img, st) = frames
            assert np.array_equal(img, full_img)
            for key in ("rays", "sdf_shades"):
                assert st[key] == full_st[key], key
            assert st["sdf_rays"] <= full_st["sdf_rays"]   # rays that cannot come near a crater ball are not marched at all
            assert st["sdf_near_evals"] < 0.97 * full_st["sdf_near_evals"], (st["sdf_near_evals"], full_st["sdf_near_evals"])
    finally:
        if old is None:
            os.environ.pop("MIPT_FULL_MARCH", None)
        else:
            os.environ["MIPT_FULL_MARCH"] = old


def test_a_frame_in_several_sample_batches_equals_one_batch(alg):
    """A frame whose paths do not fit the device at once is rendered in batches of whole samples (MIPT_BATCH_PATHS forces
    that here): each batch adds its samples' sum to the frame, so only the order of float additions differs."""
    import os
    flat, cam = S.demo_scene(alg), G.demo_closeup_camera(alg)
    w, h, spp = 64, 48, 24
    old = os.environ.get("MIPT_BATCH_PATHS")
    frames = []
    try:
        for budget in (None, 64 * 48 * 5):          # one batch; five batches of 5 samples (padded tiles: 64 x 48 = 48 tiles)
            if budget is None:
                os.environ.pop("MIPT_BATCH_PATHS", None)
            else:
                os.environ["MIPT_BATCH_PATHS"] = str(budget)
            c = api.Context()
            frames.append(c.render(c.upload(flat), cam, w, h, 8, spp, seed=5))
            c.close()
    finally:
        if old is None:
            os.environ.pop("MIPT_BATCH_PATHS", None)
        else:
            os.environ["MIPT_BATCH_PATHS"] = old
    (one, st1), (many, st2) = frames
    assert np.allclose(one, many, rtol=2e-6, atol=1e-7)
    for key in ("paths", "rays", "sdf_rays", "sdf_shades", "sdf_near_evals", "primary_traced"):
        assert st1[key] == st2[key], key
    assert st1["sdf_shades"] > 1000


def test_fractal_object_is_exact(ctx, alg, oracle, golden):
    """FractalObject (src/Object.cpp:160-252) is marched with exact arithmetic in the reference's order:
    primary hits equal the reference's fixture and the oracle bit for bit, traced paths and images with
    shared random numbers meet the analytic bars."""
    flat, cam = G.fractal_scene(alg), G.fractal_camera(alg)
    ds = ctx.upload(flat)
    assert ds.info()["sdf"] == 1
    got_prim, got_t = ctx.primary_hits(ds, cam, 64, 64)
    assert np.array_equal(golden["fr_primary_prim"], got_prim)
    assert np.array_equal(golden["fr_primary_t"], got_t)
    assert (got_prim == 0).sum() > 200
    # a denser map against the oracle
    want_prim, want_t = oracle.scene(flat).primary_hits(cam, 160, 160)
    got_prim, got_t = ctx.primary_hits(ds, cam, 160, 160)
    assert np.array_equal(want_prim, got_prim)
    assert np.array_equal(want_t, got_t)
    # traced rays aimed at the lattice
    r = np.random.RandomState(12)
    o = (np.array([0.3, -0.2, 0.5]) + _unit(r.normal(size=(3000, 3))) * 6.0).astype(np.float32)
    d = _unit(np.array([0.3, -0.2, 0.5]) + r.uniform(-0.9, 0.9, (3000, 3)) - o)
    osc = oracle.scene(flat)
    want = osc.trace(o, d, 6, rng_mode=1, seed=31)
    got, prim, t = ctx.trace_rays(ds, o, d, 6, seed=31, first_hit=True)
    wp, wt, _ = osc.closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    assert (wp == 0).sum() > 300
    close = np.all(np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    assert close.mean() >= 0.995, f"only {close.mean():.4f} of paths agree"
    # image with shared random numbers
    want, wstats = osc.render(cam, 48, 48, 6, 8, rng_mode=1, seed=9, threads=0, full_width=True)
    got, gstats = ctx.render(ds, cam, 48, 48, 6, 8, seed=9)
    assert abs(gstats["rays"] - wstats["rays"]) <= 2e-3 * wstats["rays"]
    close = np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.99



def test_roughbox_object_is_exact(ctx, alg, oracle, golden):
    """The demo's RoughBoxObject (PathTracerDemo.cpp:228-270): voronoi foam, noise() octave by octave and the
    SDFObjectCRTP march are evaluated one point per lane in the reference's order of operations, so primary
    hits equal the reference's fixture and the oracle bit for bit; the per-hit albedo (hash of the foam
    cell under the hit) enters the traced radiance."""
    flat, cam = G.roughbox_scene(alg, golden["noise12"]), G.roughbox_camera(alg)
    ds = ctx.upload(flat)
    assert ds.info()["sdf"] == 1
    got_prim, got_t = ctx.primary_hits(ds, cam, 40, 40)
    assert np.array_equal(golden["rb_primary_prim"], got_prim)
    assert np.array_equal(golden["rb_primary_t"], got_t)
    assert (got_prim == 0).sum() > 150
    osc = oracle.scene(flat)
    want_prim, want_t = osc.primary_hits(cam, 96, 96)
    got_prim, got_t = ctx.primary_hits(ds, cam, 96, 96)
    assert np.array_equal(want_prim, got_prim)
    assert np.array_equal(want_t, got_t)
    # traced rays aimed at the box, from all around (and some from inside its bounding box)
    r = np.random.RandomState(14)
    c = np.array([0.2, 0.1, 0.3])
    o = (c + _unit(r.normal(size=(2000, 3))) * r.uniform(0.9, 5.0, (2000, 1))).astype(np.float32)
    d = _unit(c + r.uniform(-0.7, 0.7, (2000, 3)) - o)
    want = osc.trace(o, d, 5, rng_mode=1, seed=33)
    got, prim, t = ctx.trace_rays(ds, o, d, 5, seed=33, first_hit=True)
    wp, wt, _ = osc.closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    assert (wp == 0).sum() > 500
    close = np.all(np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want)), axis=1)
    assert close.mean() >= 0.995, f"only {close.mean():.4f} of paths agree"
    # image with shared random numbers
    want, wstats = osc.render(cam, 40, 40, 5, 4, rng_mode=1, seed=9, threads=0, full_width=True)
    got, gstats = ctx.render(ds, cam, 40, 40, 5, 4, seed=9)
    assert abs(gstats["rays"] - wstats["rays"]) <= 2e-3 * wstats["rays"]
    close = np.abs(want - got) <= 1e-4 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.99


@pytest.mark.parametrize("tag", ["rs", "ww"])
def test_roughsphere_and_weirdwall_within_tolerance(ctx, alg, oracle, golden, tag):
    """RoughSphereObject (PathTracerDemo.cpp:156-226, a procedural material made of SDF normals) and WeirdWallObject
    (:273-349, sheared voronoi bricks) call exp2 / log2 / exp / hypot / atan2, whose device versions differ from the host
    library's in the last bits: the march (stop tolerance 1e-4 in local units) may end a step apart.  Bars: hit / miss
    disagreement <= 0.5 % of the pixels, |dt| <= 5e-4 on 99 % of the common hits, image means within 3 %."""
    if tag == "rs":
        flat, cam, n = G.roughsphere_scene(alg, golden["noise12"]), G.sdf_camera(alg, -8.0, 14.0), 32
    else:
        flat, cam, n = G.weirdwall_scene(alg, golden["noise12"]), G.sdf_camera(alg, -14.0, 20.0, -0.5), 32
    ds = ctx.upload(flat)
    assert ds.info()["sdf"] == 1
    osc = oracle.scene(flat)

    def check(want_prim, want_t, got_prim, got_t):
        assert (want_prim != got_prim).mean() <= 0.005, (want_prim != got_prim).mean()
        both = (want_prim == 0) & (got_prim == 0)
        assert both.sum() > 0.15 * want_prim.size
        dt = np.abs(want_t[both] - got_t[both])
        assert np.quantile(dt, 0.99) <= 5e-4, (np.quantile(dt, 0.99), dt.max())
        rest = (want_prim == got_prim) & (want_prim != 0)
        assert np.array_equal(want_t[rest], got_t[rest])   # the analytic primitives stay exact

    got_prim, got_t = ctx.primary_hits(ds, cam, n, n)
    check(golden[tag + "_primary_prim"], golden[tag + "_primary_t"], got_prim, got_t)   # the reference's own map
    want_prim, want_t = osc.primary_hits(cam, 80, 80)
    got_prim, got_t = ctx.primary_hits(ds, cam, 80, 80)
    check(want_prim, want_t, got_prim, got_t)
    want, wstats = osc.render(cam, 40, 40, 4, 8, rng_mode=1, seed=9, threads=0, full_width=True)
    got, gstats = ctx.render(ds, cam, 40, 40, 4, 8, seed=9)
    assert abs(gstats["rays"] - wstats["rays"]) <= 5e-3 * wstats["rays"]
    assert abs(got.mean() - want.mean()) <= 0.03 * want.mean(), (got.mean(), want.mean())
    close = np.abs(want - got) <= 1e-3 * np.maximum(1.0, np.abs(want))
    assert close.all(axis=2).mean() >= 0.9, close.all(axis=2).mean()


def test_every_sdf_kind_in_one_scene(ctx, alg, oracle, golden):
    """A crater ball, a rough box and a fractal lattice share one scene (and one march kernel) with analytic
    primitives: the exact kinds keep their bit-exact hits, the crater ball its march tolerance."""
    b = S.SceneBuilder(alg, golden["noise12"])
    m = S.material(diffuse=0.9, specular=0.1, sharpness=0.5)
    b.craterball((0.3, 0.6, 0.1), m, alg.affine_mul(S.translation(-3.2, 0.3, 0.0), S.scale(1.4)))
    b.roughbox((0.35, 0.3, 0.4), m, S.translation(3.2, 0.5, 0.2))
    b.fractal(m)   # at the origin: its march enters the WORLD box with the LOCAL ray (src/Object.cpp:161)
    b.box((8.0, 0.1, 6.0), S.material(diffuse=1.0, specular=0.0, albedo=(0.7, 0.7, 0.7)), S.translation(0.0, 2.2, 0.0))
    b.sphere(0.4, S.material(diffuse=0.0, specular=1.0, sharpness=1.0), S.translation(1.8, 1.6, -1.6))
    b.box((2.0, 0.05, 2.0), S.material(diffuse=1.0, specular=0.0, emitted=(12.0, 11.0, 10.0)), S.translation(0.0, -5.0, 0.0))
    flat = b.build()
    cam = S.camera(S.translation(0.0, -0.5, -16.0), aspect=1.0, fov=17.0, focal=10.0, blur=0.0)
    ds = ctx.upload(flat)
    assert ds.info()["sdf"] == 3
    osc = oracle.scene(flat)
    want_prim, want_t = osc.primary_hits(cam, 96, 96)
    got_prim, got_t = ctx.primary_hits(ds, cam, 96, 96)
    for k in range(6):
        assert (want_prim == k).sum() > 25, k
    ball = (want_prim == 0) | (got_prim == 0)
    assert np.array_equal(want_prim[~ball], got_prim[~ball])
    assert np.array_equal(want_t[~ball], got_t[~ball])
    assert (want_prim != got_prim).mean() <= 0.005
    both = (want_prim == 0) & (got_prim == 0)
    assert np.abs(want_t[both] - got_t[both]).max() <= 1e-3   # ball scaled by 1.4, t in units of the ray direction
    want, wstats = osc.render(cam, 48, 48, 5, 4, rng_mode=1, seed=5, threads=0, full_width=True)
    got, gstats = ctx.render(ds, cam, 48, 48, 5, 4, seed=5)
    assert abs(gstats["rays"] - wstats["rays"]) <= 5e-3 * wstats["rays"]
    assert abs(got.mean() - want.mean()) <= 0.03 * want.mean()


def test_screen_culling_never_drops_a_hit(ctx, alg, oracle):
    """Pixels outside the projected bounds of every object trace no ray.  Cameras at odd places -- inside
    the scene's box, looking away from it, rotated, scaled, wide and narrow, objects straddling the plane
    the rays start from -- must still give the oracle's primary-hit map exactly."""
    flat = G.mixed_scene(alg)
    osc = oracle.scene(flat)
    ds = ctx.upload(flat)
    r = np.random.RandomState(17)
    n_hits = 0
    for i in range(10):
        rot = alg.affine_mul(S.linear_to_affine(alg.rotation("y", float(r.uniform(-3.1, 3.1)))),
                             S.linear_to_affine(alg.rotation("x", float(r.uniform(-0.8, 0.8)))))
        where = r.uniform(-3.0, 3.0, 3) if i % 2 else r.uniform(-12.0, 12.0, 3)
        t = alg.affine_mul(S.translation(*[float(v) for v in where]), alg.affine_mul(rot, S.scale(float(r.choice([0.01, 0.3, 1.0])))))
        cam = S.camera(t, aspect=float(r.choice([0.6, 1.0, 1.9])), fov=float(r.choice([8.0, 25.0, 60.0])), focal=10.0, blur=0.0)
        want_prim, want_t = osc.primary_hits(cam, 64, 48)
        got_prim, got_t = ctx.primary_hits(ds, cam, 64, 48)
        assert np.array_equal(want_prim, got_prim), f"camera {i}: {np.sum(want_prim != got_prim)} pixels differ"
        assert np.array_equal(want_t, got_t)
        n_hits += int((want_prim >= 0).sum())
    assert n_hits > 3000


def test_present_rgba8_is_the_demo_display_step(ctx, alg):
    """mipt_present_rgba8 = scaleImage(acc, 1/count), optional tone map, copyToSFImage (clamp, * 255,
    truncate; PathTracerDemo.cpp:524-545, 963-971) for an accumulator that stays on the device."""
    import torch
    flat, cam = S.cornell_scene(alg), S.cornell_camera(alg)
    ds = ctx.upload(flat)
    w, h = 72, 40
    acc = torch.zeros((h, w, 3), dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    for frame in range(3):   # three progressive frames of 4 samples each, accumulated on the device
        ctx.render_accumulate(ds, cam, w, h, 6, 4, acc.data_ptr(), seed=1, first_sample=4 * frame)
    scale = np.float32(1.0 / 12.0)
    mean = acc.cpu().numpy() * scale
    want = (np.clip(mean, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8)
    got = ctx.present_rgba8(acc.data_ptr(), float(scale), -1, w, h)
    assert np.array_equal(got[..., :3], want) and np.all(got[..., 3] == 255)
    assert want.max() == 255 and 0 < (want > 0).mean() < 1
    mapped = ctx.tonemap(1, mean)
    got = ctx.present_rgba8(acc.data_ptr(), float(scale), 1, w, h)
    assert np.array_equal(got[..., :3], (np.clip(mapped, 0.0, 1.0) * np.float32(255.0)).astype(np.uint8))
    full, _ = ctx.render(ds, cam, w, h, 6, 12, seed=1)
    assert np.allclose(mean, full, rtol=1e-5, atol=1e-6)
    # the tone mappers on device buffers (what the multi-GPU path uses after its all-reduce) equal the host-buffer calls
    hdr = torch.from_numpy(mean).cuda()
    ldr = torch.empty_like(hdr)
    torch.cuda.synchronize()
    for kind in (0, 1):
        ctx.tonemap_device(kind, hdr.data_ptr(), ldr.data_ptr(), w, h)
        assert np.array_equal(ldr.cpu().numpy(), ctx.tonemap(kind, mean))


def test_config4_1m_primitives_first_hits(ctx, alg, oracle):
    """Config 4's scene (1,000,000 primitives, radii scaled by 0.46): closest hits of 512 camera rays
    against the brute-force scan over every primitive."""
    flat = S.random_scene(alg, 1_000_000, seed=7, radius_scale=0.46)
    cam = S.random_scene_camera(alg)
    r = np.random.RandomState(13)
    sx, sy = r.uniform(0, 1, 512).astype(np.float32), r.uniform(0, 1, 512).astype(np.float32)
    d, o = oracle.camera_rays(cam, sx, sy, rng_mode=1, seed=1)
    ds = ctx.upload(flat)
    info = ds.info()
    assert info["analytic"] == 1_000_000 and 20 <= info["depth"] <= 60
    _, prim, t = ctx.trace_rays(ds, o, d, 1, seed=1, first_hit=True)
    wp, wt, _ = oracle.scene(flat).closest_hit(o, d)
    assert np.array_equal(wp, prim)
    assert np.array_equal(wt[wp >= 0], t[wp >= 0])
    assert (wp >= 0).mean() > 0.2


def test_cancel_drains_a_running_frame(alg):
    """Renderer::cancelRender from another thread (src/Renderer.cpp:174-176): the kernels of the frame that
    have not started yet return at once, so render comes back long before the frame would have finished."""
    import threading
    import time
    from pathtracer_amd import capi
    c = api.Context()
    flat, cam = S.random_scene(alg, 100_000, seed=7), S.random_scene_camera(alg, 16.0 / 9.0)
    ds = c.upload(flat)
    c.render(ds, cam, 64, 36, 8, 1)                     # warm up (allocations, code objects)
    t0 = time.perf_counter()
    c.render(ds, cam, 1920, 1080, 8, 128, seed=1)       # the frame, uncancelled
    full_s = time.perf_counter() - t0
    result = {}

    started = threading.Event()

    def work():
        t1 = time.perf_counter()
        started.set()
        try:
            c.render(ds, cam, 1920, 1080, 8, 128, seed=1)
            result["status"] = 0
        except api.MiptError as e:
            result["status"] = e.status
        result["seconds"] = time.perf_counter() - t1

    th = threading.Thread(target=work)
    th.start()
    started.wait()
    time.sleep(0.05)                                     # the call has long begun (it clears a cancel that precedes it)
    c.cancel()
    th.join()
    assert result["status"] == capi.ERR_CANCELLED
    assert result["seconds"] < 0.6 * full_s, (result, full_s)
    img, st = c.render(ds, cam, 64, 36, 8, 2)            # the context is usable afterwards
    assert st["paths"] == 64 * 36 * 2 and img.max() > 0
    c.close()
