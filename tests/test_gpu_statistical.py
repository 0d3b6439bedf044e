"""Converged images: the GPU at N samples per pixel against the reference's own Renderer at N spp
(tests/golden/reference_converged.npz, rendered by oracle/_ref/libref.so with two seeds).

The reference draws from minstd_rand0 and the device from its counter generator, so only statistics
can agree.  Acceptance: RMSE(GPU, ref A) <= 1.25 x RMSE(ref A, ref B), the
reference's own seed-to-seed noise floor, and (at 4 x N on the device) the mean relative error of the
8x8 box-filtered images <= 2 % and the whole-image mean within 1 %."""
import os

import numpy as np
import pytest

import golden_cases as G
from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "reference_converged.npz")

CASES = {
    "cornell": (lambda a, n: S.cornell_scene(a), lambda a: S.cornell_camera(a)),
    "mixed": (lambda a, n: G.mixed_scene(a), lambda a: G.parity_camera(a, 4.0 / 3.0)),
    "demo_close": (lambda a, n: S.demo_scene(a, n), lambda a: G.demo_closeup_camera(a)),
    "roughbox": (lambda a, n: G.roughbox_scene(a, n), lambda a: G.roughbox_camera(a)),
    "roughsphere": (lambda a, n: G.roughsphere_scene(a, n), lambda a: G.sdf_camera(a, -8.0, 14.0)),
    "weirdwall": (lambda a, n: G.weirdwall_scene(a, n), lambda a: G.sdf_camera(a, -14.0, 20.0, -0.5)),
}


def _box(img, k=8):
    h, w, _ = img.shape
    return img[: h // k * k, : w // k * k].reshape(h // k, k, w // k, k, 3).mean(axis=(1, 3))


@pytest.mark.parametrize("name", sorted(CASES))
def test_converged_image_matches_reference_renderer(name):
    z = np.load(FIXTURE, allow_pickle=False)
    a, b = z[name + "_a"], z[name + "_b"]
    w, h, bounces, spp = (int(x) for x in z[name + "_cfg"])
    alg = api.HostAlgebra()
    mk_scene, mk_cam = CASES[name]
    ctx = api.Context()
    got, _ = ctx.render(ctx.upload(mk_scene(alg, z["noise12"])), mk_cam(alg), w, h, bounces, spp, seed=77)
    ctx.close()
    rmse = lambda x, y: float(np.sqrt(np.mean((x - y) ** 2)))
    floor = rmse(a, b)
    assert floor > 0
    assert rmse(got, a) <= 1.25 * floor, (rmse(got, a), floor)
    assert rmse(got, b) <= 1.25 * floor, (rmse(got, b), floor)
    # The box statistic looks for bias, so the device's own noise is cut with 4x the samples.  Where two
    # reference seeds leave more than the bound in noise themselves, the fixture holds the mean over nine.
    ctx = api.Context()
    fine, _ = ctx.render(ctx.upload(mk_scene(alg, z["noise12"])), mk_cam(alg), w, h, bounces, 4 * spp, seed=78)
    ctx.close()
    ref_mean = _box(z[name + "_m"]) if name + "_m" in z.files else 0.5 * (_box(a) + _box(b))
    rel = np.abs(_box(fine) - ref_mean).mean() / ref_mean.mean()
    assert rel <= 0.02, rel
    assert abs(float(fine.mean()) / float(ref_mean.mean()) - 1.0) <= 0.01


def test_headline_view_converges_to_the_reference_renderers_image():
    """The headline frame's own view -- the demo camera (PathTracerDemo.cpp:897-903, aspect 16 / 9 as the demo sets it on
    resize, :991-993) -- at a quarter of 1920 x 1080 and 4,096 samples per pixel, against the reference's own Renderer
    (fixture demo_view_a / _b: two seeds, each the mean of four 1,024-sample renders by oracle/_ref/libref.so).
    Accepted as the measurement rule says: RMSE(GPU, reference) <= 1.25 x the reference's seed-to-seed RMSE; at four
    times the samples on the device, the 8 x 8 box-filtered images differ by at most 2 % (per lit box on average, and
    over the whole image), and the image means by at most 1 %.  Two reference seeds differ by 0.4 % in that statistic,
    so the 2 % is a bound on bias, not on noise."""
    z = np.load(FIXTURE, allow_pickle=False)
    a, b = z["demo_view_a"], z["demo_view_b"]
    w, h, bounces, spp = (int(x) for x in z["demo_view_cfg"])
    assert (w, h, bounces, spp) == (480, 270, 8, 4096)
    alg = api.HostAlgebra()
    flat, cam = S.demo_scene(alg, z["noise12"]), S.demo_camera(alg, w / h)
    ctx = api.Context()
    ds = ctx.upload(flat)
    got, _ = ctx.render(ds, cam, w, h, bounces, spp, seed=77)
    fine, _ = ctx.render(ds, cam, w, h, bounces, 4 * spp, seed=78)
    ctx.close()
    rmse = lambda x, y: float(np.sqrt(np.mean((x.astype(np.float64) - y) ** 2)))
    floor = rmse(a, b)
    ref = 0.5 * (a.astype(np.float64) + b)
    bf, br, ba, bb = (_box(x.astype(np.float64)).mean(axis=2) for x in (fine, ref, a, b))
    lit = br > 0.05 * br.max()
    per_box = float(np.mean(np.abs(bf[lit] - br[lit]) / br[lit]))
    whole = float(np.abs(_box(fine.astype(np.float64)) - _box(ref)).mean() / _box(ref).mean())
    between_seeds = float(np.mean(np.abs(ba[lit] - bb[lit]) / bb[lit]))
    print(f"headline view {w} x {h}, {spp} spp: RMSE GPU / reference {rmse(got, a):.6f}, {rmse(got, b):.6f}, reference seeds {floor:.6f}; "
          f"box-filtered relative error {per_box:.4f} per lit box ({int(lit.sum())} boxes; {between_seeds:.4f} between the reference's seeds), "
          f"{whole:.4f} over the image; means {fine.mean():.6f} / {ref.mean():.6f}")
    assert floor > 0 and rmse(got, a) <= 1.25 * floor and rmse(got, b) <= 1.25 * floor
    assert per_box <= 0.02 and whole <= 0.02
    assert abs(float(fine.mean()) / float(ref.mean()) - 1.0) <= 0.01


def test_demo_view_at_1080p_against_the_cpu_restatement(oracle):
    """The headline frame's own view (not the close-up), 1920 x 1080, 16 spp on both sides: GPU against the CPU restatement of
    the reference renderer (reference random numbers, two seeds) on the 256 x 256 crop around the ball.  At 16 samples
    this is a noise-level check at the full frame size -- RMSE within 1.25 x the CPU's own seed-to-seed RMSE, the
    box-filtered figure no larger than what two CPU seeds show (x 1.25), crop means within 2 % -- and says nothing about
    bias below a few per cent; the bias bound (2 % box-filtered, 1 % of the mean) is
    test_headline_view_converges_to_the_reference_renderers_image, above, against the reference's own renderer."""
    from oracle import orc
    z = np.load(FIXTURE, allow_pickle=False)
    alg = api.HostAlgebra()
    flat = S.demo_scene(alg, z["noise12"])
    cam = S.demo_camera(alg)
    W, H, bounces, spp = 1920, 1080, 8, 16
    osc = oracle.scene(flat)
    cpu_a, _ = osc.render(cam, W, H, bounces, spp, rng_mode=orc.RNG_MINSTD, seed=1, threads=16, full_width=False)
    cpu_b, _ = osc.render(cam, W, H, bounces, spp, rng_mode=orc.RNG_MINSTD, seed=2, threads=16, full_width=False)
    ctx = api.Context()
    gpu, st = ctx.render(ctx.upload(flat), cam, W, H, bounces, spp, seed=77)
    ctx.close()
    ys, xs = np.nonzero((cpu_a + cpu_b).sum(axis=2) > 0)
    cy, cx = int(round(ys.mean())), int(round(xs.mean()))
    y0, x0 = max(0, min(H - 256, cy - 128)), max(0, min(W - 256, cx - 128))
    crop = (slice(y0, y0 + 256), slice(x0, x0 + 256))
    ga, ca, cb = (x[crop].astype(np.float64) for x in (gpu, cpu_a, cpu_b))
    rmse = lambda p, q: float(np.sqrt(np.mean((p - q) ** 2)))

    def box_rel(p, q):
        bp, bq = _box(p).mean(axis=2), _box(q).mean(axis=2)
        m = bq > 0.05 * bq.max()
        return float(np.mean(np.abs(bp[m] - bq[m]) / bq[m]))

    floor, got = rmse(ca, cb), rmse(ga, ca)
    print(f"demo view 1080p, {spp} spp, crop at ({x0}, {y0}): RMSE GPU-CPU {got:.5f}, CPU-CPU {floor:.5f}; box-filtered relative error "
          f"{box_rel(ga, ca):.4f} against {box_rel(cb, ca):.4f}; means {ga.mean():.5f} / {ca.mean():.5f}")
    assert floor > 0 and got <= 1.25 * floor
    assert box_rel(ga, ca) <= 1.25 * box_rel(cb, ca)
    assert abs(ga.mean() / (0.5 * (ca.mean() + cb.mean())) - 1.0) <= 0.02
    # outside the crop's object nothing is lit on either side
    assert st["paths"] == W * H * spp
