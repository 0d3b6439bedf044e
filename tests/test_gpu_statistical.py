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
