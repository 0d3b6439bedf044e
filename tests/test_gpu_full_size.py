"""The headline configurations at their full sizes, checked through properties that do not need a CPU
render of the same size (the CPU restatement takes minutes per 1080p frame at these sample counts):

  * linearity in the samples -- the frame of samples [0, n) is the mean of the frames of [0, n/2) and
    [n/2, n): what sharding over two GPUs computes, here at full size;
  * reproducibility -- the same frame twice is bitwise the same, counters included;
  * resolution independence -- the light the ball sends to the camera, integrated over the frame, is the
    same at 1920 x 1080 and at 480 x 270 (different pixels, different samples, same estimator);
  * accounting -- paths = W H spp, rays per path in the range the scene implies.
"""
import numpy as np
import pytest

from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = api.Context()
    yield c
    c.close()


def test_demo_scene_1080p_1024spp(ctx):
    alg = api.HostAlgebra()
    W, H, SPP, B = 1920, 1080, 1024, 8
    flat, cam = S.demo_scene(alg), S.demo_camera(alg, W / H)
    ds = ctx.upload(flat)
    full, st = ctx.render(ds, cam, W, H, B, SPP, seed=2024)
    assert st["paths"] == W * H * SPP
    assert 1.015 * st["paths"] < st["rays"] < 1.03 * st["paths"]      # ~1.2 % of the pixels see the ball
    assert st["sdf_near_evals"] > 4 * st["sdf_rays"] > 0
    # only the pixels inside the screen bounds of the ball's near field (a sphere of radius 0.6) trace primary rays
    # (the slab is above the frame): the ball covers 1.2 % of the frame, its bounds about 3 %
    assert 0.02 * st["paths"] < st["primary_traced"] < 0.05 * st["paths"]
    again, st2 = ctx.render(ds, cam, W, H, B, SPP, seed=2024)
    assert np.array_equal(full, again)
    assert all(st[k] == st2[k] for k in ("rays", "sdf_rays", "sdf_shades", "sdf_near_evals", "sdf_noise_calls"))
    a, _ = ctx.render(ds, cam, W, H, B, SPP // 2, seed=2024, first_sample=0)
    b, _ = ctx.render(ds, cam, W, H, B, SPP // 2, seed=2024, first_sample=SPP // 2)
    assert np.allclose(full, 0.5 * (a + b), rtol=2e-5, atol=1e-6)
    # the ball's image: non-black pixels form one blob of about 1.2 % of the frame around the centre
    lit = full.sum(axis=2) > 0
    assert 0.008 < lit.mean() < 0.02
    ys, xs = np.nonzero(lit)
    # the ball's centre (0, 0.5, -2) seen from (0, 0, -30) with a 10 degree half angle across the width
    y_centre = H / 2 * (1.0 + 0.5 / (28.0 * np.tan(np.radians(10.0)) / (W / H)))
    assert abs(xs.mean() - W / 2) < 0.01 * W and abs(ys.mean() - y_centre) < 0.01 * H
    # resolution independence of the integrated radiance (per unit of frame area)
    small, _ = ctx.render(ds, cam, W // 4, H // 4, B, 4096, seed=7)
    assert abs(full.mean() - small.mean()) <= 0.02 * small.mean()


def test_100k_primitives_1080p_256spp(ctx):
    alg = api.HostAlgebra()
    W, H, SPP, B = 1920, 1080, 256, 8
    flat, cam = S.random_scene(alg, 100_000, seed=7), S.random_scene_camera(alg, W / H)
    ds = ctx.upload(flat)
    full, st = ctx.render(ds, cam, W, H, B, SPP, seed=3)
    assert st["paths"] == W * H * SPP and st["rays"] > 3 * st["paths"]
    a, _ = ctx.render(ds, cam, W, H, B, SPP // 2, seed=3, first_sample=0)
    b, _ = ctx.render(ds, cam, W, H, B, SPP // 2, seed=3, first_sample=SPP // 2)
    assert np.allclose(full, 0.5 * (a + b), rtol=2e-5, atol=1e-6)
    small, _ = ctx.render(ds, cam, W // 4, H // 4, B, 1024, seed=11)
    assert abs(full.mean() - small.mean()) <= 0.02 * small.mean()
