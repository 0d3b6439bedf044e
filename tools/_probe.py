import os, sys
import torch  # noqa
sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S
alg = api.HostAlgebra()
flat = S.random_scene(alg, 100_000, seed=7)
w, h, spp = 1920, 1080, 64
cam = S.random_scene_camera(alg, w / h)
ctx = api.Context()
ctx.set_profiling(True)
sc = ctx.upload(flat)
prev = (0, 0, 0)
for b in (1, 2, 3, 4, 8):
    ctx.render(sc, cam, w, h, b, spp, seed=1)
    img, st = ctx.render(sc, cam, w, h, b, spp, seed=1)
    sh, ex, rays = st["stage_ms"]["shade"], st["stage_ms"]["extend"], st["rays"]
    print("bounces", b, "rays", rays, "shade ms", round(sh, 2), "extend ms", round(ex, 2), "generate", round(st["stage_ms"]["generate"], 2),
          "| this step: rays", rays - prev[2], "shade", round(sh - prev[0], 2), "extend", round(ex - prev[1], 2), flush=True)
    prev = (sh, ex, rays)
