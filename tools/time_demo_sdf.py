"""Times one frame of each lane-evaluated demo SDF scene (rough box, rough sphere, weird wall) on cuda:0."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import golden_cases as G   # noqa: E402
from pathtracer_amd import api   # noqa: E402

alg = api.HostAlgebra()
ctx = api.Context(device=0)
noise = np.load(os.path.join(ROOT, "tests", "golden", "reference_cases.npz"))["out_noise12"]
w = h = 512
for name, flat, cam in (("roughbox", G.roughbox_scene(alg, noise), G.roughbox_camera(alg)),
                        ("roughsphere", G.roughsphere_scene(alg, noise), G.sdf_camera(alg, -8.0, 14.0)),
                        ("weirdwall", G.weirdwall_scene(alg, noise), G.sdf_camera(alg, -14.0, 20.0, -0.5))):
    ds = ctx.upload(flat)
    ctx.render(ds, cam, w, h, 8, 4, seed=1)
    t0 = time.perf_counter()
    _, st = ctx.render(ds, cam, w, h, 8, 64, seed=2)
    dt = time.perf_counter() - t0
    print(f"{name}: {w}x{h} 64 spp 8 bounces {dt * 1e3:.1f} ms, {w * h * 64 / dt / 1e6:.1f} Mpaths/s, "
          f"{st['rays'] / dt / 1e6:.1f} Mrays/s, sdf rays {st['sdf_rays']}, shades {st['sdf_shades']}", flush=True)
