"""What does a leaf step of mixed kinds cost?  The 100k-object scene as it is (spheres and boxes alternate), with every
object a sphere, and with every object a box: frame and stage times, and node visits / primitive tests per ray from the
counting build.
usage: time_kinds.py [n_objects] [spp]"""
import dataclasses
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S   # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 100_000
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 64
alg = api.HostAlgebra()
flat = S.random_scene(alg, n, seed=7)
cam = S.random_scene_camera(alg, 1920 / 1080)
for label, kind in (("mixed", None), ("spheres", S.SPHERE), ("boxes", S.BOX)):
    f = flat if kind is None else dataclasses.replace(flat, kind=np.full_like(flat.kind, kind))
    ctx = api.Context()
    ctx.set_profiling(True)
    sc = ctx.upload(f)
    ctx.render(sc, cam, 1920, 1080, 8, spp, seed=1)
    img, st = ctx.render(sc, cam, 1920, 1080, 8, spp, seed=1)
    ctx.set_counting(True)
    _, c = ctx.render(sc, cam, 1920, 1080, 8, max(1, spp // 8), seed=1)
    print(f"{label:8s} {st['ms_device']:8.2f} ms  generate {st['stage_ms']['generate']:7.2f} extend {st['stage_ms']['extend']:7.2f} shade {st['stage_ms']['shade']:7.2f}"
          f"  rays {st['rays'] / 1e6:8.1f} M  ns/ray in extend {1e6 * st['stage_ms']['extend'] / st['rays']:.4f}"
          f"  visits {c['node_visits'] / c['rays']:.2f} tests {c['prim_tests'] / c['rays']:.2f}", flush=True)
    del sc, ctx
