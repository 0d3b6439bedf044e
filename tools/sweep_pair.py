"""Two rays per lane against one: frame time and stage times of an analytic scene for several settings of the leaf-step
quorum / turn-over threshold; every frame is compared bit for bit with the one-ray frame.
usage: sweep_pair.py [scene] [spp]   (scene: random100k | random1m | mixed20k)"""
import os
import sys

import numpy as np
import torch  # noqa: F401

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S   # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "random100k"
spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
alg = api.HostAlgebra()
if name == "random100k":
    flat, w, h, b = S.random_scene(alg, 100_000, seed=7), 1920, 1080, 8
elif name == "random1m":
    flat, w, h, b = S.random_scene(alg, 1_000_000, seed=7, radius_scale=0.46), 3840, 2160, 8
else:
    flat, w, h, b = S.random_scene(alg, 20_000, seed=9, radius_scale=2.0, glass_mirror_mix=True), 3840, 2160, 16
cam = S.random_scene_camera(alg, w / h)
ref = None
settings = [(0, 0)] + [(q, k) for q in (32, 40, 48) for k in (88, 100, 112)]
if len(sys.argv) > 3:
    settings = [(0, 0)] + [tuple(int(v) for v in a.split(",")) for a in sys.argv[3:]]
for q, k in settings:
    os.environ["MIPT_PAIR_QUORUM"], os.environ["MIPT_PAIR_KEEP"] = str(q), str(k)
    ctx = api.Context()
    sc = ctx.upload(flat)
    ctx.render(sc, cam, w, h, b, spp, seed=1)
    img, st = ctx.render(sc, cam, w, h, b, spp, seed=1)
    same = "ref" if ref is None else ("same bits" if np.array_equal(img, ref) else "DIFFERENT")
    if ref is None:
        ref = img
    print(f"{name} spp {spp} quorum {q:2d} keep {k:3d}: {st['ms_device']:8.2f} ms  generate {st['stage_ms']['generate']:7.2f} extend {st['stage_ms']['extend']:7.2f} "
          f"shade {st['stage_ms']['shade']:7.2f}  rays {st['rays']}  {same}", flush=True)
    del sc, ctx
