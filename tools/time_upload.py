"""Scene upload time with the tree built on the host (binned SAH) and on the device (LBVH), and the frame time with each."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from pathtracer_amd import api, capi, scene as S

alg = api.HostAlgebra()
ctx = api.Context()
for n, rs in ((100_000, 1.0), (1_000_000, 0.46)):
    flat = S.random_scene(alg, n, seed=7, radius_scale=rs)
    cam = S.random_scene_camera(alg, 16 / 9)
    for dev in (False, True):
        ts = []
        for _ in range(3):
            t = time.perf_counter()
            ds = ctx.upload(flat, device_bvh=dev)
            ts.append(time.perf_counter() - t)
        _, st = ctx.render(ds, cam, 1280, 720, 8, 16, seed=1)
        _, st = ctx.render(ds, cam, 1280, 720, 8, 16, seed=1)
        print(f"{n} objects, tree on the {'device' if dev else 'host'}: upload {min(ts) * 1e3:.1f} ms, 720p 16 spp frame {st['ms_device']:.1f} ms, info {ds.info()}")
