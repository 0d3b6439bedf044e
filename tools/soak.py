"""Soak run: the demo frame (1920x1080, 128 spp, 8 bounces) rendered again and again with changing seeds, then the same seed
again: every frame finite, work counters in a narrow band, the first frame reproduced bit for bit at the end, device memory
in use unchanged.  usage: soak.py [frames]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S   # noqa: E402

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
alg = api.HostAlgebra()
ctx = api.Context()
ds = ctx.upload(S.demo_scene(alg))
cam = S.demo_camera(alg, 1920 / 1080)
first, st0 = ctx.render(ds, cam, 1920, 1080, 8, 128, seed=1)
free0 = torch.cuda.mem_get_info()[0]
t0 = time.perf_counter()
ms, rays, means = [], [], []
for f in range(frames):
    img, st = ctx.render(ds, cam, 1920, 1080, 8, 128, seed=2 + f)
    assert np.isfinite(img).all()
    ms.append(st["ms_device"])
    rays.append(st["rays"])
    means.append(float(img.mean()))
    if f % 100 == 99:
        print(f"frame {f + 1}: {np.mean(ms[-100:]):.2f} ms per frame (min {min(ms[-100:]):.2f}, max {max(ms[-100:]):.2f}), image mean {np.mean(means[-100:]):.5f}", flush=True)
again, st1 = ctx.render(ds, cam, 1920, 1080, 8, 128, seed=1)
free1 = torch.cuda.mem_get_info()[0]
print(f"{frames} frames in {time.perf_counter() - t0:.1f} s; device time {np.mean(ms):.2f} ms per frame (min {min(ms):.2f}, max {max(ms):.2f}); "
      f"rays per frame {min(rays)} .. {max(rays)}; image mean {min(means):.5f} .. {max(means):.5f}; "
      f"first frame reproduced bit for bit: {np.array_equal(first, again) and st0['rays'] == st1['rays']}; free device memory before / after: {free0 >> 20} / {free1 >> 20} MiB")
