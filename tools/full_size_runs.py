"""The two configurations that are quoted for several GPUs, at their full size on one GPU: the 1M-primitive scene at
3840x2160, 8 bounces, 4096 spp, and the 20,000 mirrors / glass / emitters at 3840x2160, 16 bounces, 1024 spp.  Each frame is
rendered whole, and again as the eight sample ranges that eight ranks would render (pathtracer_amd/shard.py), summed on the
device as the all-reduce sums them; the two must agree to float rounding of the sums.
usage: full_size_runs.py [random1m|mixed20k ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S, shard   # noqa: E402

alg = api.HostAlgebra()
W, H = 3840, 2160
CASES = {"random1m": (lambda: S.random_scene(alg, 1_000_000, seed=7, radius_scale=0.46), 8, 4096),
         "mixed20k": (lambda: S.random_scene(alg, 20_000, seed=9, radius_scale=2.0, glass_mirror_mix=True), 16, 1024)}
for name in sys.argv[1:] or list(CASES):
    make, bounces, spp = CASES[name]
    ctx = api.Context()
    ds = ctx.upload(make())
    cam = S.random_scene_camera(alg, W / H)
    whole = torch.zeros(H * W * 3, dtype=torch.float32, device="cuda")
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    st = ctx.render_accumulate(ds, cam, W, H, bounces, spp, whole.data_ptr(), seed=2024)
    torch.cuda.synchronize()
    t_whole = time.perf_counter() - t0
    print(f"{name}: {W}x{H}, {bounces} bounces, {spp} spp in one call: {t_whole:.2f} s, {st['paths'] / t_whole / 1e6:.0f} Mpaths/s, "
          f"{st['rays'] / t_whole / 1e9:.2f} Grays/s ({st['rays'] / st['paths']:.2f} rays per path)", flush=True)
    parts = torch.zeros_like(whole)
    t0 = time.perf_counter()
    slowest = 0.0
    for rank in range(8):
        first, count = shard.sample_range(spp, 8, rank)
        t1 = time.perf_counter()
        ctx.render_accumulate(ds, cam, W, H, bounces, count, parts.data_ptr(), seed=2024, first_sample=first)
        torch.cuda.synchronize()
        slowest = max(slowest, time.perf_counter() - t1)
    mean_whole, mean_parts = whole / spp, parts / spp
    diff = (mean_whole - mean_parts).abs()
    rel = diff.max().item() / max(1e-30, mean_whole.abs().max().item())
    print(f"{name}: as eight sample ranges of {spp // 8} spp: slowest range {slowest:.2f} s (what a rank of eight takes before the all-reduce: "
          f"{t_whole / slowest:.2f}x of the whole frame's time); max |difference of the means| {diff.max().item():.3e} "
          f"({rel:.2e} of the brightest value), image mean {mean_whole.mean().item():.6f} / {mean_parts.mean().item():.6f}", flush=True)
    del whole, parts
    ctx.close()
