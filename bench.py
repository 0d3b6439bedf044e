#!/usr/bin/env python3
"""Headline benchmark: the demo scene (crater-ball SDF + emissive slab) at 1920x1080, 1024 spp,
8 bounces, rendered by the HIP wavefront path through the C ABI of libmipt.so.

    python bench.py --gpus N --steps K --warmup W

One "step" is one full frame: every rank renders its block of the frame's samples per pixel into a device
radiance-sum buffer, the buffers are summed with one RCCL all-reduce, scaled by 1/spp, tone-mapped on
the GPU and copied to the host on rank 0.  Samples are addressed by their global index.  By default the
samples scale with the GPUs (--spp per GPU, N x spp per frame: weak scaling, the work per GPU is fixed);
--scaling strong keeps the frame at --spp samples and divides them among the ranks.

Rank 0 prints one JSON line.  `value` is Mpaths/s over the whole job; `roofline` prices the dominant
kernel (the SDF march: bound by vector instruction issue, almost no memory traffic) from HIP events recorded around
each of its launches inside the timed region; `cpu_baseline` is the CPU restatement of the reference
renderer (oracle/, minstd_rand0 sampling, 8-pixel tasks) on a bounded sample of the same workload.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

from pathtracer_amd import api, scene as S, shard   # noqa: E402

# Algorithmic flop model of CraterBallObject::signedDistance (PathTracerDemo.cpp:361-466), counted on
# the reference's arithmetic: a far-field evaluation is a norm and a compare; the near-field body
# visits 64 + 8 + 8 hashed cells; noise() is 8 octaves of an affine transform and three triangle waves.
FLOP_FAR_EVAL = 12
FLOP_NEAR_BODY = 64 * 80 + 16 * 60
FLOP_NOISE_CALL = 8 * 47
PEAK_FP32_VECTOR_TFLOPS = 157.3
PEAK_HBM_GBS = 8000.0


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=3)
    p.add_argument("--warmup", type=int, default=1)
    p.add_argument("--width", type=int, default=1920)
    p.add_argument("--height", type=int, default=1080)
    p.add_argument("--spp", type=int, default=1024, help="samples per pixel: per GPU (weak scaling) or per frame (strong)")
    p.add_argument("--scaling", default="weak", choices=["weak", "strong"])
    p.add_argument("--bounces", type=int, default=8)
    p.add_argument("--scene", default="demo", choices=["demo", "random100k", "random1m", "mixed20k"])
    p.add_argument("--no-cpu-baseline", action="store_true")
    p.add_argument("--cpu-spp", type=int, default=64, help="samples per pixel of the CPU baseline's bounded sample")
    p.add_argument("--rehearse", action="store_true",
                   help="every rank uses cuda:0 and the all-reduce runs over gloo: checks the multi-process path on a one-GPU box, "
                        "measures nothing (the JSON line says so)")
    return p.parse_args()


def build_scene(name, alg, aspect):
    if name == "demo":
        return S.demo_scene(alg), S.demo_camera(alg, aspect), "demo scene (CraterBallObject + emissive BoxObject)"
    if name == "random100k":
        return (S.random_scene(alg, 100_000, seed=7), S.random_scene_camera(alg, aspect),
                "100k random spheres/boxes, diffuse+specular")
    if name == "random1m":
        return (S.random_scene(alg, 1_000_000, seed=7, radius_scale=0.46), S.random_scene_camera(alg, aspect),
                "1M random spheres/boxes, diffuse+specular")
    return (S.random_scene(alg, 20_000, seed=9, radius_scale=2.0, glass_mirror_mix=True), S.random_scene_camera(alg, aspect),
            "20k mirrors/glass/emitters")


def main():
    a = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs an MI355X: torch.cuda.is_available() is False and there is no CPU fallback")
    if a.rehearse:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist   # RCCL behind the "nccl" backend
        if a.rehearse:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    if a.gpus != world:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run --nproc-per-node {a.gpus}")

    alg = api.HostAlgebra()
    W, H = a.width, a.height
    flat, cam, scene_label = build_scene(a.scene, alg, W / H)
    # one explicit stream carries torch's own ops and the library's kernels (the null stream would make
    # the library fall back to a private stream, unordered with torch's)
    stream = torch.cuda.Stream()
    torch.cuda.set_stream(stream)
    ctx = api.Context(device=local_rank, stream=stream.cuda_stream)
    t_up = time.perf_counter()
    dscene = ctx.upload(flat)
    upload_s = time.perf_counter() - t_up
    ctx.set_profiling(True)

    # contiguous block of the frame's global sample range per rank, remainder to the low ranks
    frame_spp = a.spp * world if a.scaling == "weak" else a.spp
    my_first, my_spp = shard.sample_range(frame_spp, world, rank)

    d_sum = torch.zeros((H, W, 3), dtype=torch.float32, device="cuda")
    d_ldr = torch.empty_like(d_sum)
    h_out = torch.empty((H, W, 3), dtype=torch.float32).pin_memory() if rank == 0 else None

    agg = {}

    def step(record):
        d_sum.zero_()
        st = None
        if my_spp:
            st = ctx.render_accumulate(dscene, cam, W, H, a.bounces, my_spp, d_sum.data_ptr(), seed=2024, first_sample=my_first)
        shard.reduce_frame(d_sum, frame_spp, dist)   # RCCL all-reduce + 1/spp
        ctx.tonemap_device(1, d_sum.data_ptr(), d_ldr.data_ptr(), W, H)
        if rank == 0:
            h_out.copy_(d_ldr, non_blocking=True)
        if record and st:
            for k, v in st.items():
                if isinstance(v, dict):
                    d = agg.setdefault(k, {})
                    for kk, vv in v.items():
                        d[kk] = d.get(kk, 0) + vv
                else:
                    agg[k] = agg.get(k, 0) + v

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(a.warmup):
        step(False)
    fence()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device="cuda")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        # whole-job ray count
        r = torch.tensor([float(agg.get("rays", 0))], dtype=torch.float64, device="cuda")
        dist.all_reduce(r)
        total_rays = float(r.item())
    else:
        total_rays = float(agg.get("rays", 0))

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    total_paths = float(W) * H * frame_spp * a.steps
    ms_per_step = 1e3 * elapsed / a.steps
    out = {
        "metric": "Mpaths/s at 1080p (demo scene, 1024 spp, 8 bounces)" if (a.scene == "demo" and (W, H, a.spp) == (1920, 1080, 1024))
                  else f"Mpaths/s ({a.scene}, {W}x{H}, {a.spp} spp, {a.bounces} bounces)",
        "value": total_paths / elapsed / 1e6,
        "unit": "Mpaths/s",
        "mrays_per_s": total_rays / elapsed / 1e6,
        "rays_per_path": total_rays / total_paths,
        "n_gpus": world,
        "steps": a.steps,
        "warmup": a.warmup,
        "ms_per_step": ms_per_step,
        "higher_is_better": True,
        "scaling": a.scaling,
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic" if not a.rehearse else "synthetic; REHEARSAL: all ranks on one GPU over gloo, not a measurement",
        "config": {"workload": f"{scene_label}, {W}x{H}, {frame_spp} spp per frame ({my_spp} per GPU), {a.bounces} bounces, filmic tone map on GPU",
                   "primitives": flat.n_prims, "parallelism": f"spp-shard x{world} + RCCL all-reduce" if world > 1 else "single GPU",
                   "scene_upload_s": round(upload_s, 4)},
    }

    # ---- roofline of the dominant kernel, from rank 0's event-timed launches -------------------
    stage_ms = agg.get("stage_ms", {})
    launches = agg.get("stage_launches", {})
    out["stage_ms_per_step"] = {k: round(v / a.steps, 3) for k, v in stage_ms.items()}
    if "primary_traced" in agg and world == 1:
        # closest-hit queries that cost a ray: primaries inside the objects' screen bounds plus every later cast
        traced = agg["primary_traced"] + agg.get("rays", 0) - agg.get("paths", 0)
        out["mrays_traced_per_s"] = traced / elapsed / 1e6
    out["work_per_step"] = {k: agg[k] // a.steps for k in ("paths", "rays", "sdf_rays", "sdf_shades", "sdf_far_evals", "sdf_near_evals",
                                                            "sdf_noise_calls", "shade_evals", "shade_noise_calls", "primary_traced") if k in agg}
    dominant = max(stage_ms, key=stage_ms.get) if stage_ms else None
    if dominant == "sdf_march":
        flops = (agg["sdf_far_evals"] * FLOP_FAR_EVAL + agg["sdf_near_evals"] * FLOP_NEAR_BODY + agg["sdf_noise_calls"] * FLOP_NOISE_CALL)
        n_launch = max(1, launches["sdf_march"])
        avg_ms = stage_ms["sdf_march"] / n_launch
        achieved = flops / n_launch / (avg_ms * 1e-3) / 1e12
        # HBM bytes per launch of this kernel: a process cannot collect counters on itself, so the figure comes from
        # the committed counter passes of this very command (FETCH_SIZE and WRITE_SIZE in separate rocprofv3 runs),
        # and only when the workload is the one they were taken on
        traffic, traffic_source = None, None
        hbm_csv = os.path.join(ROOT, "profiles", "r01_pmc_hbm_demo1080p_1024spp.csv")
        if a.scene == "demo" and (W, H, a.spp, a.bounces, world) == (1920, 1080, 1024, 8, 1) and os.path.exists(hbm_csv):
            for line in open(hbm_csv):
                if "k_sdf_march" in line:
                    cols = line.strip().rsplit(",", 3)
                    traffic = (float(cols[2]) + float(cols[3])) * 1024.0 / max(1.0, float(cols[1]))
                    traffic_source = "profiles/r01_pmc_hbm_demo1080p_1024spp.csv (raw FETCH_SIZE + WRITE_SIZE, KB, per launch)"
        out["roofline"] = {"kernel": "k_sdf_march", "bound": "mfma", "bound_detail": "compute side of the roofline: fp32 vector issue, with noise()'s octave transforms on the f32 matrix core",
                           "achieved": achieved, "peak": PEAK_FP32_VECTOR_TFLOPS,
                           "unit": "TFLOP/s", "frac": achieved / PEAK_FP32_VECTOR_TFLOPS, "traffic": traffic, "traffic_source": traffic_source,
                           "avg_launch_ms": avg_ms, "launches": n_launch,
                           "note": "fp32 vector peak; the octave transforms of noise() run on the f32 matrix core, whose peak is the same 157.3; "
                                   "flops = far*12 + near*6080 + noise*376 (the reference's arithmetic)"}
    elif dominant is not None:
        # analytic scenes: traversal is a gather of 64-byte nodes and 128-byte primitive lines
        n_launch = max(1, launches[dominant])
        avg_ms = stage_ms[dominant] / n_launch
        bytes_alg = agg.get("rays", 0) * (32 + 24) if dominant == "extend" else agg.get("paths", 0) * 64
        achieved = bytes_alg / n_launch / (avg_ms * 1e-3) / 1e9
        out["roofline"] = {"kernel": "k_" + dominant, "bound": "hbm", "achieved": achieved, "peak": PEAK_HBM_GBS, "unit": "GB/s",
                           "frac": achieved / PEAK_HBM_GBS, "traffic": None, "avg_launch_ms": avg_ms, "launches": n_launch}

    # ---- CPU baseline: the restated reference renderer on a bounded sample ----------------------
    if not a.no_cpu_baseline and world == 1 and flat.n_prims > 5000:
        # The reference's closest hit tests every object for every ray (its tree never culls), so its cost per ray is
        # proportional to the number of objects: time 2,000 objects of the same generator on a 128 x 128 frame and
        # report both that figure and its extrapolation to the scene's size.
        from oracle import orc   # test infrastructure, used here only as the timed baseline
        cores = min(16, os.cpu_count() or 1)
        n_sub = 2000
        kw = dict(seed=9, radius_scale=2.0, glass_mirror_mix=True) if a.scene == "mixed20k" else dict(seed=7, radius_scale=0.46 if a.scene == "random1m" else 1.0)
        sub = S.random_scene(alg, n_sub, **kw)
        osc = orc.Oracle().scene(sub)
        t1 = time.perf_counter()
        _, cst = osc.render(S.random_scene_camera(alg, 1.0), 128, 128, a.bounces, a.cpu_spp, rng_mode=orc.RNG_MINSTD, seed=1, threads=cores, full_width=False)
        cpu_s = time.perf_counter() - t1
        raw = cst["paths"] / cpu_s / 1e6
        out["cpu_baseline"] = {"value": raw * n_sub / flat.n_prims, "unit": "Mpaths/s", "cores": cores, "kind": "port",
                               "sample": f"{n_sub} objects of the same generator, 128x128, {a.cpu_spp} spp ({cst['paths']} paths, {cpu_s:.1f} s): "
                                         f"{raw:.4f} Mpaths/s measured, scaled by {n_sub}/{flat.n_prims} (cost per ray is linear in the object count)",
                               "measured_value": raw}
    elif not a.no_cpu_baseline and world == 1:
        from oracle import orc   # test infrastructure, used here only as the timed baseline
        cores = min(16, os.cpu_count() or 1)
        osc = orc.Oracle().scene(flat)
        t1 = time.perf_counter()
        _, cst = osc.render(cam, W, H, a.bounces, a.cpu_spp, rng_mode=orc.RNG_MINSTD, seed=1, threads=cores, full_width=False)
        cpu_s = time.perf_counter() - t1
        out["cpu_baseline"] = {"value": cst["paths"] / cpu_s / 1e6, "unit": "Mpaths/s", "cores": cores, "kind": "port",
                               "sample": f"same scene and camera, {W}x{H}, {a.cpu_spp} spp ({cst['paths']} paths, {cpu_s:.1f} s)",
                               "mrays_per_s": cst["rays"] / cpu_s / 1e6}
        if orc.fast_lib() is not None:
            # the same restatement compiled with -ffast-math (the reference's MSVC build sets /fp:fast), on half the sample
            fspp = max(1, a.cpu_spp // 2)
            fsc = orc.Oracle(fast_math=True).scene(flat)
            t2 = time.perf_counter()
            _, fst = fsc.render(cam, W, H, a.bounces, fspp, rng_mode=orc.RNG_MINSTD, seed=1, threads=cores, full_width=False)
            fast_s = time.perf_counter() - t2
            out["cpu_baseline"]["fast_math_value"] = fst["paths"] / fast_s / 1e6
            out["cpu_baseline"]["fast_math_sample"] = f"{fspp} spp ({fast_s:.1f} s), g++ -O3 -ffast-math"
    print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
