"""Camera rays through the fused generate + closest-hit kernel against camera rays written out and traced by the
two-rays-per-lane kernel (MIPT_SPLIT_PRIMARY): frame time, per-stage times, and that the frames are identical.

    python tools/time_split_primary.py [scene] [spp] [width height]      (on the GPU box)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scene as S   # noqa: E402


def main():
    name = sys.argv[1] if len(sys.argv) > 1 else "random100k"
    spp = int(sys.argv[2]) if len(sys.argv) > 2 else 128
    w, h = (int(sys.argv[3]), int(sys.argv[4])) if len(sys.argv) > 4 else (1920, 1080)
    alg = api.HostAlgebra()
    bounces = 8
    if name == "random100k":
        flat = S.random_scene(alg, 100_000, seed=7)
    elif name == "random1m":
        flat = S.random_scene(alg, 1_000_000, seed=7, radius_scale=0.46)
    elif name == "mixed20k":
        flat, bounces = S.random_scene(alg, 20_000, seed=9, radius_scale=2.0, glass_mirror_mix=True), 16
    else:
        flat = S.demo_scene(alg)
    cam = S.demo_camera(alg, w / h) if name == "demo" else S.random_scene_camera(alg, w / h)
    frames = {}
    for mode in ("0", "1"):
        os.environ["MIPT_SPLIT_PRIMARY"] = mode
        ctx = api.Context()
        ctx.set_profiling(True)
        ds = ctx.upload(flat)
        best = None
        for _ in range(3):
            img, st = ctx.render(ds, cam, w, h, bounces, spp, seed=2024)
            if best is None or st["ms_device"] < best["ms_device"]:
                best = st
        frames[mode] = (img, best)
        print(f"{name} {w}x{h} {spp} spp, split primary {mode}: {best['ms_device']:.1f} ms, stages "
              f"{ {k: round(v, 1) for k, v in best['stage_ms'].items() if v} }, rays {best['rays']}", flush=True)
        ctx.close()
    same = np.array_equal(frames["0"][0], frames["1"][0]) and frames["0"][1]["rays"] == frames["1"][1]["rays"]
    print("identical frames and ray counts:", same)


if __name__ == "__main__":
    main()
