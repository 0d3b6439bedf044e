"""Times the demo frame under the march's scheduling switches and checks that the frame and the work counters do not
depend on them.  Each configuration is a set of NAME=VALUE pairs (MIPT_CARRY_LAG, MIPT_CARRY_AFTER,
MIPT_LANE_THRESHOLD, MIPT_NO_FEW); the first one is the base the others are compared with.

    python tools/sweep_carry.py SPP[,SPP...] "MIPT_CARRY_LAG=0" "MIPT_CARRY_LAG=4 MIPT_CARRY_AFTER=32" ...   (on the GPU box)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pathtracer_amd import api, scene as S   # noqa: E402

W, H, BOUNCES = 1920, 1080, 8
SAME = ("paths", "rays", "sdf_rays", "sdf_shades", "sdf_far_evals", "sdf_near_evals", "sdf_noise_calls", "shade_evals")
SWITCHES = ("MIPT_CARRY_LAG", "MIPT_CARRY_AFTER", "MIPT_LANE_THRESHOLD", "MIPT_NO_FEW", "MIPT_CELL_TABLE_BALLS", "MIPT_FULL_MARCH", "MIPT_SORT_MARCHES")


def run(env, spp, frames=4):
    for k in SWITCHES:
        os.environ.pop(k, None)
    os.environ.update(env)
    alg = api.HostAlgebra()
    ctx = api.Context()
    ctx.set_profiling(True)
    ds = ctx.upload(S.demo_scene(alg))
    cam = S.demo_camera(alg, W / H)
    best, img, st, march = 1e30, None, None, 0.0
    for _ in range(frames):
        img, st = ctx.render(ds, cam, W, H, BOUNCES, spp, seed=2024)
        if st["ms_device"] < best:
            best, march = st["ms_device"], st["stage_ms"]["sdf_march"]
    ctx.close()
    return best, march, img, st


def main():
    spps = [int(x) for x in sys.argv[1].split(",")]
    configs = [dict(kv.split("=") for kv in arg.split()) for arg in sys.argv[2:]] or [{"MIPT_CARRY_LAG": "0"}, {}]
    for spp in spps:
        base = None
        for env in configs:
            ms, march, img, st = run(env, spp)
            if base is None:
                base = (img, st)
            same = np.array_equal(img, base[0]) and all(st[k] == base[1][k] for k in SAME)
            label = " ".join(f"{k[5:].lower()}={v}" for k, v in env.items()) or "defaults"
            print(f"spp {spp:5d}  {label:48s} {ms:8.2f} ms  march {march:7.2f}  carried {st['marches_carried']:8d}  identical {same}", flush=True)


if __name__ == "__main__":
    main()
