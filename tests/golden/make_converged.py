"""Regenerates tests/golden/reference_converged.npz: converged images rendered by the reference's own
Renderer (oracle/_ref/libref.so, one worker thread per image, two seeds each) for the statistical parity
tests, plus the mean over nine seeds (<name>_m, <name>_n) of the scenes whose image mean two seeds do not settle.
Takes about a quarter of an hour on 8 cores from nothing; cases already in the file are kept unless --all is given.

    make -C oracle ref && python tests/golden/make_converged.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import golden_cases as G                      # noqa: E402
from oracle import orc                        # noqa: E402
from pathtracer_amd import scene as S         # noqa: E402

CASES = {
    # name: (scene builder, camera builder, width, height, bounces, spp)
    "cornell": (lambda a, n: S.cornell_scene(a), lambda a: S.cornell_camera(a), 32, 32, 8, 2048),
    "mixed": (lambda a, n: G.mixed_scene(a), lambda a: G.parity_camera(a, 4.0 / 3.0), 32, 24, 12, 2048),
    "demo_close": (lambda a, n: S.demo_scene(a, n), lambda a: G.demo_closeup_camera(a), 32, 32, 8, 384),
    # the demo's other SDF classes, with their procedural materials
    "roughbox": (lambda a, n: G.roughbox_scene(a, n), lambda a: G.roughbox_camera(a), 24, 24, 5, 1024),
    "roughsphere": (lambda a, n: G.roughsphere_scene(a, n), lambda a: G.sdf_camera(a, -8.0, 14.0), 24, 24, 5, 1024),
    "weirdwall": (lambda a, n: G.weirdwall_scene(a, n), lambda a: G.sdf_camera(a, -14.0, 20.0, -0.5), 24, 24, 5, 768),
    # the headline frame's own view (PathTracerDemo.cpp:897-903 with the aspect the demo sets on resize, :991-993) at a
    # quarter of 1920 x 1080
    "demo_view": (lambda a, n: S.demo_scene(a, n), lambda a: S.demo_camera(a, 16.0 / 9.0), 480, 270, 8, 4096),
}

# Cases whose images are means over that many independent renders of spp / parts samples each (one worker thread per
# render keeps every part reproducible; the parts of seed s use the seeds 10 s + part).
PARTS = {"demo_view": 4}


EXTRA_SEEDS = (11, 12, 13, 14, 15, 16, 17)     # with a and b: nine images behind every <name>_m
NOISY = ("roughbox", "roughsphere", "weirdwall")  # two seeds alone leave ~1 % of noise in these scenes' image means


def _render(job):
    name, seed, part = job
    ref = orc.Reference()
    noise = ref.init_noise(G.NOISE_SEED)
    mk_scene, mk_cam, w, h, bounces, spp = CASES[name]
    parts = PARTS.get(name, 1)
    run_seed = seed if parts == 1 else 10 * seed + part
    return name, seed, ref.scene(mk_scene(ref, noise)).render(mk_cam(ref), w, h, bounces, spp // parts, seed=run_seed, threads=1)


def main():
    import multiprocessing as mp
    ref = orc.Reference()
    noise = ref.init_noise(G.NOISE_SEED)
    path = os.path.join(HERE, "reference_converged.npz")
    blob = {"noise12": noise}
    if os.path.exists(path) and "--all" not in sys.argv:   # keep what is there, render only the cases that are missing
        old = np.load(path, allow_pickle=False)
        assert np.array_equal(old["noise12"], noise)
        blob.update({k: old[k] for k in old.files})
    jobs = [(name, seed, part) for name in CASES if name + "_cfg" not in blob for seed in (1001, 2002)
            for part in range(PARTS.get(name, 1))]
    jobs += [(name, seed, 0) for name in NOISY if name + "_m" not in blob for seed in EXTRA_SEEDS]
    extra = {name: [] for name in NOISY}
    sums = {}
    with mp.get_context("spawn").Pool(min(8, os.cpu_count() or 1)) as pool:
        for name, seed, img in pool.imap_unordered(_render, jobs):
            print(name, seed, "mean", float(img.mean()), flush=True)
            if seed in EXTRA_SEEDS:
                extra[name].append((seed, img))
            else:
                acc = sums.setdefault((name, seed), [np.zeros(img.shape, np.float64), 0])
                acc[0] += img
                acc[1] += 1
    for (name, seed), (total, count) in sums.items():
        assert count == PARTS.get(name, 1)
        blob[f"{name}_{'a' if seed == 1001 else 'b'}"] = (total / count).astype(np.float32)
    for name, (_, _, w, h, bounces, spp) in CASES.items():
        blob[f"{name}_cfg"] = np.array([w, h, bounces, spp], np.int64)
        if extra.get(name):
            imgs = [blob[name + "_a"], blob[name + "_b"]] + [img for _, img in sorted(extra[name], key=lambda e: e[0])]
            blob[name + "_m"] = np.mean(np.stack(imgs).astype(np.float64), axis=0).astype(np.float32)
            blob[name + "_n"] = np.array(len(imgs), np.int64)
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
