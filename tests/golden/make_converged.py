"""Regenerates tests/golden/reference_converged.npz: converged images rendered by the reference's own
Renderer (oracle/_ref/libref.so, one worker thread, two seeds each) for the statistical parity tests.
Takes a few minutes (the crater-ball SDF costs ~10 us per near-field evaluation on a CPU).

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
}


def main():
    ref = orc.Reference()
    noise = ref.init_noise(G.NOISE_SEED)
    blob = {"noise12": noise}
    for name, (mk_scene, mk_cam, w, h, bounces, spp) in CASES.items():
        flat, cam = mk_scene(ref, noise), mk_cam(ref)
        sc = ref.scene(flat)
        for tag, seed in (("a", 1001), ("b", 2002)):
            blob[f"{name}_{tag}"] = sc.render(cam, w, h, bounces, spp, seed=seed, threads=1)
            print(name, tag, "mean", float(blob[f"{name}_{tag}"].mean()), flush=True)
        blob[f"{name}_cfg"] = np.array([w, h, bounces, spp], np.int64)
    path = os.path.join(HERE, "reference_converged.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes")


if __name__ == "__main__":
    main()
