"""Times a frame rendered by k shards that share GPU 0 (mipt_init_all, simulated mode): k contexts, k streams, k host
threads -- do the stages of one shard (launch tails; memory-bound shading against instruction-bound traversal) overlap with
the others' work?  usage: time_shards.py [demo|random100k|mixed20k] [spp ...]"""
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S   # noqa: E402

alg = api.HostAlgebra()
args = sys.argv[1:]
name = args.pop(0) if args and not args[0].isdigit() else "demo"
if name == "demo":
    flat, cam, w, h, b = S.demo_scene(alg), S.demo_camera(alg, 1920 / 1080), 1920, 1080, 8
elif name == "random100k":
    flat, cam, w, h, b = S.random_scene(alg, 100_000, seed=7), S.random_scene_camera(alg, 1920 / 1080), 1920, 1080, 8
else:
    flat, cam, w, h, b = S.random_scene(alg, 20_000, seed=9, radius_scale=2.0, glass_mirror_mix=True), S.random_scene_camera(alg, 3840 / 2160), 3840, 2160, 16
for spp in [int(a) for a in args] or [128, 1024]:
    for k in (1, 2, 3, 4):
        node = api.Node(devices=[0] * k, simulate=True)
        ns = node.upload(flat)
        node.render(ns, cam, w, h, b, spp, seed=1)
        t0 = time.perf_counter()
        for _ in range(3):
            node.render(ns, cam, w, h, b, spp, seed=1)
        print(f"{name} spp {spp:5d}  shards {k}: {1e3 * (time.perf_counter() - t0) / 3:8.2f} ms per frame", flush=True)
        node.close()
