"""Times the demo frame rendered by k shards that share GPU 0 (mipt_init_all, simulated mode): k contexts, k streams,
k host threads -- do the launch tails of one shard overlap with the others' work?  usage: time_shards.py [spp ...]"""
import sys
import time

import torch  # noqa: F401

sys.path.insert(0, ".")
from pathtracer_amd import api, scene as S   # noqa: E402

alg = api.HostAlgebra()
flat, cam = S.demo_scene(alg), S.demo_camera(alg, 1920 / 1080)
for spp in [int(a) for a in sys.argv[1:]] or [128, 1024]:
    for k in (1, 2, 3, 4):
        node = api.Node(devices=[0] * k, simulate=True)
        ns = node.upload(flat)
        node.render(ns, cam, 1920, 1080, 8, spp, seed=1)
        t0 = time.perf_counter()
        for _ in range(3):
            node.render(ns, cam, 1920, 1080, 8, spp, seed=1)
        print(f"spp {spp:5d}  shards {k}: {1e3 * (time.perf_counter() - t0) / 3:8.2f} ms per frame", flush=True)
        node.close()
