"""The multi-rank path on CPU: two gloo ranks each render their block of the global sample range,
the radiance sums are all-reduced, and the result equals the single-rank frame.  The per-rank
renderer is the CPU restatement with the counter-based generator (the same sample addressing the
device path uses); the sharding and reduction code is the product's (pathtracer_amd/shard.py)."""
import os
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
W, H, BOUNCES, SPP, SEED = 24, 20, 6, 7, 5


def _frame(first, count):
    sys.path.insert(0, ROOT)
    from oracle import orc
    from pathtracer_amd import scene as S
    o = orc.Oracle()
    flat, cam = S.cornell_scene(o), S.cornell_camera(o)
    img, _ = o.scene(flat).render(cam, W, H, BOUNCES, count, rng_mode=orc.RNG_COUNTER, seed=SEED, first_sample=first,
                                  threads=1, full_width=True)
    return img


def _worker(rank, world, port, out_dir):
    sys.path.insert(0, ROOT)
    from pathtracer_amd import shard
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    first, count = shard.sample_range(SPP, world, rank)
    total = torch.zeros((H, W, 3), dtype=torch.float32)
    if count:
        total += torch.from_numpy(_frame(first, count)) * float(count)   # the rank's unnormalised sum
    img = shard.reduce_frame(total, SPP, dist)
    np.save(os.path.join(out_dir, f"rank{rank}.npy"), img.numpy())
    dist.destroy_process_group()


def test_sample_range_partitions_exactly():
    from pathtracer_amd import shard
    for spp in (1, 7, 8, 1024, 4096):
        for world in (1, 2, 3, 4, 8):
            blocks = [shard.sample_range(spp, world, r) for r in range(world)]
            assert blocks[0][0] == 0
            for (f0, c0), (f1, _) in zip(blocks, blocks[1:]):
                assert f0 + c0 == f1
            assert blocks[-1][0] + blocks[-1][1] == spp
            assert max(c for _, c in blocks) - min(c for _, c in blocks) <= 1
    with pytest.raises(ValueError):
        shard.sample_range(8, 2, 2)


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_frame_equals_single_rank_frame(world, tmp_path):
    port = 29530 + world
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    want = _frame(0, SPP)
    for r in range(world):
        got = np.load(tmp_path / f"rank{r}.npy")
        # same samples, different order of float additions
        assert np.allclose(got, want, rtol=1e-5, atol=1e-6), r
    assert want.max() > 0
