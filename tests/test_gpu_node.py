"""One process, several GPUs (mipt_init_all / mipt_node_render): the frame's samples are divided among the shards, the
shards' radiance sums are combined -- by one RCCL all-reduce, or, with MIPT_NODE_SIMULATE, by adding the buffers of
several shards that share one GPU -- and the frame is scaled and tone-mapped on the first device.

On a one-GPU box: three simulated shards on GPU 0 equal the single-context frame up to the order of the float sums,
and a one-rank node runs the real RCCL calls (ncclCommInitAll, ncclAllReduce through the dlopen'ed library).
"""
import numpy as np
import pytest

from pathtracer_amd import api, scene as S

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("scene", ["cornell", "demo"])
def test_simulated_shards_equal_the_single_gpu_frame(scene):
    alg = api.HostAlgebra()
    flat, cam = (S.cornell_scene(alg), S.cornell_camera(alg)) if scene == "cornell" else (S.demo_scene(alg), S.demo_camera(alg, 1.0))
    w, h, bounces, spp = (64, 64, 6, 10) if scene == "cornell" else (256, 256, 8, 16)
    ctx = api.Context()
    try:
        want, wst = ctx.render(ctx.upload(flat), cam, w, h, bounces, spp, seed=12)
        want_mapped = ctx.tonemap(1, want)
    finally:
        ctx.close()
    node = api.Node(devices=[0, 0, 0], simulate=True)   # 10 samples -> shards of 4, 3, 3
    try:
        assert node.size() == 3
        ns = node.upload(flat)
        got, st = node.render(ns, cam, w, h, bounces, spp, seed=12)
        assert st["paths"] == wst["paths"] and st["rays"] == wst["rays"] and st["samples_done"] == spp
        assert np.allclose(got, want, rtol=2e-6, atol=1e-7) and want.max() > 0
        mapped, _ = node.render(ns, cam, w, h, bounces, spp, seed=12, tonemap=1)
        assert np.allclose(mapped, want_mapped, rtol=1e-5, atol=1e-6)
        again, _ = node.render(ns, cam, w, h, bounces, spp, seed=12)
        assert np.array_equal(got, again)
    finally:
        node.close()


def test_node_sums_the_counting_builds_fields():
    """mipt_set_counting on every shard's context: the node's statistics carry the summed node visits and primitive
    tests, equal to the single context's (the same rays walk the same tree)."""
    alg = api.HostAlgebra()
    flat, cam = S.random_scene(alg, 3000, seed=5), S.random_scene_camera(alg, 1.0)
    ctx = api.Context()
    try:
        ctx.set_counting(True)
        _, wst = ctx.render(ctx.upload(flat), cam, 64, 64, 4, 6, seed=2)
    finally:
        ctx.close()
    assert wst["node_visits"] > 0 and wst["prim_tests"] > 0
    node = api.Node(devices=[0, 0, 0], simulate=True)
    try:
        for shard in range(node.size()):
            api.check(api.capi.dll().mipt_set_counting(node.ctx_handle(shard), 1), "mipt_set_counting")
        _, st = node.render(node.upload(flat), cam, 64, 64, 4, 6, seed=2)
        assert st["rays"] == wst["rays"]
        for key in ("node_visits", "prim_tests"):   # (a second ray that starts over revisits nodes: scheduling may move these by a hair)
            assert abs(st[key] - wst[key]) <= 0.01 * wst[key], (key, st[key], wst[key])
        assert st["trav_rounds"] > 0 and st["trav_leaf_steps"] > 0
    finally:
        node.close()


def test_one_rank_node_runs_the_rccl_all_reduce():
    alg = api.HostAlgebra()
    flat, cam = S.cornell_scene(alg), S.cornell_camera(alg)
    ctx = api.Context()
    try:
        want, _ = ctx.render(ctx.upload(flat), cam, 48, 48, 6, 8, seed=3)
    finally:
        ctx.close()
    node = api.Node(n=1)
    try:
        assert node.size() == 1
        got, st = node.render(node.upload(flat), cam, 48, 48, 6, 8, seed=3)
        assert np.array_equal(got, want)   # one rank: the all-reduce returns the buffer as it is
        assert st["samples_done"] == 8
    finally:
        node.close()


def test_node_refuses_what_rccl_cannot_do():
    with pytest.raises(api.MiptError):
        api.Node(devices=[0, 0])           # two ranks on one device need the simulated mode
    with pytest.raises(api.MiptError):
        api.Node(devices=[99])
