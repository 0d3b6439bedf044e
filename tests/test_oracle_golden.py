"""The CPU restatement (oracle/liboracle.so) against outputs captured from the reference's own code.

tests/golden/reference_cases.npz was written by tests/golden/make_golden.py from
oracle/_ref/libref.so (the reference sources compiled in place).  Every case must come out
bit-identical: the restatement keeps the reference's operation order and is built without fused
multiply-add.  The reference ships no tests of its own, so these fixtures are what pins the oracle.
"""
import os

import numpy as np
import pytest

import golden_cases as G

FIXTURE = os.path.join(os.path.dirname(__file__), "golden", "reference_cases.npz")


@pytest.fixture(scope="module")
def golden():
    z = np.load(FIXTURE, allow_pickle=False)
    inp = {k[3:]: z[k] for k in z.files if k.startswith("in_")}
    out = {k[4:]: z[k] for k in z.files if k.startswith("out_")}
    return inp, out


@pytest.fixture(scope="module")
def oracle_results(oracle, golden):
    inp, out = golden
    return G.evaluate(G.OracleBackend(oracle, out["noise12"]), inp)


def test_inputs_are_reproducible(golden):
    """The stored inputs are what make_inputs() generates today (fixtures are not stale)."""
    inp, _ = golden
    now = G.make_inputs()
    assert sorted(now) == sorted(inp)
    for k in now:
        assert np.array_equal(now[k], inp[k]), k


CASES = sorted(k[4:] for k in np.load(FIXTURE, allow_pickle=False).files if k.startswith("out_"))


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_bitwise(name, golden, oracle_results):
    want, got = golden[1][name], oracle_results[name]
    assert want.shape == got.shape
    assert np.array_equal(want, got, equal_nan=True), (
        f"{name}: {np.sum(want != got)} of {want.size} values differ from the reference")


def test_fixture_coverage(golden):
    """The fixtures exercise what they claim to: hits and misses of every kind, rejected inverses,
    the total-internal-reflection branch, SDF hits, and non-black rendered images."""
    _, out = golden
    for k in ("sphere_hit", "box_hit", "tri_hit", "sdf_hit", "inv_ok"):
        assert 0 < out[k].sum() < out[k].size, k
    assert (out["deflect"][4][:, 1] == 1.0).sum() > 10          # cyan emission: ior > 2 inside hit
    assert (out["demo_close_prim"] == 0).mean() > 0.4           # the ball fills the close-up
    for k in ("sb_render", "mx_render", "cb_render", "demo_render", "demo_close_render"):
        assert out[k].max() > 0, k
    assert len(np.unique(out["sb_primary_prim"])) > 100
