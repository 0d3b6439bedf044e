"""The four-child tree the traversal kernels walk, checked on the host (no GPU): tests/cpp/bvh4_check.cpp, built here with
g++ against pathtracer_amd/csrc/bvh.cpp, walks the tree collapse_bvh4 makes and verifies that every primitive is a leaf
exactly once with its own box, that children lie inside the box their parent holds for them, that absent children can
never be hit, and that the reported traversal-stack need covers a walk that defers every child it can.

The tree replaces ObjectTree::Tree::makeTree (src/ObjectTree.cpp:90-184); because the reference's tree never culls, any
tree over the same primitives returns the same closest hit, which the GPU parity tests check bit for bit."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "pathtracer_amd", "csrc")


@pytest.fixture(scope="module")
def checker(tmp_path_factory):
    exe = str(tmp_path_factory.mktemp("bvh4") / "bvh4_check")
    subprocess.run(["g++", "-O2", "-std=c++17", "-Wall", "-I", CSRC, "-o", exe, os.path.join(ROOT, "tests", "cpp", "bvh4_check.cpp"),
                    os.path.join(CSRC, "bvh.cpp")], check=True)
    return exe


@pytest.mark.parametrize("n,seed,layout", [(0, 1, 0), (1, 1, 0), (2, 1, 0), (3, 2, 0), (4, 3, 0), (5, 4, 0), (7, 5, 0), (64, 6, 0), (1000, 7, 0),
                                           (20_000, 8, 0), (200_000, 9, 0), (5000, 10, 1), (5000, 11, 2)])
def test_collapse_invariants(checker, n, seed, layout):
    out = subprocess.run([checker, str(n), str(seed), str(layout)], capture_output=True, text=True, timeout=120)
    print(out.stdout.strip())
    assert out.returncode == 0 and out.stdout.startswith("ok"), out.stdout + out.stderr
