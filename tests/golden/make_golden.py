"""Regenerates tests/golden/reference_cases.npz from the compiled reference (oracle/_ref/libref.so).

Run in a container that holds the reference checkout:  make -C oracle ref && python tests/golden/make_golden.py
The file stores the deterministic inputs of tests/golden_cases.py (prefix `in_`) and what the
reference's own code returned for them (prefix `out_`): arrays of numbers only.
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
sys.path.insert(0, os.path.dirname(HERE))

import golden_cases as G   # noqa: E402
from oracle import orc     # noqa: E402


def main():
    be = G.RefBackend(orc.Reference())
    inp = G.make_inputs()
    out = G.evaluate(be, inp)
    blob = {"in_" + k: v for k, v in inp.items()}
    blob.update({"out_" + k: v for k, v in out.items()})
    path = os.path.join(HERE, "reference_cases.npz")
    np.savez_compressed(path, **blob)
    print("wrote", path, os.path.getsize(path), "bytes,", len(out), "cases")


if __name__ == "__main__":
    main()
