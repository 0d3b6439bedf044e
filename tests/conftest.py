import os
import sys

import pytest
import torch  # noqa: F401  -- first: torch brings its own HIP runtime, which finds no GPU when it is loaded after libmipt.so's

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs an MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def oracle():
    from oracle import orc
    if not os.path.exists(os.path.join(ROOT, "oracle", "liboracle.so")):
        orc.build(ref=False)
    return orc.Oracle()


@pytest.fixture(scope="session")
def reference():
    """The reference's own code (oracle/_ref/libref.so); tests that need it are skipped without it."""
    from oracle import orc
    if orc.ref() is None:
        pytest.skip("oracle/_ref/libref.so not built (needs the reference checkout)")
    return orc.Reference()
