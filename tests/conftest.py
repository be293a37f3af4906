import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    def load(name):
        return np.load(os.path.join(GOLDEN, name + ".npz"), allow_pickle=False)

    return load


@pytest.fixture(scope="session", autouse=True)
def _built_library():
    """The C-ABI library is an in-tree build product (git-ignored): build it (incrementally, hipcc cross-compiles
    gfx950 without a GPU) before any test touches it, so a fresh checkout or a stale object never decides a test."""
    from scaleprotoseg_amd.build import build

    build()
