import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run through gpurun)")


@pytest.fixture(scope="session")
def golden():
    return np.load(os.path.join(ROOT, "tests", "golden", "intree_v1.npz"), allow_pickle=False)


@pytest.fixture(scope="session")
def fc():
    """The product package with its HIP library initialised on device 0."""
    import firecode_amd as fc

    fc.init(0)
    return fc
