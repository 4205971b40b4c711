import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def orc():
    """The CPU oracle (checker).  Built on demand with g++."""
    from oracle import oracle
    o = oracle()
    o.set_backend("builtin")
    return o


@pytest.fixture(scope="session")
def hip():
    """The product library; loading fails loudly if it was not built."""
    import dqmc_amd
    lib = dqmc_amd.lib()
    if lib.device_count() == 0:
        pytest.fail("HIP library loaded but no GPU is visible: -m gpu tests need a gfx950 device")
    return lib


def rel_err(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(1.0, np.abs(np.asarray(b)).max()))
