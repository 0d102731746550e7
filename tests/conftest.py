import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu)")


def load_golden(name):
    return np.load(os.path.join(GOLDEN, name))


@pytest.fixture(scope="session")
def oracle_mod():
    import oracle
    oracle.build()
    return oracle


@pytest.fixture(scope="session")
def gpu_device():
    import torch
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    return torch.device("cuda:0")


class _LabEnv:
    """Flip a lab switch of the kernels' dispatch for one test: the process is routed through the lab build
    (taiyaki_amd/csrc/libtaiyaki_amd_flipflop_lab.so, -DTK_LAB) -- the release library reads none of them."""

    def __init__(self, monkeypatch):
        self._mp = monkeypatch

    def setenv(self, name, value):
        from taiyaki_amd import _lib
        _lib.use_lab(True)
        self._mp.setenv(name, value)

    def delenv(self, name, raising=False):
        self._mp.delenv(name, raising=raising)

    def lib(self):
        from taiyaki_amd import _lib
        return _lib.use_lab(True)


@pytest.fixture
def labenv(monkeypatch):
    from taiyaki_amd import _lib
    yield _LabEnv(monkeypatch)
    _lib.use_lab(False)
