import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "differentiable-piso_amd")
for p in (ROOT, PKG):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden_dir():
    return os.path.join(ROOT, "tests", "golden")


@pytest.fixture
def piso_option():
    """Set tuning / test knobs of libpiso_hip.so for one test (piso_set_option); restored afterwards."""
    import diffpiso._native as N
    saved = {}

    def set_(name, value):
        saved.setdefault(name, N.get_option(name))
        N.set_option(name, value)
    yield set_
    for name, value in saved.items():
        N.set_option(name, value)
