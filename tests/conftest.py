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
    config.addinivalue_line("markers", "eight_ranks: starts EIGHT processes on the box's one GPU (runs before this process creates its own GPU context)")


def pytest_collection_modifyitems(config, items):
    """Tests that put eight rank processes on the ONE GPU of the test box run FIRST, while the pytest process itself has no GPU
    context yet.  The driver gives eight processes a hardware context each; a ninth (this process, once any GPU test has run in
    it) makes it time-slice whole processes, and a rank whose kernel spins on a peer's mailbox then waits for a peer that is not
    scheduled - the bounded waits give up (seen: `a wait on a peer's mailbox gave up` in the full suite only, never with the
    multi-process files alone).  On a real node every rank has its own GPU and the question does not arise."""
    first = [it for it in items if it.get_closest_marker("eight_ranks") is not None]
    if first:
        rest = [it for it in items if it.get_closest_marker("eight_ranks") is None]
        items[:] = first + rest


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
