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
def golden_dir():
    return GOLDEN


@pytest.fixture(scope="session", autouse=True)
def _built():
    """Make sure the in-tree HIP library and the C oracle exist (cross-compiles fine without a GPU)."""
    import __graft_entry__ as g

    g.build()


@pytest.fixture
def cpu_kernels(monkeypatch):
    """Run the product's host logic on CPU tensors with oracle-backed kernels (test-only seam)."""
    import cpu_kernels as ck

    ck.install(monkeypatch)
    return ck
