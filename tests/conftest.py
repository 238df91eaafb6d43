import importlib.util
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, "tests", "golden")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "pins_oracle: oracle-vs-golden checks (no GPU needed) that ALSO run under -m gpu, so that the "
                                       "checker is pinned on the box where it checks")


@pytest.hookimpl(tryfirst=True)
def pytest_collection_modifyitems(config, items):
    """`-m gpu` (the driver's round-end run on the MI355X) also selects the tests that pin the oracle against the reference's
    golden vectors: they need no GPU, but they are what makes "HIP == oracle" mean "HIP == reference" on that box."""
    if (config.getoption("-m") or "").strip() == "gpu":
        for item in items:
            if item.get_closest_marker("pins_oracle") is not None:
                item.add_marker(pytest.mark.gpu)


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


REF_SRC = "/root/reference/src"


@pytest.fixture
def reference_models(cpu_kernels, monkeypatch):
    import surfacenetworks_amd.utils_pt as U

    monkeypatch.setattr(sys, "dont_write_bytecode", True)        # (the reference mount is read-only)
    monkeypatch.syspath_prepend(REF_SRC)
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, k)
    import utils                                                  # the reference's package (graph.py, mesh.py stay its own)

    assert os.path.realpath(os.path.dirname(utils.__file__)) == os.path.realpath(os.path.join(REF_SRC, "utils"))
    # ---- the import swap: the one line a maintainer changes in each models.py / main.py --------------------------------
    monkeypatch.setitem(sys.modules, "utils.utils_pt", U)
    monkeypatch.setattr(utils, "utils_pt", U, raising=False)
    mods = {}
    for task in ("as_rigid_as_possible", "mesh_mnist", "dense_correspondence"):
        spec = importlib.util.spec_from_file_location(f"ref_{task}_models", os.path.join(REF_SRC, task, "models.py"))
        mod = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(mod)
        assert mod.utils is U, "the reference model file did not pick up the swapped operator layer"
        mods[task] = mod
    yield mods
    for k in [k for k in sys.modules if k == "utils" or k.startswith("utils.")]:
        monkeypatch.delitem(sys.modules, k, raising=False)
