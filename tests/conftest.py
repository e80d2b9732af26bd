import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
# The tests look inside the library (occurrence tables, class counts, A/B knobs): they load the development build,
# libgaml_hip_dev.so -- the same sources plus include/gaml_hip_debug.h. The product library is checked by
# tests/test_abi.py (no debug surface) and tests/test_gpu_release_lib.py (parity through it), smoke() and bench.py.
os.environ.setdefault("GAML_HIP_FLAVOUR", "dev")
sys.path.insert(0, os.path.join(ROOT, "oracle"))  # oracle_py: the checker (tests only)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


def _have_gpu():
    try:
        import torch
        return torch.cuda.is_available()
    except Exception:
        return False


@pytest.fixture(scope="session")
def built():
    """Product + oracle built from THIS tree: make is dependency-tracked, so this is a no-op when the shared objects are
    current and a rebuild when a source changed (tests/test_abi.py additionally checks the source hash the library carries)."""
    import __graft_entry__ as ge
    ge.build()
    return True


def pytest_collection_modifyitems(config, items):
    if _have_gpu():
        return
    skip = pytest.mark.skip(reason="no GPU in this container")
    for it in items:
        if "gpu" in it.keywords:
            it.add_marker(skip)
