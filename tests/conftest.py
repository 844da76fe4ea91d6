"""pytest configuration: markers + shared fixtures.

``-m "not gpu"`` runs in the build container (no GPU; /root/reference present);
``-m gpu`` runs on the MI355X box (GPU present; /root/reference ABSENT) -- GPU tests therefore only
use the committed fixtures under tests/golden/.
"""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
if os.path.join(ROOT, "tests") not in sys.path:
    sys.path.insert(0, os.path.join(ROOT, "tests"))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, "tests", "golden")
REFERENCE = os.environ.get("GRID2OP_REFERENCE", "/root/reference")


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "reference: needs the read-only reference checkout (/root/reference)")


def have_reference() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "grid2op"))


# tests/test_backend_conformance.py imports the reference's own test classes at module level: without the reference checkout
# (GPU box) there is nothing to collect -- ignored rather than reported as a skipped module in every `-m gpu` run
collect_ignore = [] if have_reference() else ["test_backend_conformance.py"]


def golden_path(name: str) -> str:
    return os.path.join(GOLDEN, name)


@pytest.fixture(scope="session")
def load_model():
    from grid2op_amd.grid_model import GridModel
    cache = {}

    def _load(name):
        if name not in cache:
            cache[name] = GridModel.load_npz(golden_path(f"{name}.grid.npz"))
        return cache[name]
    return _load


@pytest.fixture(scope="session")
def load_npz():
    def _load(fname):
        return dict(np.load(golden_path(fname)))
    return _load
