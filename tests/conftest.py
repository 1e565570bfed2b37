import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "reference_recipes.npz"))


@pytest.fixture(scope="session")
def ensure_built():
    """Build libpbn_hip.so / the oracle if they are missing (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()
