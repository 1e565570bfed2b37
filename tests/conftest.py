import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")
    config.addinivalue_line("markers", "extra: round-1 surface outside SURVEY.md section 8 (BDe, user-defined network types, discrete-only models) - "
                                       "runs only when asked for by name: -m extra (add PBN_EXTRA_GPU=1 on a GPU box for the halves that fit factors)")
    # torch first: it ships its own HIP runtime, and a process in which libpbn_hip.so (linked against /opt/rocm) has touched the
    # GPU before torch was imported ends up with torch.cuda.is_available() == False (measured on the GPU box, tools/torch_after.py).
    # bench.py imports torch first for the same reason; the library itself never needs torch.
    try:
        import torch  # noqa: F401
    except Exception:  # pragma: no cover - torch is only plumbing for a few tests
        pass


@pytest.fixture(scope="session")
def golden():
    import numpy as np

    return np.load(os.path.join(ROOT, "tests", "golden", "reference_recipes.npz"))


@pytest.fixture(scope="session")
def ensure_built():
    """Build libpbn_hip.so / the oracle if they are missing (hipcc cross-compiles without a GPU)."""
    import __graft_entry__ as g

    g.build()


def pytest_collection_modifyitems(config, items):
    """Tests marked `extra` never run in the driver's tiers (`-m "not gpu"`, `-m gpu`): they are deselected unless the marker expression
    names them.  The GPU tier's time goes to the path."""
    expr = config.getoption("-m") or ""
    if "extra" in expr:
        return
    keep, drop = [], []
    for it in items:
        (drop if it.get_closest_marker("extra") else keep).append(it)
    if drop:
        config.hook.pytest_deselected(items=drop)
        items[:] = keep
