"""Shared helpers for the parity tests (data frames from the golden arrays, tolerances)."""
import numpy as np
import pandas as pd

COLS = ["a", "b", "c", "d"]
VARSETS = [["a"], ["b", "a"], ["c", "a", "b"], ["d", "a", "b", "c"]]
CKDE_SETS = [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]

# BASELINE.json north_star: slogl within 1e-6 relative (fp64) / 1e-3 (fp32)
RTOL_F64 = 1e-6
RTOL_F32 = 1e-3


def frame(arr, dtype=None):
    df = pd.DataFrame(np.asarray(arr), columns=COLS)
    return df.astype(dtype) if dtype else df


def rel_err(a, b):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.max(np.abs(a - b) / np.maximum(np.abs(b), 1e-300))
