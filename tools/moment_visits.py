"""Lane utilisation of the moment pass: (tile, group) pairs taken per (64-tile batch, group) pass (PBN_SWEEP_COUNT_REDO counters), on a C3-shaped CV term."""
import ctypes as C, os, sys
import numpy as np, pandas as pd, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pybnesian_amd as pbn
from pybnesian_amd import _lib
lib = _lib.load()
n, cols = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (500000, 32)
t = bench.make_dag_table(torch, torch.device("cpu"), n, cols, 2, torch.float64, nonlinear=True).numpy()
df = pd.DataFrame({f"x{i}": t[i] for i in (0, 1, 10, 21)})
model = pbn.SemiparametricBN(list(df.columns))
os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
for cand in (("x0", []), ("x1", ["x0"]), ("x21", ["x10"])):
    lib.pbn_debug_moment_pairs(None, None, 1); lib.pbn_debug_sweep_visits(None, None, 1); lib.pbn_debug_moment_left(None, 1)
    s = pbn.CVLikelihood(df, k=10, seed=0)
    s.local_score_node_type(model, pbn.CKDEType(), *cand)
    p, b, v, tt, mv = (C.c_ulonglong(0) for _ in range(5))
    lib.pbn_debug_moment_pairs(C.byref(p), C.byref(b), 0); lib.pbn_debug_sweep_visits(C.byref(v), C.byref(tt), 0); lib.pbn_debug_moment_visits(C.byref(mv))
    lf = C.c_ulonglong(0); lib.pbn_debug_moment_left(C.byref(lf), 0)
    print(f"    (batch, group) masks with pairs left for the sweep: {lf.value:.3e}")
    print(f"{cand}: moment pairs {p.value:.3e} in {b.value:.3e} (batch, group) passes = {p.value / max(b.value, 1):.1f} of 64 lanes, {mv.value:.3e} (batch, group) masks walked; "
          f"sweep pairs left {v.value:.3e} of {tt.value:.3e} offered", flush=True)
    del os.environ["PBN_SWEEP_COUNT_REDO"]
    import time
    for mp in ("1", "0"):
        os.environ["PBN_MOMENT_PASS"] = mp
        s = pbn.CVLikelihood(df, k=10, seed=0); s.local_score_node_type(model, pbn.CKDEType(), *cand)
        s = pbn.CVLikelihood(df, k=10, seed=0); t0 = time.perf_counter(); s.local_score_node_type(model, pbn.CKDEType(), *cand); dt = time.perf_counter() - t0
        print(f"    PBN_MOMENT_PASS={mp}: {dt * 1e3:.1f} ms", flush=True)
    del os.environ["PBN_MOMENT_PASS"]; os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
