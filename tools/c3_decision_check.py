"""C3 at FULL size: are the search's first decisions the reference arithmetic's decisions?  ("missing" 6: the serial oracle
cannot finish a C3 search, so "identical trace at 500 000 rows" only ever meant identical to the previous build.)
For the first ITERS iterations of the C3 search (32-node SemiparametricBN, all CKDE, 10-fold CVLikelihood, 500 000 rows, arcs + node types)
the device's best operator and its runner-up are re-scored on the CPU - the local scores behind both deltas by the tuned CPU port of the
reference algorithm (oracle/pbn_baseline.cpp: held to the checker at 1e-10; bandwidths and folds by the oracle) on ALL rows - and the order of
the two CPU deltas must be the device's.  ~30 s of 16-thread CPU work per CKDE local score.
    python tools/c3_decision_check.py [iterations=2] > profiles/r6/c3_decision_check.txt"""
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import pybnesian_amd as pbn  # noqa: E402
from oracle import baseline, oracle  # noqa: E402
from pybnesian_amd import _lib  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 2
n_rows, n_cols, k, seed = int(os.environ.get("C3_ROWS", "500000")), 32, 10, 0
try:
    q = bench.cpu_quota()
    if q:
        baseline.set_num_threads(max(1, int(q)))
        oracle.set_num_threads(max(1, int(q)))
except Exception:
    pass
ctx = pbn.Context(0)
dev = torch.device("cuda", 0)
t = bench.make_dag_table(torch, dev, n_rows, n_cols, 2, torch.float64, nonlinear=True)
names = [f"x{i}" for i in range(n_cols)]
torch.cuda.synchronize()
table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
score = pbn.CVLikelihood(None, k, seed, table=table)
host = t.T.cpu().numpy()                       # rows x columns
perm, limits, n_cv, _ = score._layout()
perm = perm[:n_cv]
col = {n: i for i, n in enumerate(names)}


def cpu_local(var, parents, node_type):
    """CVLikelihood::local_score (cv_likelihood.cpp:5-25) on all rows: per fold fit on the other folds, slogl of the fold."""
    cols = [col[var]] + [col[p] for p in parents]
    total = 0.0
    for f in range(k):
        te = perm[limits[f]: limits[f + 1]]
        tr = np.concatenate([perm[: limits[f]], perm[limits[f + 1]:]])
        dtr, dte = host[np.ix_(tr, cols)], host[np.ix_(te, cols)]
        if node_type == "lg":
            beta, var_ = oracle.lg_fit(dtr)
            total += float(oracle.lg_logl(dte, beta, var_).sum())
        else:
            cov, _ = oracle.cov(dtr)
            H = oracle.bandwidth(0, 0, cov, dtr.shape[0])           # normal reference rule, full matrix (CKDE.hpp:186-199)
            total += float(baseline.ckde_logl(dtr, H, dte).sum() if len(cols) > 1 else baseline.kde_logl(dtr, H, dte).sum())
    return total


def cpu_delta(model, op):
    nt = lambda v: "lg" if model.node_type(v) == pbn.LinearGaussianCPDType() else "ckde"
    kind = type(op).__name__
    if kind == "ChangeNodeType":
        v = op.node()
        new = "lg" if op.node_type() == pbn.LinearGaussianCPDType() else "ckde"
        return cpu_local(v, model.parents(v), new) - cpu_local(v, model.parents(v), nt(v))
    s, d = op.source(), op.target()
    pa = list(model.parents(d))
    if kind == "AddArc":
        return cpu_local(d, pa + [s], nt(d)) - cpu_local(d, pa, nt(d))
    if kind == "RemoveArc":
        return cpu_local(d, [p for p in pa if p != s], nt(d)) - cpu_local(d, pa, nt(d))
    ps = list(model.parents(s))                                        # flip s -> d into d -> s
    return (cpu_local(d, [p for p in pa if p != s], nt(d)) + cpu_local(s, ps + [d], nt(s))) - cpu_local(d, pa, nt(d)) - cpu_local(s, ps, nt(s))


model = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
pool = pbn.OperatorPool([pbn.ArcOperatorSet(max_indegree=3), pbn.ChangeNodeTypeSet()])
t0 = time.perf_counter()
pool.cache_scores(model, score)
print(f"# C3 table {n_rows} x {n_cols}, {k}-fold CVLikelihood; cache_scores {time.perf_counter() - t0:.2f} s; CPU threads {baseline.num_threads()}", flush=True)
ok = True
for it in range(1, iters + 1):
    best = pool.find_max(model)
    tabu = pbn.OperatorTabuSet()
    tabu.insert(best)
    second = pool.find_max_tabu(model, tabu)
    t0 = time.perf_counter()
    cb, cs = cpu_delta(model, best), cpu_delta(model, second)
    dt = time.perf_counter() - t0
    agree = (cb > cs) == (best.delta() > second.delta())
    ok = ok and agree
    print(f"iteration {it}: best {best} device delta {best.delta():.6f} cpu {cb:.6f} (rel {abs(best.delta() - cb) / abs(cb):.2e}); "
          f"runner-up {second} device {second.delta():.6f} cpu {cs:.6f} (rel {abs(second.delta() - cs) / abs(cs):.2e}); "
          f"gap device {best.delta() - second.delta():.4f} cpu {cb - cs:.4f}; same order: {agree}; cpu {dt:.0f} s", flush=True)
    best.apply(model)
    pool.update_scores(model, score, best.nodes_changed(model))
print("ALL DECISIONS AGREE" if ok else "DISAGREEMENT", flush=True)
sys.exit(0 if ok else 1)
