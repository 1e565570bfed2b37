"""Worker of tests/test_switches_gpu.py: evaluates a fixed set of scores / p-values with whatever PBN_* switches its environment
carries and prints them as one JSON line.  Run in a subprocess because the switches are read once per process."""
import json
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import pybnesian_amd as pbn  # noqa: E402
from test_mi_gpu import hybrid_table  # noqa: E402

out = {}
rng = np.random.default_rng(12)

# (1) CV-likelihood CKDE local scores + a short hill-climb on a table large enough for the pruned sweeps (issue lanes,
#     prepass-bound offsets, one-wave workgroups)
n = 60000   # 45 000 training rows per fold: above the 32 768-row threshold of the pruned sweeps
x = rng.normal(size=(n, 4))
x[:, 1] += 0.8 * x[:, 0]
x[:, 2] += 0.5 * x[:, 0] * x[:, 1]
x[:, 3] += np.sin(x[:, 2])
for dtype in ("float64", "float32"):
    df = pd.DataFrame(x.astype(dtype), columns=list("abcd"))
    score = pbn.CVLikelihood(df, 4, 3)
    spbn = pbn.SemiparametricBN(list("abcd"))
    vals = []
    for var, par in (("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"]), ("a", [])):
        vals.append(score.local_score_node_type(spbn, pbn.CKDEType(), var, par))
    out[f"cv_ckde_{dtype}"] = vals
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.KDENetwork(list("abcd")), max_iters=3)
    out[f"hc_arcs_{dtype}"] = sorted(map(list, res.arcs()))

# (2) hybrid MutualInformation p-values (per-grouping moments against per-test kernels) incl. a table with nulls
for tag, nulls in (("plain", False), ("nulls", True)):
    hdf = hybrid_table(30000, 7)
    if nulls:
        hdf.loc[hdf.index[::17], "c2"] = np.nan
        hdf.loc[hdf.index[::23], "d1"] = np.nan
    test = pbn.MutualInformation(hdf)
    names = list(hdf.columns)
    pv = []
    r2 = np.random.default_rng(3)
    for _ in range(60):
        k = int(r2.integers(0, 4))
        sel = [names[i] for i in r2.choice(len(names), size=k + 2, replace=False)]
        pv.append(test.pvalue(sel[0], sel[1], sel[2:]) if k else test.pvalue(sel[0], sel[1]))
    out[f"mi_{tag}"] = pv

# (3) BIC of Gaussian candidates (Gram kernels)
g = pd.DataFrame(rng.normal(size=(30001, 20)) @ (np.eye(20) + 0.2 * np.tril(rng.normal(size=(20, 20)), -1)).T + 50.0,
                 columns=[f"x{i}" for i in range(20)])
bic = pbn.BIC(g)
gbn = pbn.GaussianNetwork(list(g.columns))
out["bic"] = [bic.local_score(gbn, "x7", ["x1", "x19", "x4"]), bic.local_score(gbn, "x0", []), bic.local_score(gbn, "x12", ["x3"])]
# (4) likelihood scores of hybrid candidates (CKDE and LinearGaussian children of discrete parents, a discrete child)
for dtype in ("float64", "float32"):
    hdf = hybrid_table(40000, 13, dtype)
    hs = pbn.CVLikelihood(hdf, 4, 1)
    hnet = pbn.SemiparametricBN(list(hdf.columns))
    cands = [("c2", pbn.CKDEType(), ["c1", "d1"]), ("c4", pbn.CKDEType(), ["d2", "c2", "d3"]), ("c3", pbn.LinearGaussianCPDType(), ["d3", "c1"]),
             ("c1", pbn.CKDEType(), ["d1"]), ("d2", pbn.DiscreteFactorType(), ["d1", "d3"]), ("c2", pbn.LinearGaussianCPDType(), ["d1", "d2", "c1", "c3"])]
    out[f"hybrid_{dtype}"] = [hs.local_score_node_type(hnet, t, v, p) for v, t, p in cands]
    # the same candidates - and some that share terms with them, one of them twice - as ONE batch of a fresh engine: their slices go
    # through one grouped chain (PBN_HYBRID_BATCH=0: one chain per candidate)
    hb = pbn.CVLikelihood(hdf, 4, 1)
    more = [("c1", pbn.CKDEType(), ["c2", "d1"]), ("c2", pbn.CKDEType(), ["d1", "c1"]), ("c3", pbn.CKDEType(), ["d1"]), ("c4", pbn.CKDEType(), ["c1", "d1"])]
    out[f"hybrid_batch_{dtype}"] = [float(x) for x in hb._batch(hnet, [(v, t, p) for v, t, p in cands + more], hb._kind)]
print("RESULT " + json.dumps(out))
