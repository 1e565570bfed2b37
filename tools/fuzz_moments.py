"""Shape fuzz of the segmented Gram of the score-data constructor (gram_glds_kernel / gram_glds_f32_kernel on pieces of fold segments):
random rows (2 ... 300 000), columns (1 ... 70), dtypes and fold counts; BIC and Gaussian CVLikelihood local scores against the oracle.
    python3 tools/fuzz_moments.py [cases, default 60] [seed]"""
import os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn
from oracle import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 3)
worst, bad, t0 = {"bic": 0.0, "cv": 0.0}, 0, time.time()
for case in range(cases):
    rows = int(rng.choice([int(rng.integers(30, 600)), int(rng.integers(600, 20000)), int(rng.integers(20000, 300000))]))
    cols = int(rng.choice([1, 2, 7, 16, 17, 32, 33, 48, 64, 65, 70, int(rng.integers(1, 71))]))
    if rows * cols > 6_000_000:
        rows = 6_000_000 // cols
    dtype = "float64" if rng.random() < 0.6 else "float32"
    mix = np.eye(cols) + 0.15 * np.tril(rng.normal(size=(cols, cols)), -1)
    data = (rng.normal(size=(rows, cols)) @ mix.T * rng.uniform(0.5, 2.0, size=cols) + rng.uniform(-30, 30, size=cols)).astype(dtype)
    df = pd.DataFrame(data, columns=[f"x{i}" for i in range(cols)])
    net = pbn.GaussianNetwork(list(df.columns))
    bic = pbn.BIC(df)
    k = int(rng.integers(2, 11))
    seed = int(rng.integers(0, 1000))
    cv = pbn.CVLikelihood(df, k=k, seed=seed) if rows >= 20 * k else None
    d64 = data.astype(np.float64)
    for _ in range(4):
        v = int(rng.integers(cols))
        p = int(rng.integers(0, min(cols, 9)))
        par = [int(q) for q in rng.choice([c for c in range(cols) if c != v], size=min(p, cols - 1), replace=False)]
        if rows <= len(par) + 2:
            continue
        names = [f"x{q}" for q in par]
        got, want = bic.local_score(net, f"x{v}", names), oracle.bic_lg(d64[:, [v] + par])
        err = abs(got - want) / max(abs(want), 1e-9)
        worst["bic"] = max(worst["bic"], err)
        if err > 1e-8:
            bad += 1; print("MISMATCH bic", rows, cols, dtype, v, par, got, want, err)
        if cv is not None:
            got, want = cv.local_score(net, f"x{v}", names), oracle.cv_likelihood(data[:, [v] + par], "lg", k, seed)
            err = abs(got - want) / max(abs(want), 1e-9)
            tol = 1e-7 if dtype == "float64" else 5e-4
            if dtype == "float64":
                worst["cv"] = max(worst["cv"], err)
            if err > tol:
                bad += 1; print("MISMATCH cv", rows, cols, dtype, k, v, par, got, want, err)
print(f"{cases} shapes x 4 candidates in {time.time() - t0:.0f} s: {'all ok' if not bad else str(bad) + ' MISMATCHES'}; worst relative difference BIC {worst['bic']:.2e}, "
      f"Gaussian CV likelihood (fp64 tables) {worst['cv']:.2e}")
sys.exit(1 if bad else 0)
