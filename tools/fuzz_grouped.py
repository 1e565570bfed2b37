"""Randomised comparison of the grouped evaluation (kde_group.hip) against the per-(set, fold) chains and against the unpruned sweeps:
CV-likelihood CKDE local scores on awkward tables (heavy tails, clusters, duplicated rows, lattice-valued columns, large offsets,
nearly collinear columns), 1-3 parents, 2-10 folds, fp64 and fp32.   python tools/fuzz_grouped.py [n_cases] [seed]"""
import os, sys
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn

n_cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {"float64": [0.0, 0.0], "float32": [0.0, 0.0]}
for case in range(n_cases):
    d = 4
    dtype = "float64" if rng.random() < 0.6 else "float32"
    k = int(rng.choice([2, 3, 5, 10]))
    n = int(rng.choice([70_001, 90_000, 150_000]))
    kind = rng.choice(["cauchy", "clusters", "dups", "lattice", "offset", "line", "normal"])
    if kind == "cauchy":
        x = rng.standard_t(2.5, size=(n, d))
    elif kind == "clusters":
        c = rng.uniform(-50, 50, size=(5, d))
        x = c[rng.integers(0, 5, size=n)] + rng.normal(scale=rng.uniform(0.3, 3.0), size=(n, d))
    elif kind == "dups":
        base = rng.normal(size=(2000, d))
        x = base[rng.integers(0, 2000, size=n)] + rng.normal(scale=1e-2, size=(n, d))
    elif kind == "lattice":
        x = rng.integers(-5, 6, size=(n, d)).astype(float) + rng.normal(scale=0.2, size=(n, d))
    elif kind == "offset":
        x = 1e3 + rng.normal(size=(n, d)) * np.arange(1, d + 1)
    elif kind == "line":
        t = rng.normal(size=(n, 1))
        x = t @ np.ones((1, d)) + rng.normal(scale=0.1, size=(n, d))
    else:
        x = rng.normal(size=(n, d)) @ (np.eye(d) + 0.4 * np.tril(rng.normal(size=(d, d)), -1)).T
    names = list("abcd")
    df = pd.DataFrame(x, columns=names).astype(dtype)
    cands = [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]
    bn = pbn.SemiparametricBN(names)
    vals = {}
    seed = int(rng.integers(0, 100))
    try:
        for tag, env in (("grouped", {"PBN_SCORE_GROUPED": "1"}), ("per_unit", {"PBN_SCORE_GROUPED": "0"}), ("unpruned", {"PBN_SCORE_GROUPED": "0", "PBN_SWEEP_PRUNE": "0"})):
            for k_, v_ in env.items():
                os.environ[k_] = v_
            score = pbn.CVLikelihood(df, k, seed)
            vals[tag] = np.array([score.local_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in cands])
            os.environ.pop("PBN_SWEEP_PRUNE", None)
    except Exception as ex:
        print(f"case {case} {dtype} {kind} n={n} k={k}: {type(ex).__name__}: {str(ex)[:100]}", flush=True)
        continue
    r1 = float(np.max(np.abs(vals["grouped"] - vals["per_unit"]) / np.abs(vals["per_unit"])))
    r2 = float(np.max(np.abs(vals["grouped"] - vals["unpruned"]) / np.abs(vals["unpruned"])))
    worst[dtype][0] = max(worst[dtype][0], r1)
    worst[dtype][1] = max(worst[dtype][1], r2)
    print(f"case {case:3d} {dtype} {kind:8s} n={n} k={k}: grouped vs per-unit {r1:.2e}, vs unpruned {r2:.2e}", flush=True)
print("worst (grouped vs per-unit, grouped vs unpruned):", worst)
# what the three forms may differ by is the mass their own pruning drops: at most 1.1e-7 (+ 8e-8 fp32 tail) of a sum per form at the shipped
# sum-only margin (fp32: 1.5e-5) - DESIGN.md section 4.  Measured: 5e-10 with the sigma/16 key cells of round 3, 2e-9 with the compact tiles of
# round 5 (their boxes leave less slack, so more of the allowed mass is really dropped); with the margins pinned (52 / 40) the forms agree to 1e-11.
assert worst["float64"][0] < 3e-7 and worst["float64"][1] < 3e-7 and worst["float32"][0] < 1e-4 and worst["float32"][1] < 1e-4, worst
