"""Grouped (one launch chain per batch of variable sets, kde_group.hip) against the per-(set, fold) chain of round 2 and, on a
small table, against the oracle: CV-likelihood and hold-out CKDE local scores.   python3 tools/group_check.py"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402


def table(n, seed=5):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, 5))
    x[:, 1] += 0.8 * x[:, 0]
    x[:, 2] += 0.5 * x[:, 0] * x[:, 1]
    x[:, 3] += np.sin(x[:, 2])
    x[:, 4] = np.tanh(x[:, 3]) + 0.3 * x[:, 4]
    return pd.DataFrame(x, columns=list("abcde"))


CANDS = (("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"]), ("a", []), ("e", ["d"]), ("a", ["b"]), ("e", ["a", "c", "d"]))


def scores(df, make, grouped, min_rows=None):
    os.environ["PBN_SCORE_GROUPED"] = "1" if grouped else "0"
    if min_rows is not None:
        os.environ["PBN_PRUNE_MIN_ROWS"] = str(min_rows)
    score = make(df)
    bn = pbn.SemiparametricBN(list(df.columns))
    t0 = time.perf_counter()
    out = [score.local_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in CANDS]
    return np.array(out), time.perf_counter() - t0


worst = 0.0
for n, min_rows in ((3000, 64), (60000, None), (300000, None)):
    df = table(n)
    for name, make in (("cv", lambda d: pbn.CVLikelihood(d, 5, 3)), ("holdout", lambda d: pbn.HoldoutLikelihood(d, 0.2, 3))):
        a, ta = scores(df, make, True, min_rows)
        b, tb = scores(df, make, False, min_rows)
        rel = np.max(np.abs(a - b) / np.abs(b))
        worst = max(worst, rel)
        print(f"n={n} {name}: grouped vs per-unit max rel diff {rel:.3e}  ({ta:.3f}s vs {tb:.3f}s)", flush=True)
        if n == 3000 and name == "cv":
            from oracle import oracle

            want = [oracle.cv_likelihood(df[[v] + p].to_numpy(), "ckde", 5, 3) for v, p in CANDS]
            relo = np.max(np.abs(a - np.array(want)) / np.abs(want))
            print(f"   vs oracle: {relo:.3e}")
            worst = max(worst, relo)
assert worst < 3e-7, worst   # two pruned forms: each may drop up to 1.1e-7 (+ 8e-8) of a sum; measured 1.2e-9

# hybrid candidates (discrete parents: one pool per configuration and term), fp64 and fp32 (f16x2 fragments), CV and validation scores
def hybrid(n, dtype, seed=9):
    rng = np.random.default_rng(seed)
    A = rng.integers(0, 3, size=n)
    B = rng.integers(0, 2, size=n)
    x = rng.normal(size=n) + 1.5 * A
    y = 0.6 * x + np.where(B == 1, 1.0, -1.0) + rng.normal(scale=0.7, size=n)
    z = np.tanh(y) + 0.4 * x + rng.normal(scale=0.5, size=n)
    df = pd.DataFrame({"x": x.astype(dtype), "y": y.astype(dtype), "z": z.astype(dtype)})
    df["A"] = pd.Categorical.from_codes(A, ["a0", "a1", "a2"])
    df["B"] = pd.Categorical.from_codes(B, ["b0", "b1"])
    return df


HC = (("y", ["x", "B"]), ("z", ["x", "y", "A"]), ("x", ["A"]), ("z", ["A", "B"]), ("y", ["x", "z", "A", "B"]))
for dtype, tol in (("float64", 1e-9), ("float32", 2e-5)):
    df = hybrid(400000, dtype)
    bn = pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])
    res = {}
    for grouped in (True, False):
        os.environ["PBN_SCORE_GROUPED"] = "1" if grouped else "0"
        os.environ.pop("PBN_PRUNE_MIN_ROWS", None)
        score = pbn.ValidatedLikelihood(df, 0.2, 5, 1)
        t0 = time.perf_counter()
        res[grouped] = ([score.local_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in HC] +
                        [score.vlocal_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in HC], time.perf_counter() - t0)
    a, b = np.array(res[True][0]), np.array(res[False][0])
    rel = np.max(np.abs(a - b) / np.abs(b))
    print(f"hybrid {dtype}: grouped vs per-slice max rel diff {rel:.3e}  ({res[True][1]:.3f}s vs {res[False][1]:.3f}s)", flush=True)
    assert rel < tol, (dtype, rel)
print("group_check ok")
