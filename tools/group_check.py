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
assert worst < 1e-9, worst
print("group_check ok")
