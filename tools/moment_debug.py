"""Debug aid (round 5): CV-likelihood CKDE terms with the tile-moment pass on / off, against each other and against the unpruned sweeps."""
import os, sys
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn

def scores(df, cands, k=3, seed=5):
    model = pbn.SemiparametricBN(list(df.columns))
    out = {}
    for tag, env in (("moment", {"PBN_MOMENT_PASS": "1"}), ("sweep", {"PBN_MOMENT_PASS": "0"}), ("unpruned", {"PBN_SWEEP_PRUNE": "0"})):
        for k_, v_ in env.items(): os.environ[k_] = v_
        s = pbn.CVLikelihood(df, k=k, seed=seed)
        out[tag] = [s.local_score_node_type(model, pbn.CKDEType(), v, ev) for v, ev in cands]
        for k_ in env: os.environ.pop(k_)
    return out

rng = np.random.default_rng(11)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 90000
a = rng.normal(size=n)
b = np.tanh(a) + 0.4 * rng.normal(size=n)
c = 0.5 * a - 0.7 * b + 0.5 * rng.standard_t(5, size=n)
tables = {"normal": pd.DataFrame({"a": a, "b": b}), "heavy": pd.DataFrame({"a": c, "b": b}), "heavy1": pd.DataFrame({"a": c, "b": a})}
for pin in ({}, {"PBN_PRUNE_MARGIN": "52"}):
    for k_, v_ in pin.items(): os.environ[k_] = v_
    for name, df in tables.items():
        r = scores(df, [("a", []), ("b", []), ("b", ["a"])])
        for i, cand in enumerate(["a|", "b|", "b|a"]):
            m, s, u = r["moment"][i], r["sweep"][i], r["unpruned"][i]
            print(f"pin={pin} {name:7s} {cand:4s}: moment {m:.10f} sweep {s:.10f} unpruned {u:.10f}  rel(m,u) {abs(m-u)/abs(u):.2e} rel(s,u) {abs(s-u)/abs(u):.2e}", flush=True)
    for k_ in pin: os.environ.pop(k_)
