"""Does tile pruning pay at 6 and 7 dimensions?  KDE.slogl at 1e6 x 1e5 rows, fp64 and fp32, correlated / independent / heavy-tailed
data, PBN_PRUNE_MAX_DIMS = 5 against 7 (kde_prune_applies: 6 is the default).  python tools/prune_dims67.py"""
import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn
rng = np.random.default_rng(0)
N, M = 1_000_000, 100_000
for kind in ("correlated", "independent", "heavy-tailed"):
    for dtype in ("float64", "float32"):
        for d in (6, 7):
            names = [f"v{i}" for i in range(d)]
            if kind == "correlated":
                mix = np.eye(d) + 0.3 * rng.normal(size=(d, d))
                tr = rng.normal(size=(N, d)) @ mix; te = rng.normal(size=(M, d)) @ mix
            elif kind == "independent":
                tr = rng.normal(size=(N, d)); te = rng.normal(size=(M, d))
            else:
                tr = rng.standard_t(3, size=(N, d)); te = rng.standard_t(3, size=(M, d))
            trd = pd.DataFrame(tr.astype(dtype), columns=names); ted = pd.DataFrame(te.astype(dtype), columns=names)
            out = []
            for md in ("5", "7"):
                os.environ["PBN_PRUNE_MAX_DIMS"] = md
                k = pbn.KDE(names); k.fit(trd); k.slogl(ted)
                best = 1e9
                for _ in range(3):
                    t0 = time.perf_counter(); s = k.slogl(ted); best = min(best, time.perf_counter() - t0)
                out.append(f"max_dims {md}: {best*1e3:.1f} ms ({s:.6f})")
            print(kind, dtype, f"d={d}", " | ".join(out), flush=True)
