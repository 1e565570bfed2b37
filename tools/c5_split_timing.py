"""C5 hill-climb: where the time outside the CKDE sweeps goes.  Every score batch is issued as two calls - the CKDE candidates, the others
(LinearGaussian / discrete factor candidates) - and timed; the rest of the run is the search loop itself (hc.hip + the Python callback)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import bench
import pybnesian_amd as pbn
from pybnesian_amd import _lib, scores

acc = {"ckde": [0.0, 0, 0], "other": [0.0, 0, 0]}
orig = scores.DeviceScore._batch_raw if hasattr(scores, "DeviceScore") else None
cls = [c for c in vars(scores).values() if isinstance(c, type) and "_batch_raw" in vars(c)][0]
orig = cls._batch_raw

def timed(self, model, var, ntype, off, par, kind):
    n = len(var)
    out = np.zeros(n)
    for tag, idx in (("ckde", [i for i in range(n) if ntype[i] == _lib.PBN_NODE_CKDE]), ("other", [i for i in range(n) if ntype[i] != _lib.PBN_NODE_CKDE])):
        if not idx:
            continue
        o, p = [0], []
        for i in idx:
            p.extend(par[off[i]: off[i + 1]])
            o.append(len(p))
        t0 = time.perf_counter()
        out[idx] = orig(self, model, [var[i] for i in idx], [ntype[i] for i in idx], o, p, kind)
        acc[tag][0] += time.perf_counter() - t0
        acc[tag][1] += 1
        acc[tag][2] += len(idx)
    return out

cls._batch_raw = timed
ctx = pbn.default_context()
res = bench.bench_hill_climb(torch, pbn, _lib, ctx, torch.device("cuda", 0), "c5", 0, 1_000_000, cpu=False)
tot = res["estimate_s"]
print(f"estimate {tot:.2f} s, cells {res['cells_scored']}: CKDE candidates {acc['ckde'][0]:.2f} s ({acc['ckde'][2]} in {acc['ckde'][1]} calls), "
      f"other candidates {acc['other'][0]:.2f} s ({acc['other'][2]} in {acc['other'][1]} calls), search loop + callbacks {tot - acc['ckde'][0] - acc['other'][0]:.2f} s")
