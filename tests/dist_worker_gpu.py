"""Worker of tests/test_distributed_gpu.py (not a test): one rank of a gloo world on ONE GPU.  Runs the CVLikelihood
hill-climb of a small semiparametric network - its CKDE candidates are sharded over the ranks with real device scores - and a
sharded KDE slogl; prints one JSON line."""
import json
import os
import sys

import numpy as np
import pandas as pd

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def table(n=1500, seed=4):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=n)
    b = np.tanh(1.5 * a) + rng.normal(scale=0.3, size=n)
    c = 0.8 * a - 0.5 * b + rng.normal(scale=0.5, size=n)
    d = np.sin(c) + rng.normal(scale=0.4, size=n)
    e = rng.normal(size=n)
    return pd.DataFrame({"a": a, "b": b, "c": c, "d": d, "e": e})


def run(fail_rank=-1):
    import pybnesian_amd as pbn
    from pybnesian_amd import distributed

    df = table()
    names = list(df.columns)
    score = pbn.CVLikelihood(df, 3, 0)
    hc = pbn.GreedyHillClimbing()
    start = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
    res = hc.estimate(pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), score, start, max_indegree=2)
    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    trace = [[3, op.node(), str(op.node_type())] if isinstance(op, pbn.ChangeNodeType) else [kinds[type(op)], op.source(), op.target()]
             for op in hc.last.trace]
    entries, sweeps = score.kde_cache_stats()
    kde = pbn.KDE(["a", "b", "c"])
    kde.fit(df)
    test = table(700, 9)
    # BGe / BIC on a linear-Gaussian table (score-equivalent orientations tie to the last ulps of the moments): with
    # torch.distributed initialised the Score takes row-sharded moments (distributed.reduce_moments); the world-size-invariant
    # summation must reproduce the single-process deltas bit for bit, hence the same tie decisions
    rng = np.random.default_rng(17)
    m, cols = 40000, 12
    g = np.zeros((m, cols))
    for j in range(cols):
        g[:, j] = rng.normal(scale=rng.uniform(0.5, 1.5), size=m)
        for pj in rng.choice(j, size=min(j, int(rng.integers(0, 3))), replace=False) if j else []:
            g[:, j] += rng.uniform(-1.5, 1.5) * g[:, pj]
    gdf = pd.DataFrame(g, columns=[f"g{i}" for i in range(cols)])
    gauss = {}
    for tag, sc in (("bge", pbn.BGe(gdf)), ("bic", pbn.BIC(gdf))):
        h2 = pbn.GreedyHillClimbing()
        r2 = h2.estimate(pbn.ArcOperatorSet(), sc, pbn.GaussianNetwork(list(gdf.columns)))
        gauss[tag + "_trace"] = [[kinds[type(op)], op.source(), op.target()] for op in h2.last.trace]
        gauss[tag + "_deltas"] = [op.delta().hex() for op in h2.last.trace]
        gauss[tag + "_arcs"] = sorted(r2.arcs())
    # hybrid table (CKDE children of discrete parents): the slices of every such candidate are shared by the ranks (64 fixed parts,
    # pbn_score_batch_parts) - the per-part sums added over the ranks and then in part order are the one-process score, bit for bit
    hr = np.random.default_rng(23)
    hn = 9000
    d1 = hr.integers(0, 3, size=hn)
    d2 = (hr.random(hn) < 0.35).astype(np.int64)
    x = hr.normal(size=hn) + 0.9 * d1
    y = np.tanh(x) * (1 + 0.5 * d2) + hr.normal(scale=0.4, size=hn)
    z = 0.5 * y + hr.normal(scale=0.7, size=hn) - 0.6 * d2
    hdf = pd.DataFrame({"x": x, "y": y, "z": z})
    hdf["d1"] = pd.Categorical.from_codes(d1, ["a", "b", "c"])
    hdf["d2"] = pd.Categorical.from_codes(d2, ["p", "q"])
    hs = pbn.CVLikelihood(hdf, 4, 2)
    hstart = pbn.SemiparametricBN(list(hdf.columns), [], [(n, pbn.CKDEType()) for n in "xyz"])
    h3 = pbn.GreedyHillClimbing()
    r3 = h3.estimate(pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), hs, hstart, max_indegree=3)
    hyb = {"hyb_trace": [[3, op.node(), str(op.node_type())] if isinstance(op, pbn.ChangeNodeType) else [kinds[type(op)], op.source(), op.target()]
                         for op in h3.last.trace],
           "hyb_deltas": [op.delta().hex() for op in h3.last.trace], "hyb_arcs": sorted(r3.arcs()), "hyb_sweeps": hs.kde_cache_stats()[1]}
    return {**gauss, **hyb, "trace": trace, "deltas": [op.delta() for op in hc.last.trace], "arcs": sorted(res.arcs()),
            "types": [str(res.node_type(n)) for n in names], "cells": hc.last.cells_scored, "sweeps": sweeps,
            "slogl": distributed.sharded_slogl(kde, test)}


if __name__ == "__main__":
    import torch.distributed as dist

    world = int(os.environ.get("WORLD_SIZE", "1"))
    backend = sys.argv[sys.argv.index("--backend") + 1] if "--backend" in sys.argv else "gloo"
    seen = None
    if world > 1:
        if backend == "nccl":   # RCCL over xGMI: one device per rank (PBN_DEVICE = LOCAL_RANK, set by the launcher)
            import torch

            dev = torch.device("cuda", int(os.environ["LOCAL_RANK"]))
            torch.cuda.set_device(dev)
            dist.init_process_group("nccl", device_id=dev)
            ones = torch.ones(1, dtype=torch.float64, device=dev)
            dist.all_reduce(ones)
            seen = int(ones.item())
        else:
            dist.init_process_group("gloo")
    out = run()
    out["rank"] = int(os.environ.get("RANK", "0"))
    out["backend"] = backend if world > 1 else None
    out["ranks_seen"] = seen
    print("RESULT " + json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()
