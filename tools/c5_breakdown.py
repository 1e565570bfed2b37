"""Where config C5's MMPC-restricted hill-climb spends its time: kernel-class totals (HIP events on the context stream)
beside the wall time.  python tools/c5_breakdown.py [max_iters] [config: c5mmhc (default), c3, c4, cv64]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pybnesian_amd as pbn
from pybnesian_amd import _lib
from pybnesian_amd.dataset import default_context

max_iters = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
config = sys.argv[2] if len(sys.argv) > 2 else "c5mmhc"
device = torch.device("cuda:0")
ctx = default_context()
ctx.set_profiling(True)
out = bench.bench_hill_climb(torch, pbn, _lib, ctx, device, config, None, max_iters)
ctx.sync()
names = {0: "pack", 1: "sweep", 2: "finish", 3: "gram"}
for k, nm in names.items():
    ms, n = ctx.kernel_time(k)
    print(f"{nm:7s} {ms / 1e3:8.2f} s  {n:8d} launches")
print({k: out[k] for k in ("estimate_s", "mmpc_s", "cells_scored", "local_score_evals", "iterations", "arcs_found", "score_ctor_s") if k in out})
