"""Estimate of the weak-scaling leg (bench.py secondary_cv_weak) on ONE GPU: the N-rank job's CKDE shares are computed one after the
other by one process (tools/shard_emulate.py: the library's own plan, every rank played in turn) and timed per rank.  An N-rank job waits per batch for its slowest
share, and every rank repeats the unsharded work (constructor, LinearGaussian candidates, the search's own host logic):
    T_N ~ (wall - sum of all shares) + sum over batches of the slowest share          (collective latency not included)
    efficiency ~ T_1(nodes(1)) / T_N(nodes(N)),   nodes(N) = round(64 sqrt(N / 8))  - cells per rank constant.
Caveat: one process owns every set-function cache entry, an N-rank job only those of the sets it was dealt.
    python3 tools/scale_emulate.py [worlds, default 1,2,4,8] [max_iters, default 5] [rows, default 100000]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pybnesian_amd as pbn
from pybnesian_amd import _lib
from shard_emulate import EmulatedRanks
import contextlib

worlds = [int(w) for w in (sys.argv[1] if len(sys.argv) > 1 else "1,2,4,8").split(",")]
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 5
rows = int(sys.argv[3]) if len(sys.argv) > 3 else 100_000
# SCALE_WHICH=c5: BASELINE config 5's hill-climb (hybrid candidates, dealt whole by variable set) with FIXED work on every world size
which = os.environ.get("SCALE_WHICH", "cv64")


dev = torch.device("cuda", 0)
ctx = pbn.default_context()
base = None
for w in worlds:
    nodes = round(64 * (w / 8) ** 0.5) if which == "cv64" else 48
    fake = EmulatedRanks(w)
    t0 = time.perf_counter()
    with (fake if w > 1 else contextlib.nullcontext()):
        if which == "cv64":
            res = bench.bench_hill_climb(torch, pbn, _lib, ctx, dev, "cv64", rows, iters, n_cols=nodes, cpu=False)
        else:
            res = bench.bench_hill_climb(torch, pbn, _lib, ctx, dev, "c5", 0, 1_000_000, cpu=False)
    wall = time.perf_counter() - t0
    est = res["estimate_s"] + res.get("score_ctor_s", 0.0)
    if w == 1:
        t_n, shares, slow = est, 0.0, 0.0
    else:
        shares = sum(sum(t) for t in fake.batches)
        slow = sum(max(t) for t in fake.batches)
        t_n = est - shares + slow
    if base is None:
        base = (w, t_n)
    imb = (slow / (shares / w)) if shares else 1.0
    print(f"world {w}: {nodes} nodes, cells {res['cells_scored']}, one process {est:.2f} s; shares {shares:.2f} s in {len(fake.batches)} batches, slowest-share sum {slow:.2f} s "
          f"(imbalance {imb:.3f}), unsharded {est - shares:.2f} s -> T_N ~ {t_n:.2f} s, efficiency vs world {base[0]} ~ {base[1] / t_n:.2f}", flush=True)
    if w > 1 and os.environ.get("SCALE_DETAIL"):
        for t in fake.batches:
            print("   batch: " + " ".join(f"{x * 1e3:7.1f}" for x in t) + " ms")
