"""Visited fraction of the pruned fp32 sweeps over config 5's hill-climb (PBN_SWEEP_COUNT_REDO counters; slower with them on):
python3 tools/c5_visits.py [max_iters]"""
import ctypes as C, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
import torch  # noqa: F401
import bench, pybnesian_amd as pbn
from pybnesian_amd import _lib
lib = _lib.load()
ctx = pbn.Context(0)
lib.pbn_debug_sweep_visits(None, None, 1)
out = bench.bench_hill_climb(torch, pbn, _lib, ctx, torch.device("cuda", 0), "c5mmhc", 0, int(sys.argv[1]) if len(sys.argv) > 1 else 1000000, cpu=False)
v, t = C.c_ulonglong(0), C.c_ulonglong(0)
lib.pbn_debug_sweep_visits(C.byref(v), C.byref(t), 0)
print("hill-climb %.2f s (counters on), %d cells; (wave, tile) pairs visited %.4e of %.4e offered = %.3f" % (out["estimate_s"], out["cells_scored"], v.value, t.value, v.value / max(t.value, 1)))
