"""What the tile pruning of the fp64 sweeps leaves to do, and what the kernel makes of it: fraction of (wave, training tile)
pairs visited (one call with the PBN_SWEEP_COUNT_REDO counters on), time of the pruned slogl WITH THE COUNTERS OFF (their
per-batch atomics cost time that grows with the number of workgroups), and the time the unpruned sweep would need for the
visited fraction alone (its rate x the fraction) - 1e6 x 1e5 rows, KDE d = 1..5.  python tools/prune_visits.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import pybnesian_amd as pbn
from pybnesian_amd import _lib

lib = _lib.load()
rng = np.random.default_rng(0)
N, M = int(os.environ.get("PV_TRAIN", 1_000_000)), int(os.environ.get("PV_TEST", 100_000))
for d in (1, 2, 3, 4, 5):
    names = [f"v{i}" for i in range(d)]
    mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
    tr = rng.normal(size=(N, d)) @ mix.T
    te = rng.normal(size=(M, d)) @ mix.T
    trb = pa.RecordBatch.from_arrays([pa.array(tr[:, i]) for i in range(d)], names=names)
    teb = pa.RecordBatch.from_arrays([pa.array(te[:, i]) for i in range(d)], names=names)
    out = {}
    for prune in ("0", "1"):
        os.environ["PBN_SWEEP_PRUNE"] = prune
        k = pbn.KDE(names)
        k.fit(trb)
        k.slogl(teb)
        lib.pbn_debug_sweep_visits(None, None, 1)
        lib.pbn_debug_sweep_redo(None, None, 1)
        os.environ["PBN_SWEEP_COUNT_REDO"] = "1"   # read per call (kde_model.hip env_int)
        k.slogl(teb)
        os.environ["PBN_SWEEP_COUNT_REDO"] = "0"
        v, t, r, u = (C.c_ulonglong(0) for _ in range(4))
        lib.pbn_debug_sweep_visits(C.byref(v), C.byref(t), 0)
        lib.pbn_debug_sweep_redo(C.byref(r), C.byref(u), 0)
        best = 1e9
        for _ in range(3):
            t0 = time.perf_counter(); k.slogl(teb); best = min(best, time.perf_counter() - t0)
        out[prune] = (best, v.value, t.value, r.value, u.value)
    t0_, t1_ = out["0"][0], out["1"][0]
    frac = out["1"][1] / max(out["1"][2], 1)
    print(f"d={d}: unpruned {t0_*1e3:.1f} ms; pruned {t1_*1e3:.1f} ms; visited {frac:.3f} of the (wave, tile) pairs -> {t0_*frac*1e3:.1f} ms at the "
          f"unpruned rate; blind batches redone {out['1'][3]} of {out['1'][4]}", flush=True)
