"""Pruned fp32 sweeps at the slice sizes of config C5 (per-configuration folds of a 1e6-row hybrid table): whole-evaluation
time unpruned / pruned, fraction of (wave, tile) pairs visited and the time the unpruned sweep would need for that fraction -
how far the pruned kernels are from their visit-bound rate when the slice no longer fills the GPU.  python tools/slice_visits.py"""
import ctypes as C, os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import pybnesian_amd as pbn
from pybnesian_amd import _lib
from pybnesian_amd.dataset import default_context

lib = _lib.load()
rng = np.random.default_rng(0)
ft = np.float32 if os.environ.get("PV_DTYPE", "f32") == "f32" else np.float64
for ckde, d in ((0, 1), (0, 2), (1, 2), (1, 3), (1, 4)):
    for N, M in ((720_000, 80_000), (360_000, 40_000), (240_000, 26_667), (180_000, 20_000)):
        names = [f"v{i}" for i in range(d)]
        mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
        tr = (rng.normal(size=(N, d)) @ mix.T).astype(ft)
        te = (rng.normal(size=(M, d)) @ mix.T).astype(ft)
        trb = pa.RecordBatch.from_arrays([pa.array(tr[:, i]) for i in range(d)], names=names)
        teb = pa.RecordBatch.from_arrays([pa.array(te[:, i]) for i in range(d)], names=names)
        out = {}
        for prune in ("0", "1"):
            os.environ["PBN_SWEEP_PRUNE"] = prune
            k = pbn.CKDE(names[0], names[1:]) if ckde else pbn.KDE(names)
            k.fit(trb)
            k.slogl(teb)
            lib.pbn_debug_sweep_visits(None, None, 1)
            os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
            k.slogl(teb)
            os.environ["PBN_SWEEP_COUNT_REDO"] = "0"
            v, t = C.c_ulonglong(0), C.c_ulonglong(0)
            lib.pbn_debug_sweep_visits(C.byref(v), C.byref(t), 0)
            best = 1e9
            for _ in range(5):
                t0 = time.perf_counter(); k.slogl(teb); best = min(best, time.perf_counter() - t0)
            ctx = default_context()
            ctx.set_profiling(True)
            ms0, n0 = ctx.kernel_time(1)
            for _ in range(3):
                k.slogl(teb)
            ctx.sync()
            ms1, n1 = ctx.kernel_time(1)
            ctx.set_profiling(False)
            out[prune] = (best, v.value, t.value, (ms1 - ms0) / max(n1 - n0, 1), (n1 - n0) // 3)
        t0_, t1_ = out["0"][0], out["1"][0]
        frac = out["1"][1] / max(out["1"][2], 1)
        k0, k1 = out["0"][3], out["1"][3]
        print(f"{'CKDE' if ckde else 'KDE '} d={d} {N:7d} x {M:6d}: evaluation unpruned {t0_*1e3:6.2f} ms, pruned {t1_*1e3:6.2f} ms; sweep kernel "
              f"unpruned {k0:6.3f} ms, pruned {k1:6.3f} ms x {out['1'][4]}; visited {frac:.3f} -> {k0*frac:6.3f} ms at the unpruned rate", flush=True)
