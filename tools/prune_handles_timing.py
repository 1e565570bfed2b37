"""KDE / CKDE handle logl + slogl at low dimension with and without tile pruning (PBN_SWEEP_PRUNE), 1e6 x 1e5 rows."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import pybnesian_amd as pbn

rng = np.random.default_rng(0)
N, M = 1_000_000, 100_000
for dtype in ("float64", "float32"):
    for d in (1, 2, 3, 4, 5):
        names = [f"v{i}" for i in range(d)]
        mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
        tr = (rng.normal(size=(N, d)) @ mix.T).astype(dtype)
        te = (rng.normal(size=(M, d)) @ mix.T).astype(dtype)
        trb = pa.RecordBatch.from_arrays([pa.array(tr[:, i]) for i in range(d)], names=names)
        teb = pa.RecordBatch.from_arrays([pa.array(te[:, i]) for i in range(d)], names=names)
        row = [dtype, f"d={d}"]
        for what in ("KDE", "CKDE"):
            if what == "CKDE" and d == 1:
                continue
            for prune in ("0", "1"):
                os.environ["PBN_SWEEP_PRUNE"] = prune
                k = pbn.KDE(names) if what == "KDE" else pbn.CKDE(names[0], names[1:])
                k.fit(trb)               # warm-up of the fit kernels of this shape (first use of a variant costs 10-100 ms)
                t0 = time.perf_counter(); k.fit(trb); tf = time.perf_counter() - t0
                k.slogl(teb); k.logl(teb)   # warm-up: arenas, first launches of these kernel variants
                ts = tl = 1e9
                for _ in range(3):       # best of 3
                    t0 = time.perf_counter(); s = k.slogl(teb); ts = min(ts, time.perf_counter() - t0)
                    t0 = time.perf_counter(); l = k.logl(teb); tl = min(tl, time.perf_counter() - t0)
                row.append(f"{what} prune={prune}: fit {tf*1e3:.1f} ms slogl {ts*1e3:.1f} ms logl {tl*1e3:.1f} ms (slogl {s:.6f})")
        print(" | ".join(row), flush=True)
