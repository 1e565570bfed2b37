"""Race screen for the LDS-DMA Gram kernels: random shapes (rows, columns, first row, dtype), every shape evaluated several times
and compared with numpy.  The kernels count their DMA completions by hand (wait_vmcnt); a miscount shows as a rare wrong
tile that comes and goes with the shape and the load of the machine - hence many shapes and repeats.  python tools/gram_fuzz.py [cases]"""
import os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
rng = np.random.default_rng(2024)
ctx = pbn.default_context()
worst = 0.0
bad = 0
t0 = time.time()
for case in range(cases):
    d = int(rng.integers(1, 65))
    n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20000)), int(rng.integers(20000, 400000))]))
    dtype = "float64" if rng.random() < 0.6 else "float32"
    data = (rng.normal(size=(n, d)) * rng.uniform(0.5, 3.0, size=d) + rng.uniform(-50, 50, size=d)).astype(dtype)
    names = [f"x{i}" for i in range(d)]
    table, _ = pbn.DeviceTable.from_dataframe(ctx, pd.DataFrame(data, columns=names), names)
    for rep in range(4):
        row0 = int(rng.integers(0, max(1, n // 3)))
        rows = int(rng.integers(1, n - row0 + 1))
        k = int(rng.integers(1, d + 1))
        cols = [names[i] for i in rng.choice(d, size=k, replace=False)]
        idx = [names.index(c) for c in cols]
        x = data[row0:row0 + rows][:, idx].astype(np.float64)
        mean = x.mean(axis=0)
        c = x - mean
        want = c.T @ c
        means, sse = table.sse(cols, row0, rows)
        scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1.0
        err = max(np.max(np.abs(sse - want) / scale), np.max(np.abs(means - mean) / (np.abs(mean) + 1.0)))
        worst = max(worst, err)
        if not err < 1e-9:
            bad += 1
            print(f"MISMATCH case {case} rep {rep}: n {n} d {d} {dtype} row0 {row0} rows {rows} k {k}: {err:.3e}", flush=True)
print(f"{cases} shapes x 4 ranges in {time.time() - t0:.1f} s: {'all ok' if bad == 0 else str(bad) + ' MISMATCHES'}; worst relative difference {worst:.3e}")
