"""Workload for profiling the kernels outside the main sweep: CKDE.cdf, CKDE.sample (weights-only sweep + pick kernels) and the
hybrid MutualInformation pass.  Run as
   rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof_aux -- python3 tools/profile_aux.py"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402

rng = np.random.default_rng(0)
N, M = 1_000_000, 100_000
e = rng.normal(size=(N + M, 3))
y = e @ np.array([0.5, -1.0, 0.8]) + np.tanh(e[:, 0]) + rng.normal(scale=0.6, size=N + M)
names = ["y", "e0", "e1", "e2"]
train = pd.DataFrame(np.column_stack([y, e])[:N], columns=names)
test = pd.DataFrame(np.column_stack([y, e])[N:], columns=names)
cpd = pbn.CKDE("y", names[1:])
cpd.fit(train)
for _ in range(2):
    t0 = time.perf_counter()
    c = cpd.cdf(test)
    t_cdf = time.perf_counter() - t0
t0 = time.perf_counter()
s = cpd.sample(20_000, test.iloc[:20_000, 1:], 0)
t_sample = time.perf_counter() - t0
print(f"CKDE.cdf {M} rows vs {N} training rows (3 evidence vars): {t_cdf * 1e3:.1f} ms = {N * M / t_cdf / 1e9:.0f} Gpairs/s; "
      f"CKDE.sample 20000: {t_sample * 1e3:.1f} ms", flush=True)

n = 1_000_000
d1 = rng.integers(0, 3, size=n)
d2 = rng.integers(0, 4, size=n)
df = pd.DataFrame({f"c{i}": (rng.normal(size=n) + 0.3 * d1).astype(np.float32) for i in range(6)})
df["d1"] = pd.Categorical.from_codes(d1, ["a", "b", "c"])
df["d2"] = pd.Categorical.from_codes(d2, ["p", "q", "r", "s"])
mi = pbn.MutualInformation(df)
cases = [("c0", "c1", None), ("d1", "c0", ["c1", "c2"]), ("d1", "d2", ["c0", "c1", "c2"]), ("c0", "c1", ["d1", "d2", "c2", "c3", "c4"])]
for x, yv, z in cases:
    mi.pvalue(x, yv, z)
    t0 = time.perf_counter()
    for _ in range(20):
        mi.pvalue(x, yv, z)
    dt = (time.perf_counter() - t0) / 20
    print(f"MutualInformation {x},{yv}|{z}: {dt * 1e6:.0f} us per test at {n} rows", flush=True)
