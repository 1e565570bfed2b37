"""KMutualInformation timings: python tools/kmi_timing.py [rows]  (k = 10, 1000 permutations)."""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
rng = np.random.default_rng(0)
a = rng.normal(size=n)
df = pd.DataFrame({"a": a, "b": 0.5 * a + rng.normal(size=n), "c": rng.normal(size=n), "d": rng.normal(size=n), "e": rng.normal(size=n)})
test = pbn.KMutualInformation(df, 10, seed=0, samples=1000)
for x, y, z in (("a", "b", None), ("a", "b", "c"), ("a", "b", ["c", "d", "e"])):
    test.mi(x, y, z)
    t0 = time.perf_counter()
    for _ in range(10):
        v = test.mi(x, y, z)
    t1 = time.perf_counter()
    p = test.pvalue(x, y, z)
    t2 = time.perf_counter()
    print(f"rows {n}  z={z}:  mi {v:.5f} in {(t1 - t0) / 10 * 1e3:.2f} ms   pvalue {p} (1000 permutations) in {t2 - t1:.2f} s")
