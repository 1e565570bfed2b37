"""KMutualInformation at scale (round 5, verdict item 9): one pbn_kmi_value and permutation p-values at N = 1e5 and 1e6, x _||_ y | z
(3 variables), k = 10.  The device evaluation is brute force over all N^2 pairs on integer ranks (kmi.hip: two kernels per evaluation -
k-th neighbour distances in the joint space, then the strictly-inside counts of the three subspaces); the reference walks kd-trees
(learning/independences/continuous/mutual_information.cpp, O(N log N) per evaluation on one core).
    python tools/kmi_scale.py  -> profiles/rN/kmi_timing.txt"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402

for n, samples in ((100_000, 100), (1_000_000, 12)):
    rng = np.random.default_rng(0)
    a = rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": 0.5 * a + rng.normal(size=n), "c": 0.3 * a + rng.normal(size=n)})
    t0 = time.perf_counter()
    test = pbn.KMutualInformation(df, 10, seed=0, samples=samples)
    t_ctor = time.perf_counter() - t0
    test.mi("a", "b", "c")          # warm
    reps = 3 if n <= 100_000 else 1
    t0 = time.perf_counter()
    for _ in range(reps):
        v = test.mi("a", "b", "c")
    dt = (time.perf_counter() - t0) / reps
    pairs = float(n) * float(n)
    t0 = time.perf_counter()
    p = test.pvalue("a", "b", "c")
    dp = time.perf_counter() - t0
    print(f"N = {n}: constructor (ranks, upload) {t_ctor:.2f} s; mi(a, b | c) = {v:.5f} in {dt * 1e3:.1f} ms = {pairs / dt:.3e} row pairs/s "
          f"(each pair visited by both kernels); pvalue with {samples} permutations {dp:.2f} s = {dp / samples * 1e3:.1f} ms per permuted sample "
          f"-> 1000 permutations ~ {dp / samples * 1000:.0f} s", flush=True)
