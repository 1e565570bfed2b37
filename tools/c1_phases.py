"""Where the 6 ms of BASELINE config 1 (4-node Gaussian network, 10 k rows) go on the device path: per-phase wall times, warm."""
import os, sys, time
import numpy as np, pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: F401  (before the library touches the GPU)
import pybnesian_amd as pbn

rng = np.random.default_rng(0)
n = 10_000
a = rng.normal(size=n); b = 0.7 * a + rng.normal(scale=0.8, size=n); c = -0.5 * a + 1.2 * b + rng.normal(scale=0.6, size=n); d = 0.9 * c + rng.normal(size=n)
df = pd.DataFrame({"a": a, "b": b, "c": c, "d": d})
names = list(df.columns)
for rep in range(3):
    t = [time.perf_counter()]
    score = pbn.BIC(df); t.append(time.perf_counter())
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names)); t.append(time.perf_counter())
    res.fit(df); t.append(time.perf_counter())
    sl = res.slogl(df); t.append(time.perf_counter())
    ph = [1e3 * (t[i + 1] - t[i]) for i in range(4)]
    print("rep %d: BIC ctor %.2f ms, hill-climb %.2f ms (%d cells), fit %.2f ms, slogl %.2f ms, total %.2f ms" % (
        rep, ph[0], ph[1], hc.last.cells_scored, ph[2], ph[3], 1e3 * (t[-1] - t[0])), flush=True)
import cProfile, pstats
pr = cProfile.Profile(); pr.enable()
score = pbn.BIC(df); res = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names)); res.fit(df); res.slogl(df)
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(22)
