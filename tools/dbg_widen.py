import os, sys, time
sys.path.insert(0, os.getcwd())
import numpy as np, pandas as pd
import torch
import pybnesian_amd as pbn
from oracle import oracle
rng = np.random.default_rng(77)
n, m, d = 40_000, 600, 3
def draw(k):
    t = rng.normal(size=(k, 1))
    return (t @ np.ones((1, d)) + rng.normal(scale=0.02, size=(k, d))).astype(np.float32)
names = [f"v{i}" for i in range(d)]
train, test = pd.DataFrame(draw(n), columns=names), pd.DataFrame(draw(m), columns=names)
tr64, te64 = train.to_numpy().astype(np.float64), test.to_numpy().astype(np.float64)
for w in ("1", "0"):
    os.environ["PBN_F32_WIDEN"] = w
    a = pbn.ProductKDE(names); a.fit(train)
    bw = np.asarray(a.bandwidth)
    truth = oracle.product_kde_logl(tr64, bw, te64)
    t0 = time.perf_counter(); got = a.logl(test); dt = time.perf_counter() - t0
    z2 = ((tr64 - tr64.mean(0)) ** 2 / bw).sum(1).max() * 1.4427
    print("widen", w, "bw", bw, "max z2", z2, "err", np.abs(got - truth).max(), "dtype", a.data_type(), "ms", dt * 1e3)
    b = pbn.KDE(names); b.fit(train); b.bandwidth = np.eye(d) * 4e-5
    truth = oracle.kde_logl(tr64, np.eye(d) * 4e-5, te64)
    fin = np.isfinite(truth)
    got = b.logl(test)
    print("  KDE user bw err", np.abs(got[fin] - truth[fin]).max(), fin.sum())
