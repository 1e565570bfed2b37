import time, numpy as np, pandas as pd, sys
sys.path.insert(0, '.')
import pybnesian_amd as pbn
rng = np.random.default_rng(0)
d = 8
mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
names = [f"v{i}" for i in range(d)]
train = pd.DataFrame(rng.normal(size=(1_000_000, d)) @ mix.T, columns=names)
test = pd.DataFrame(rng.normal(size=(100_000, d)) @ mix.T, columns=names)
k = pbn.ProductKDE(names)
k.fit(train.iloc[:1000]); k.slogl(test.iloc[:100])  # warm-up (context, first launch)
t0 = time.perf_counter(); k.fit(train); t1 = time.perf_counter()
s = k.slogl(test); t2 = time.perf_counter()
s2 = k.slogl(test); t3 = time.perf_counter()
print(f"fit(1e6x8 pandas->HBM, cov, bandwidth, pack) {t1-t0:.3f}s; slogl(1e5x8 from pandas) {t2-t1:.4f}s, again {t3-t2:.4f}s -> {1e5/(t3-t2)/1e6:.3f} M-samples/s PCIe-inclusive; slogl={s}")
