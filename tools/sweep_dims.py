"""Sweep time against the number of dimensions (fp64 KDE.slogl, training x test resident in HBM):
   python tools/sweep_dims.py [n_train] [n_test]      (PBN_SWEEP_FOLD=0 disables the norm-in-a-free-K-slot variant)"""
import os
import sys
import time

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402
from pybnesian_amd import DeviceTable, default_context  # noqa: E402

n_train = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
n_test = int(sys.argv[2]) if len(sys.argv) > 2 else 100_000
rng = np.random.default_rng(0)
names = [f"v{i}" for i in range(8)]
mix = np.eye(8) + 0.3 * rng.normal(size=(8, 8))
train = pd.DataFrame(rng.normal(size=(n_train, 8)) @ mix, columns=names)
test = pd.DataFrame(rng.normal(size=(n_test, 8)) @ mix, columns=names)
ctx = default_context()
ttrain, _ = DeviceTable.from_dataframe(ctx, train, names)
ttest, _ = DeviceTable.from_dataframe(ctx, test, names)
print(f"fold={os.environ.get('PBN_SWEEP_FOLD', '1')}  n_train={n_train} n_test={n_test}")
for cls in (pbn.KDE, pbn.CKDE):
    for d in range(1, 9):
        k = cls(names[:d]) if cls is pbn.KDE else cls(names[0], names[1:d])
        k.fit_table(ttrain) if cls is pbn.CKDE else k.fit(train)
        vals = []
        for rep in range(4):
            t0 = time.perf_counter()
            s = k.slogl_table(ttest)
            vals.append(time.perf_counter() - t0)
        print(f"{cls.__name__:5s} d={d}  {1e3 * min(vals[1:]):8.2f} ms   slogl={s:.10f}")
