"""One pruned fp64 KDE handle (1e6 x 1e5 rows, d = PV_D, default 2): fit, then 5 slogl calls - to be run under
rocprofv3 --kernel-trace --stats for the per-kernel split of a pruned evaluation (sorts, pack, subsample sweep, prepass, sweep, finish)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pyarrow as pa
import pybnesian_amd as pbn

d = int(os.environ.get("PV_D", 2))
rng = np.random.default_rng(0)
N, M = 1_000_000, 100_000
names = [f"v{i}" for i in range(d)]
mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
tr = rng.normal(size=(N, d)) @ mix.T
te = rng.normal(size=(M, d)) @ mix.T
trb = pa.RecordBatch.from_arrays([pa.array(tr[:, i]) for i in range(d)], names=names)
teb = pa.RecordBatch.from_arrays([pa.array(te[:, i]) for i in range(d)], names=names)
k = pbn.KDE(names)
k.fit(trb)
for _ in range(5):
    print(k.slogl(teb))
