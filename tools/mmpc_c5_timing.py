import sys; sys.path.insert(0, '.')
import numpy as np, pandas as pd, time
import pybnesian_amd as pbn
from pybnesian_amd.independences import mmpc_cpcs
import bench, torch
n_rows = int(sys.argv[1])
rng = np.random.default_rng(3)
n_disc, n_cont = 16, 32
cards = rng.integers(2, 5, size=n_disc)
disc = {}
for j in range(n_disc):
    base = rng.integers(0, cards[j], size=n_rows)
    if j > 0:
        prev = disc[f"D{j - 1}"]
        flip = rng.random(n_rows) < 0.3
        base = np.where(flip, prev % cards[j], base)
    disc[f"D{j}"] = base.astype(np.int32)
t = bench.make_dag_table(torch, torch.device("cuda", 0), n_rows, n_cont, 3, torch.float32, nonlinear=True).cpu().numpy()
cols = {}
for j in range(n_cont):
    cols[f"x{j}"] = t[j] + 1.5 * disc[f"D{j % n_disc}"].astype(np.float32)
df = pd.DataFrame(cols)
for j in range(n_disc):
    df[f"D{j}"] = pd.Categorical.from_codes(disc[f"D{j}"], [f"c{v}" for v in range(cards[j])])
names = list(df.columns)
test = pbn.MutualInformation(df)
t_all = time.perf_counter()
t0 = time.perf_counter()
try:
    cpcs, nt = mmpc_cpcs(test, names, 0.05)
    dt = time.perf_counter() - t0
    print("ok tests", nt, "s", dt, "tests/s", nt / dt, "cpc sizes", [len(c) for c in cpcs], "passes", test.passes())
except Exception as ex:
    print("error", repr(ex), time.perf_counter() - t0, test.passes())
del test
import gc; gc.collect()
print("wall incl. handle teardown", time.perf_counter() - t_all)
