"""Share of the unpruned sum-only sweeps' training tiles (per wave) that pass the guard of exp2_magic (kde_sweep_body: GUARD):
python tools/guard_open_share.py [n_train n_test d [mix]]   (mix = 1: the correlated table of bench.py's C2, lower-triangular 0.3)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
import numpy as np, pandas as pd
import pybnesian_amd as pbn
from pybnesian_amd import _lib
lib = _lib.load()
n, m, d = (int(x) for x in sys.argv[1:4]) if len(sys.argv) > 3 else (1_000_000, 100_000, 8)
mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d) if (len(sys.argv) > 4 and sys.argv[4] == "1") else np.eye(d)
rng = np.random.default_rng(0)
names = [f"v{i}" for i in range(d)]
train = pd.DataFrame(rng.normal(size=(n, d)) @ mix.T, columns=names)
test = pd.DataFrame(rng.normal(size=(m, d)) @ mix.T, columns=names)
for cls in ("ProductKDE", "KDE"):
    k = getattr(pbn, cls)(names); k.fit(train)
    lib.pbn_debug_sweep_visits(None, None, 1); lib.pbn_debug_sweep_redo(None, None, 1)
    s = k.slogl(test)
    v, t, r, u = (C.c_ulonglong(0) for _ in range(4))
    lib.pbn_debug_sweep_visits(C.byref(v), C.byref(t), 0); lib.pbn_debug_sweep_redo(C.byref(r), C.byref(u), 0)
    print(f"{cls} {n} x {m} d={d}: slogl {s:.6f}; tiles x waves {t.value}, without the clamp {v.value} ({v.value / max(t.value, 1):.4f}), overflow redos {r.value} of {u.value}")
