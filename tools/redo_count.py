"""How often the unchecked first pass of the fp64 sweep has to redo a split (PBN_SWEEP_COUNT_REDO=1): bench.py's C2 workload."""
import ctypes as C, os, sys
os.environ["PBN_SWEEP_COUNT_REDO"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import bench
import pybnesian_amd as pbn
from pybnesian_amd import _lib
dev = torch.device("cuda", 0)
ctx = pbn.Context(0)
names = [f"v{i}" for i in range(8)]
tr, te = bench.make_tables(torch, dev, 1_000_000, 100_000, 0, 1, torch.float64)
torch.cuda.synchronize()
train = pbn.DeviceTable.from_device_pointer(ctx, tr.data_ptr(), 1_000_000, names, 1_000_000, _lib.PBN_F64, keepalive=tr)
test = pbn.DeviceTable.from_device_pointer(ctx, te.data_ptr(), 100_000, names, 100_000, _lib.PBN_F64, keepalive=te)
lib = _lib.load()
for cls in (pbn.ProductKDE, pbn.KDE):
    k = cls(names); k.fit_table(train)
    lib.pbn_debug_sweep_redo(None, None, 1)
    s = k.slogl_table(test)
    r, u = C.c_ulonglong(0), C.c_ulonglong(0)
    lib.pbn_debug_sweep_redo(C.byref(r), C.byref(u), 0)
    print(cls.__name__, "slogl", s, "redo units", r.value, "of", u.value)
