"""The C4 Gram pass on its own: 2M x 64 fp64 table resident in HBM, pbn_table_sse (pilot + gram + reduce) a few times.
Run under rocprofv3 --kernel-trace --stats for the gram kernel's duration (tools/gram_timing.sh).  PBN_GRAM_DEBUG=1 / 2 give
the kernel without its MFMAs / without its global loads (floors; the statistics are then garbage); GRAM_DTYPE=f32 times the
float table (gram_lds_kernel), PBN_GRAM_LDS=0/1 the older kernels.
GRAM_MODE=segments times the product's score-data constructor (pbn_scoredata_create: the segmented Gram of BGe / BIC, C4's constructor);
GRAM_MODE=gather the gathered, segmented Gram of a MutualInformation grouping (all continuous columns per configuration of GRAM_CARD
categories through the grouping's row list, mi.hip ensure_full) - GRAM_GROUPS discrete columns, one grouping each."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import pybnesian_amd as pbn
from pybnesian_amd import _lib

n_rows, n_cols = int(os.environ.get("GRAM_ROWS", 2_000_000)), int(os.environ.get("GRAM_COLS", 64))
dev = torch.device("cuda", 0)
g = torch.Generator(device=dev); g.manual_seed(0)
f32 = os.environ.get("GRAM_DTYPE", "f64") == "f32"
t = torch.randn((n_cols, n_rows), generator=g, device=dev, dtype=torch.float32 if f32 else torch.float64)
torch.cuda.synchronize()
ctx = pbn.Context(0)
names = [f"x{i}" for i in range(n_cols)]
table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F32 if f32 else _lib.PBN_F64, keepalive=t)
mode = os.environ.get("GRAM_MODE", "sse")
if mode == "segments":
    import ctypes as C
    lib = _lib.load()
    def ctor():
        h = C.c_void_p()
        _lib.check(lib.pbn_scoredata_create(ctx.handle, table.handle, 0, 0, 0, C.c_double(0.0), C.byref(h)))
        lib.pbn_scoredata_destroy(h)
    for _ in range(3):
        ctor()
    ctx.sync()
    t0 = time.perf_counter()
    for _ in range(5):
        ctor()
    ctx.sync()
    print(f"pbn_scoredata_create {n_rows} x {n_cols}: {(time.perf_counter() - t0) / 5 * 1e6:.0f} us per call (pilot + segmented gram + reduce + D2H + host)")
    sys.exit(0)
if mode == "gather":
    import ctypes as C
    import numpy as np
    lib = _lib.load()
    card, groups = int(os.environ.get("GRAM_CARD", 4)), int(os.environ.get("GRAM_GROUPS", 8))
    rng = np.random.default_rng(1)
    codes = [np.ascontiguousarray(rng.integers(0, card, n_rows), dtype=np.int32) for _ in range(groups)]
    ptrs = (C.c_void_p * groups)(*[c.ctypes.data for c in codes])
    h = C.c_void_p()
    _lib.check(lib.pbn_mi_create(ctx.handle, table.handle, n_rows, groups, ptrs, _lib.int_array([card] * groups), 1, C.byref(h)))
    mi, df = C.c_double(0.0), C.c_double(0.0)
    t0 = time.perf_counter()
    for g in range(groups):   # x0 _|_ x1 | d_g: builds grouping {d_g} and its full Gram (one gathered segmented launch each)
        _lib.check(lib.pbn_mi_value(h, 0, 1, 1, _lib.int_array([n_cols + g]), C.byref(mi), C.byref(df)))
    ctx.sync()
    print(f"pbn_mi_value with a new grouping, {n_rows} x {n_cols} + {card} categories: {(time.perf_counter() - t0) / groups * 1e3:.2f} ms per grouping "
          f"(sort + gathered gram + D2H + host), mi = {mi.value:.3e}")
    lib.pbn_mi_destroy(h)
    sys.exit(0)
for _ in range(3):
    table.sse(names)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    table.sse(names)
ctx.sync()
print(f"pbn_table_sse {n_rows} x {n_cols}: {(time.perf_counter() - t0) / 5 * 1e6:.0f} us per call (pilot + gram + reduce + D2H)")
