"""The C4 Gram pass on its own: 2M x 64 fp64 table resident in HBM, pbn_table_sse (pilot + gram + reduce) a few times.
Run under rocprofv3 --kernel-trace --stats for the gram kernel's duration (tools/gram_timing.sh).  PBN_GRAM_DEBUG=1 / 2 give
the kernel without its MFMAs / without its global loads (floors; the statistics are then garbage); GRAM_DTYPE=f32 times the
float table (gram_lds_kernel), PBN_GRAM_LDS=0/1 the older kernels."""
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
for _ in range(3):
    table.sse(names)
ctx.sync()
t0 = time.perf_counter()
for _ in range(5):
    table.sse(names)
ctx.sync()
print(f"pbn_table_sse {n_rows} x {n_cols}: {(time.perf_counter() - t0) / 5 * 1e6:.0f} us per call (pilot + gram + reduce + D2H)")
