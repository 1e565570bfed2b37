"""Unpruned fp32 ProductKDE.slogl at 1e6 x 1e5 for several dimensions (PBN_SWEEP_PRUNE=0): sweep milliseconds from the library's HIP events.
What the matrix side of a pair value costs in WALL time: d = 4 takes one 32-slot MFMA per 16x16 tile pair, d = 8 two (bf16x3) - the exp + add work is the same."""
import os, sys, time
os.environ.setdefault("PBN_SWEEP_PRUNE", "0")
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench, pybnesian_amd as pbn
from pybnesian_amd import _lib
ctx = pbn.Context(0); dev = torch.device("cuda", 0)
dims = [int(x) for x in sys.argv[1:]] or [2, 4, 5, 8, 9, 10]
for d in dims:
    bench.D = d
    names = [f"v{i}" for i in range(d)]
    for dt, pd in ((torch.float32, _lib.PBN_F32),):
        tr, te = bench.make_tables(torch, dev, 1_000_000, 100_000, 0, 1, dt)
        torch.cuda.synchronize()
        a = pbn.DeviceTable.from_device_pointer(ctx, tr.data_ptr(), 1_000_000, names, 1_000_000, pd, keepalive=tr)
        b = pbn.DeviceTable.from_device_pointer(ctx, te.data_ptr(), 100_000, names, 100_000, pd, keepalive=te)
        k = pbn.ProductKDE(names); k.fit_table(a)
        buf = torch.zeros(16, dtype=torch.float64, device=dev)
        for i in range(3): k.slogl_table_async(b, buf.data_ptr() + 8 * i)
        ctx.sync(); ctx.set_profiling(True)
        for i in range(8): k.slogl_table_async(b, buf.data_ptr() + 8 * (3 + i))
        ctx.sync()
        ms, n = ctx.kernel_time(_lib.PBN_K_SWEEP); ctx.set_profiling(False)
        print(f"d={d} f32: {ms / max(n, 1):.3f} ms per sweep ({n} launches), slogl {float(buf[3].item()):.6f}", flush=True)
