"""What a large device allocation costs on this box, fresh and after the memory has been used and freed by this process, and what
config 5's hill-climb costs on a context whose arenas are already grown (second run in one process):   python3 tools/alloc_probe.py"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
hip = C.CDLL("libamdhip64.so")


def malloc_ms(gb, touch=False):
    p = C.c_void_p()
    hip.hipDeviceSynchronize()
    t0 = time.perf_counter()
    rc = hip.hipMalloc(C.byref(p), C.c_size_t(int(gb * (1 << 30))))
    hip.hipDeviceSynchronize()
    t1 = time.perf_counter()
    assert rc == 0
    if touch:
        hip.hipMemset(p, 0, C.c_size_t(int(gb * (1 << 30))))
        hip.hipDeviceSynchronize()
    t2 = time.perf_counter()
    hip.hipFree(p)
    hip.hipDeviceSynchronize()
    t3 = time.perf_counter()
    return (t1 - t0) * 1e3, (t2 - t1) * 1e3, (t3 - t2) * 1e3


for rnd in ("fresh", "after use"):
    for gb in (1, 4, 8, 16, 32):
        m, s, f = malloc_ms(gb, touch=True)
        print(f"{rnd}: hipMalloc {gb:3d} GB {m:8.1f} ms, memset {s:7.1f} ms, hipFree {f:7.1f} ms", flush=True)

import torch  # noqa: E402,F401
import bench, pybnesian_amd as pbn  # noqa: E402
from pybnesian_amd import _lib  # noqa: E402
ctx = pbn.Context(0)
for i in range(2):
    out = bench.bench_hill_climb(torch, pbn, _lib, ctx, torch.device("cuda", 0), "c5mmhc", 0, 1000000, cpu=False)
    print(f"c5 run {i}: {out['estimate_s']:.3f} s, {out['cells_scored']} cells", flush=True)
