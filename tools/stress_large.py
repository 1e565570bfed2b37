"""Ad-hoc size stress: index arithmetic beyond 2^31 pairs-per-launch scales, large tile counts, both dtypes."""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402
from pybnesian_amd import _lib  # noqa: E402

ctx = pbn.Context(0)
dev = torch.device("cuda", 0)
for dtype, n_train, n_test, d in ((torch.float32, 8_000_000, 1_000_000, 8), (torch.float64, 4_000_000, 300_000, 3), (torch.float64, 3_000_001, 70_001, 16)):
    g = torch.Generator(device=dev)
    g.manual_seed(0)
    tr = torch.randn((d, n_train), generator=g, device=dev, dtype=dtype)
    te = torch.randn((d, n_test), generator=g, device=dev, dtype=dtype)
    names = [f"v{i}" for i in range(d)]
    code = _lib.PBN_F64 if dtype == torch.float64 else _lib.PBN_F32
    ttab = pbn.DeviceTable.from_device_pointer(ctx, tr.data_ptr(), n_train, names, n_train, code, keepalive=tr)
    qtab = pbn.DeviceTable.from_device_pointer(ctx, te.data_ptr(), n_test, names, n_test, code, keepalive=te)
    kde = pbn.ProductKDE(names)
    kde.fit_table(ttab)
    t0 = time.perf_counter()
    whole = kde.slogl_table(qtab)
    dt = time.perf_counter() - t0
    h = n_test // 2
    parts = kde.slogl_table(qtab, row0=0, n=h) + kde.slogl_table(qtab, row0=h, n=n_test - h)
    assert abs(whole - parts) <= (1e-11 if dtype == torch.float64 else 1e-8) * abs(whole), (whole, parts)
    # a sample of rows against the closed form for a product KDE of N(0,1) data is not available; check a tiny prefix
    # against a direct numpy evaluation instead
    m = 3
    x = te[:, :m].T.double().cpu().numpy()
    bw = np.asarray(kde.bandwidth, dtype=np.float64)
    ref = []
    trn = tr.double()
    for q in range(m):
        z = ((trn - torch.tensor(x[q], device=dev)[:, None]) ** 2 / torch.tensor(bw, device=dev)[:, None]).sum(0)
        ref.append(float(torch.logsumexp(-0.5 * z, 0)) - 0.5 * np.log(bw).sum() - 0.5 * d * np.log(2 * np.pi) - np.log(n_train))
    got = kde.slogl_table(qtab, row0=0, n=m)
    tol = 1e-9 if dtype == torch.float64 else 2e-4
    assert abs(got - sum(ref)) <= tol * abs(sum(ref)), (got, sum(ref))
    print(f"{dtype} train {n_train} test {n_test} d {d}: slogl {whole:.6f} in {dt * 1e3:.0f} ms = {n_train * n_test / dt / 1e12:.2f} Tpairs/s ok", flush=True)
