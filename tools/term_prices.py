"""Device milliseconds of one CKDE likelihood TERM A(S, m) by number of variables, table size and dtype - what csrc/shard.hip's price list
(pbn_shard_term_cost) and hybrid.hip's part prices stand for.  local(v | P) = A({v} u P) - A(P): timing local scores with 0, 1, 2, 3 parents on
fresh score handles (no cache) gives T(p) = A_(p+1) + A_p, solved for the terms in turn.  Printed beside the price list's own ratios.
    python tools/term_prices.py > profiles/rN/term_prices.txt"""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
import pybnesian_amd as pbn  # noqa: E402
from pybnesian_amd import _lib  # noqa: E402

lib = _lib.load()
lib.pbn_shard_term_cost.restype = C.c_double
ctx = pbn.Context(0)
dev = torch.device("cuda", 0)
for dt, pdt, label in ((torch.float64, _lib.PBN_F64, "fp64"), (torch.float32, _lib.PBN_F32, "fp32")):
    for rows in (100_000, 500_000):
        t = bench.make_dag_table(torch, dev, rows, 8, 2, dt, nonlinear=True)
        names = [f"x{i}" for i in range(8)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), rows, names, rows, pdt, keepalive=t)
        model = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
        T = []
        for p in range(4):
            best = None
            for rep in range(3):
                score = pbn.CVLikelihood(None, 10, 0, table=table)       # fresh caches
                ctx.sync()
                t0 = time.perf_counter()
                score.local_score_node_type(model, pbn.CKDEType(), "x7", names[:p])
                dtm = (time.perf_counter() - t0) * 1e3
                best = dtm if best is None else min(best, dtm)
            T.append(best)
        terms = [T[0]]
        for p in range(1, 4):
            terms.append(max(T[p] - terms[p - 1], 0.0))
        ntr, nte = rows - rows // 10, rows // 10
        price = [lib.pbn_shard_term_cost(d, C.c_int64(ntr), C.c_int64(nte)) for d in range(1, 5)]
        print(f"{label} {rows} rows, 10 folds: local-score ms by parents {['%.1f' % x for x in T]} -> term ms by variables {['%.1f' % x for x in terms]}; "
              f"measured ratios to d = 1: {['%.2f' % (x / terms[0]) for x in terms]}; price list's: {['%.2f' % (x / price[0]) for x in price]}", flush=True)
