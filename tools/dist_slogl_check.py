"""2-rank check of pybnesian_amd.distributed.sharded_slogl on ONE GPU (gloo), run with
   PBN_DEVICE=0 python -m torch.distributed.run --nproc-per-node 2 --master-addr 127.0.0.1 tools/dist_slogl_check.py"""
import os
import sys

import numpy as np
import pandas as pd
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn  # noqa: E402
from pybnesian_amd.distributed import sharded_slogl  # noqa: E402

dist.init_process_group("gloo")
rng = np.random.default_rng(0)
train = pd.DataFrame(rng.normal(size=(20000, 3)), columns=list("abc"))
test = pd.DataFrame(rng.normal(size=(3001, 3)), columns=list("abc"))
for f in (pbn.KDE(list("abc")), pbn.CKDE("a", ["b", "c"])):
    f.fit(train)
    whole = f.slogl(test)
    got = sharded_slogl(f, test)
    assert abs(got - whole) <= 1e-12 * abs(whole), (got, whole)
if dist.get_rank() == 0:
    print("sharded_slogl ok on", dist.get_world_size(), "ranks")
dist.destroy_process_group()
