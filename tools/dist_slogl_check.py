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

rng = np.random.default_rng(0)
# row-sharded Gram moments of the scores: reference values from unsharded handles, built BEFORE the process group exists
sdf = pd.DataFrame(rng.normal(size=(50001, 5)) @ (np.eye(5) + np.triu(np.full((5, 5), 0.3), 1)), columns=list("vwxyz"))
cands = [("v", []), ("w", ["v"]), ("z", ["v", "w", "x", "y"])]
single = {}
for name, mk in (("bge", lambda: pbn.BGe(sdf)), ("bic", lambda: pbn.BIC(sdf)), ("cv", lambda: pbn.CVLikelihood(sdf, k=5, seed=1))):
    sc = mk()
    single[name] = [sc.local_score(pbn.GaussianNetwork(list("vwxyz")), v, e) for v, e in cands]
# MMPC with the hybrid MutualInformation test: single-process CPCs first
from pybnesian_amd.independences import mmpc_cpcs  # noqa: E402

hn = 30000
hd = rng.integers(0, 3, size=hn)
hdf = pd.DataFrame({"h0": rng.normal(size=hn) + hd, "h1": rng.normal(size=hn), "h2": rng.normal(size=hn)})
hdf["h1"] += 0.8 * hdf["h0"]
hdf["h2"] -= 0.6 * hdf["h1"]
hdf["hd"] = pd.Categorical.from_codes(hd, ["a", "b", "c"])
single_cpcs = mmpc_cpcs(pbn.MutualInformation(hdf), list(hdf.columns), 0.05)
dist.init_process_group("gloo")
mi_test = pbn.MutualInformation(hdf)
sharded_cpcs = mmpc_cpcs(mi_test, list(hdf.columns), 0.05)      # test batches dealt over the ranks, one all_gather each
assert sharded_cpcs == single_cpcs, (sharded_cpcs, single_cpcs)
assert mi_test.passes()[0] < single_cpcs[1]
for name, mk in (("bge", lambda: pbn.BGe(sdf)), ("bic", lambda: pbn.BIC(sdf)), ("cv", lambda: pbn.CVLikelihood(sdf, k=5, seed=1))):
    sc = mk()   # now sharded: each rank takes the Gram of half of every region, one all_gather of the moments
    got = [sc.local_score(pbn.GaussianNetwork(list("vwxyz")), v, e) for v, e in cands]
    assert np.allclose(got, single[name], rtol=1e-11, atol=0), (name, got, single[name])
train = pd.DataFrame(rng.normal(size=(20000, 3)), columns=list("abc"))
test = pd.DataFrame(rng.normal(size=(3001, 3)), columns=list("abc"))
for f in (pbn.KDE(list("abc")), pbn.CKDE("a", ["b", "c"])):
    f.fit(train)
    whole = f.slogl(test)
    got = sharded_slogl(f, test)
    assert abs(got - whole) <= 1e-12 * abs(whole), (got, whole)
if dist.get_rank() == 0:
    print("row-sharded moments + sharded MMPC + sharded_slogl ok on", dist.get_world_size(), "ranks")
dist.destroy_process_group()
