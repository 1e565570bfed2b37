"""The RCCL path of the sharded delta cache executed at world size 1 on ONE GPU (a fresh process: bench.py and
tests/test_distributed_gpu.py start it with subprocess, nothing is exec'ed after the GPU was touched).

RCCL allows a communicator of one rank.  With PBN_FORCE_DIST the product takes its one-process-per-GPU path at world size 1: every
delta-cache batch is planned by csrc/shard.hip, this rank is dealt everything, and the batch's all-gather really runs through
torch.distributed's "nccl" backend (= RCCL) on device buffers (distributed._all_gather: pinned staging -> device -> all_gather_into_tensor ->
pinned -> host).  Checked here, bit for bit against the plain one-process calls made first in the same process:
  * distributed._all_gather of a vector (identity at world 1), an all_reduce of ones (`rccl_ranks_seen`);
  * BIC / BGe local scores on row-sharded moments (pbn_scoredata_create_sharded + pbn_scoredata_reduce_moments);
  * a CV-likelihood CKDE hill-climb (terms dealt, gathered, installed: pbn_score_batch on a handle with a communicator), a hybrid one
    (slice parts gathered) and a sharded KDE slogl (pbn_kde_slogl_sharded);
and which libamdhip64 / librccl / libpbn_hip this process mapped (the library is built against /opt/rocm's HIP headers while torch brings
its own runtime: the N > 1 mode runs on whatever /proc/self/maps shows here).  Prints one JSON line `RESULT {...}`; exit code 0 = all equal."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def mapped_libraries():
    seen = {}
    with open("/proc/self/maps") as f:
        for line in f:
            path = line.split()[-1] if "/" in line else ""
            base = os.path.basename(path)
            for tag in ("libamdhip64", "librccl", "libpbn_hip", "libhsa-runtime64"):
                if base.startswith(tag):
                    seen.setdefault(tag, set()).add(os.path.realpath(path))
    return {k: sorted(v) for k, v in seen.items()}


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29611")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch
    import torch.distributed as dist

    t_start = time.perf_counter()
    dev = torch.device("cuda", int(os.environ.get("PBN_DEVICE", "0")))
    torch.cuda.set_device(dev)
    import pybnesian_amd as pbn
    from pybnesian_amd import distributed
    from dist_worker_gpu import run as worker_run

    out = {"world": 1, "backend": "nccl"}
    ok = False
    plain = worker_run()                       # no process group yet: the one-process values
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)
    try:
        ones = torch.ones(1, dtype=torch.float64, device=dev)
        dist.all_reduce(ones)
        out["rccl_ranks_seen"] = int(round(float(ones.item())))
        assert distributed.comm() is None      # world 1 without the force switch: the plain path
        distributed.FORCE = True
        cm = distributed.comm()
        assert cm is not None and cm.struct.world == 1 and dist.get_backend() == "nccl"
        v = np.random.default_rng(0).normal(size=1000)
        got = distributed._all_gather(dist, v)
        out["all_gather_identity"] = bool(got.shape == v.shape and (got == v).all())
        g0 = cm.gathers
        forced = worker_run()                  # every score built now shards its moments and binds the communicator
        out["collectives"] = cm.gathers - g0
        same = {k: plain[k] == forced[k] for k in plain}
        out["bit_identical"] = same
        out["hill_climb_batches_through_rccl"] = out["collectives"]
        ok = out["all_gather_identity"] and out["rccl_ranks_seen"] == 1 and all(same.values()) and out["collectives"] >= 10
    finally:
        distributed.FORCE = False
        out["mapped"] = mapped_libraries()
        out["torch"] = torch.__version__
        out["hip_runtime_of_torch"] = torch.version.hip
        # The SUPPORTED pairing of an N > 1 job under torch (INTEGRATION.md "one HIP runtime per process"): ONE libamdhip64 in the process - the
        # first one loaded, i.e. torch's - and RCCL from the same distribution; libpbn_hip.so names libamdhip64.so.7 by SONAME, so the loader hands
        # it the copy torch already mapped (its RUNPATH to /opt/rocm only matters in a process without torch: the plain-C hosts).  Asserted here,
        # not only recorded: two HIP runtimes in one process, or RCCL from another tree than the HIP runtime, fail the run.
        hip, rccl = out["mapped"].get("libamdhip64", []), out["mapped"].get("librccl", [])
        one_runtime = len(hip) == 1 and len(rccl) == 1 and os.path.dirname(hip[0]) == os.path.dirname(rccl[0])
        out["hip_runtime_pairing"] = ("torch" if one_runtime and "torch" in hip[0] else "rocm" if one_runtime else "MIXED")
        ok = ok and one_runtime
        dist.destroy_process_group()
    out["seconds"] = time.perf_counter() - t_start
    out["ok"] = bool(ok)
    print("RESULT " + json.dumps(out), flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
