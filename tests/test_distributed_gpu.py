"""GPU tier, world_size 2 over gloo on ONE GPU: the N > 1 path of the delta-score cache with REAL device scores
(shards learning/operators/operators.cpp:100-132).  Two ranks run the CVLikelihood hill-climb of a
semiparametric network: the unknown CKDE TERMS of every batch (A(S, m) of local = A(joint) - A(marginal)) are dealt to the ranks,
each rank sweeps only its share on the device, one all_gather per batch hands every rank all term totals, every rank assembles the
deltas from the same doubles - the very sums the one-process run forms - and both ranks must reproduce the single-process run bit
for bit, without sweeping any term twice.  The ranks are separate child processes (tests/dist_worker_gpu.py)."""
import json
import os
import socket
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _launch(world, backend="gloo"):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank),
                   PBN_DEVICE="0" if backend == "gloo" else str(rank),   # gloo on one GPU: every rank's context on device 0; nccl: one device per rank
                   HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist_worker_gpu.py"), "--backend", backend], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = []
    results = [p.communicate(timeout=600) for p in procs]
    assert all(p.returncode == 0 for p in procs), "\n----\n".join(e[-1500:] for _, e in results)
    for o, e in results:
        line = [l for l in o.splitlines() if l.startswith("RESULT ")][-1]
        outs.append(json.loads(line[len("RESULT "):]))
    return outs


def _device_count():
    import torch

    return torch.cuda.device_count()   # does not initialise the GPU in this process


@pytest.mark.parametrize("backend", ["gloo", "nccl"])
def test_sharded_delta_cache_with_device_ckde_scores_world2(backend):
    """gloo: two ranks on ONE GPU (what a gpurun box has).  nccl: the same job over RCCL / xGMI with one device per rank - runs
    wherever two GPUs are visible (the driver's 8-GPU node), skipped otherwise: same assertions, the only code that differs is the
    staging of the gathered vector (distributed._all_gather)."""
    if backend == "nccl" and _device_count() < 2:
        pytest.skip("RCCL needs one device per rank: fewer than 2 GPUs visible")
    single = _launch(1)[0]
    ranks = _launch(2, backend)
    if backend == "nccl":
        assert all(r["ranks_seen"] == 2 and r["backend"] == "nccl" for r in ranks)
    assert len(single["trace"]) >= 3 and any(t[0] == 3 for t in single["trace"]) or len(single["arcs"]) >= 2
    for r in ranks:
        assert r["trace"] == single["trace"], (r["trace"], single["trace"])
        assert r["arcs"] == single["arcs"] and r["types"] == single["types"]
        assert r["cells"] == single["cells"]
        assert r["deltas"] == single["deltas"]   # bit for bit: terms summed in region order, joint sum minus marginal sum, on every rank
        assert abs(r["slogl"] - single["slogl"]) <= 1e-11 * abs(single["slogl"])
        # row-sharded moments (BGe / BIC): the summation does not depend on the number of ranks - identical deltas, bit for bit,
        # and therefore identical decisions at every score-equivalence tie
        for tag in ("bge", "bic"):
            assert len(single[tag + "_trace"]) >= 8
            assert r[tag + "_trace"] == single[tag + "_trace"]
            assert r[tag + "_deltas"] == single[tag + "_deltas"]
            assert r[tag + "_arcs"] == single[tag + "_arcs"]
        # hybrid candidates: their slices shared by the ranks, per-part sums added in part order - the one-process deltas, bit for bit
        assert len(single["hyb_trace"]) >= 3 and any(a in ("d1", "d2") for a, _ in single["hyb_arcs"])
        assert r["hyb_trace"] == single["hyb_trace"] and r["hyb_deltas"] == single["hyb_deltas"] and r["hyb_arcs"] == single["hyb_arcs"]
    assert max(r["hyb_sweeps"] for r in ranks) < single["hyb_sweeps"]
    # the device work was split, and no term was swept twice: together the ranks made exactly the single process's (term, fold) sweeps
    assert max(r["sweeps"] for r in ranks) < single["sweeps"]
    assert sum(r["sweeps"] for r in ranks) == single["sweeps"]


def test_rccl_world1_on_one_gpu():
    """The RCCL path EXECUTED on the one GPU a box has: a fresh child (tools/rccl_world1.py) initialises torch.distributed's "nccl"
    backend with one rank, forces the product's one-process-per-GPU path (PBN_FORCE_DIST / distributed.FORCE) and runs the searches of
    dist_worker_gpu.py through it - every delta-cache batch planned by csrc/shard.hip, gathered by RCCL on device buffers
    (distributed._all_gather's nccl branch), the moments row-sharded and reduced (pbn_scoredata_reduce_moments), the KDE slogl through
    pbn_kde_slogl_sharded - and compares every trace, delta and score with the plain calls, bit for bit.  Also records which
    libamdhip64 / librccl the process mapped (libpbn_hip.so is linked against /opt/rocm's runtime, torch brings its own)."""
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()), HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4", PBN_DEVICE="0")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "rccl_world1.py")], env=env, capture_output=True, text=True, timeout=600)
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
    assert p.returncode == 0 and lines, (p.stdout[-1500:], p.stderr[-3000:])
    out = json.loads(lines[-1][len("RESULT "):])
    assert out["ok"] and out["rccl_ranks_seen"] == 1 and out["all_gather_identity"]
    assert all(out["bit_identical"].values()), out["bit_identical"]
    assert out["collectives"] >= 10                       # one all-gather per delta-cache batch + the moments
    assert out["mapped"].get("librccl") and out["mapped"].get("libamdhip64") and out["mapped"].get("libpbn_hip")


def test_bench_two_ranks_on_one_gpu():
    """bench.py's own N > 1 path (launch_ranks -> torch.distributed.run -> init_process_group, the fixed-work cv64 leg sharded over the ranks,
    the weak-scaling leg, per-rank estimates) as the driver's SCALE run would start it - two gloo ranks on the ONE device a box has.  The launcher
    is a fresh child that spawns torch.distributed.run before any HIP call.  Shards learning/operators/operators.cpp:100-132,296-347,
    learning/scores/cv_likelihood.cpp:5-25."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    env.update(PBN_BENCH_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0", OMP_NUM_THREADS="4")
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--backend", "gloo", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"],
                       env=env, capture_output=True, text=True, timeout=900, cwd=ROOT)
    assert p.returncode == 0, p.stderr[-3000:]
    lines = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, p.stdout[-2000:]
    assert len(lines[0]) < 6000
    out = json.loads(lines[0])
    assert out["n_gpus"] == 2 and out["config"]["ranks"] == 2 and out["config"]["ranks_seen"] == 2
    assert out["config"]["backend"] == "gloo" and out["scaling"] == "weak"
    legs = out["legs"]
    strong = legs["cv64"]
    assert strong["ranks"] == 2 and strong["scaling"] == "strong" and strong["iterations"] == 5
    assert len(strong["per_rank_estimate_s"]) == 2 and all(x > 0 for x in strong["per_rank_estimate_s"])
    weak = legs["cv_weak"]
    assert weak["nodes"] == 32 and weak["ranks"] == 2 and weak["scaling"] == "weak" and len(weak["per_rank_estimate_s"]) == 2
    # the headline's exchange: the all_reduce of the per-step partial sums = the sum of the two ranks' own one-rank calls
    import torch

    sys.path.insert(0, ROOT)
    import bench
    import pybnesian_amd as pbn
    from pybnesian_amd import _lib

    ctx = pbn.Context(0)
    dev = torch.device("cuda", 0)
    names = [f"v{i}" for i in range(bench.D)]
    want = 0.0
    for rank in range(2):
        tr, te = bench.make_tables(torch, dev, 1_000_000, 100_000, 0, 1 + rank, torch.float64)
        torch.cuda.synchronize()
        a = pbn.DeviceTable.from_device_pointer(ctx, tr.data_ptr(), 1_000_000, names, 1_000_000, _lib.PBN_F64, keepalive=tr)
        b = pbn.DeviceTable.from_device_pointer(ctx, te.data_ptr(), 100_000, names, 100_000, _lib.PBN_F64, keepalive=te)
        k = pbn.ProductKDE(names)
        k.fit_table(a)
        buf = torch.zeros(1, dtype=torch.float64, device=dev)
        k.slogl_table_async(b, buf.data_ptr())
        ctx.sync()
        want += float(buf.item())
    assert abs(out["config"]["slogl_step0_rank_sum"] - want) <= 1e-11 * abs(want)
