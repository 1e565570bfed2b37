#!/usr/bin/env python3
"""bench.py — headline benchmark of the KDE hot path on MI355X.

Workload (BASELINE.json configs[1], "C2"): ProductKDE.slogl, fp64, N_train = 1e6, N_test = 1e5, d = 8,
diagonal ("product") bandwidth from the normal reference rule, synthetic correlated Gaussian table
(SURVEY.md §8d).  A *step* is one full slogl(test) on an already fitted model with train and test tables
resident in HBM: pack the queries -> fused pairwise/logsumexp sweep -> finish/reduce -> one scalar.

    python bench.py --gpus N --steps K --warmup W        # N > 1 without WORLD_SIZE: starts its own N ranks (see launch_ranks)
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 ... bench.py --gpus N ...

Multi-GPU (weak scaling): the fitted training set is replicated on every GPU, each rank owns its own
N_test test rows (independent units = test rows); there is no data-path collective — the
only exchange is one all-reduce of the K per-step partial sums at the end of the timed region.

Prints ONE JSON line (rank 0) with the contract fields plus `roofline` (dominant kernel = kde_sweep,
timed with HIP events on the library's own stream) and `cpu_baseline` (the oracle's restatement of the
reference algorithm on the host cores, bounded sample; rank 0, N=1 only).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

D = 8
FLOPS_PER_PAIR = 3 * D + 2          # d sub + d fma(2) + exp(1) + add(1)
FP64_PEAK_TFLOPS = 78.6             # MI355X FP64 vector == matrix peak (AMD CDNA4 spec; 256 CU x 4 SIMD x 32 flop/clk x 2.4 GHz)
FP32_PEAK_TFLOPS = 157.3            # MI355X FP32 vector == f32-matrix peak (MI355X_MICROARCH.md); the --dtype f32 run is priced against it
HBM_PEAK_GBS = 8000.0               # MI355X_MICROARCH.md: 8.0 TB/s spec
F32_VALU_CYCLES_PER_VALUE = 12.0    # fp32 sweep: v_exp_f32 (8 issue cycles per wave64) + v_add_f32 (4) per pair value


def make_tables(torch, device, n_train, n_test, seed_train, seed_test, dtype):
    """Column-major (d x n contiguous) correlated Gaussian tables, generated on the GPU."""
    mix = torch.tril(torch.full((D, D), 0.3, dtype=torch.float64, device=device), -1) + torch.eye(D, dtype=torch.float64, device=device)

    def gen(n, seed):
        g = torch.Generator(device=device)
        g.manual_seed(seed)
        z = torch.randn((D, n), generator=g, device=device, dtype=torch.float64)
        return (mix @ z).to(dtype).contiguous()  # row c = column c of the table

    return gen(n_train, seed_train), gen(n_test, seed_test)


def make_dag_table(torch, device, n_rows, n_cols, seed, dtype, nonlinear=False):
    """SURVEY.md §8d generator: random linear-Gaussian DAG (max indegree 3, coefficients U(-1.5, 1.5), noise
    sigma U(0.5, 1.5)), optionally tanh on half of the nodes.  Returns a (n_cols, n_rows) tensor (column-major table)."""
    rng = np.random.default_rng(seed)
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    out = torch.empty((n_cols, n_rows), dtype=torch.float64, device=device)
    for j in range(n_cols):
        k = int(rng.integers(0, min(3, j) + 1))
        parents = rng.choice(j, size=k, replace=False) if k else []
        sigma = float(rng.uniform(0.5, 1.5))
        col = torch.randn(n_rows, generator=g, device=device, dtype=torch.float64) * sigma
        for p in parents:
            col += float(rng.uniform(-1.5, 1.5)) * out[int(p)]
        if nonlinear and j % 2 == 1:
            col = torch.tanh(col) * 2.0 + 0.1 * col
        out[j] = col
    return out.to(dtype).contiguous()


def bench_hill_climb(torch, pbn, _lib, ctx, device, which, n_rows, max_iters):
    """Secondary metric: candidate-arcs (delta cells) scored per second during GreedyHillClimbing.estimate."""
    if which == "c4":
        n_cols = 64
        n_rows = n_rows or 2_000_000
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64)
        names = [f"x{i}" for i in range(n_cols)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
        t0 = time.perf_counter()
        score = pbn.BGe(None, table=table)
        t_ctor = time.perf_counter() - t0
        start, ops = pbn.GaussianNetwork(names), pbn.ArcOperatorSet()
        label = f"C4: 64-node GaussianNetwork, BGe, ArcOperatorSet, {n_rows} rows fp64"
        kw = {}
    elif which in ("c5", "c5mmhc"):
        # BASELINE config 5 (HC phase only; the MMPC restriction phase of MMHC is replaced by a supplied arc blacklist,
        # SURVEY.md §8d): 32 continuous + 16 dictionary columns, fp32, ValidatedLikelihood(0.2, 10, seed 0)
        import pandas as pd

        n_rows = n_rows or 1_000_000
        max_iters = max_iters or 1
        rng = np.random.default_rng(3)
        n_disc, n_cont = 16, 32
        cards = rng.integers(2, 5, size=n_disc)
        disc = {}
        for j in range(n_disc):
            base = rng.integers(0, cards[j], size=n_rows)
            if j > 0:  # dependence on the previous discrete column
                prev = disc[f"D{j - 1}"]
                flip = rng.random(n_rows) < 0.3
                base = np.where(flip, prev % cards[j], base)
            disc[f"D{j}"] = base.astype(np.int32)
        t = make_dag_table(torch, device, n_rows, n_cont, 3, torch.float32, nonlinear=True).cpu().numpy()
        cols = {}
        for j in range(n_cont):
            shift = 1.5 * disc[f"D{j % n_disc}"].astype(np.float32)  # means shifted per discrete parent
            cols[f"x{j}"] = t[j] + shift
        df = pd.DataFrame(cols)
        for j in range(n_disc):
            df[f"D{j}"] = pd.Categorical.from_codes(disc[f"D{j}"], [f"c{v}" for v in range(cards[j])])
        names = list(df.columns)
        t0 = time.perf_counter()
        score = pbn.ValidatedLikelihood(df, 0.2, 10, 0)
        t_ctor = time.perf_counter() - t0
        start = pbn.SemiparametricBN(names, [], [(f"D{j}", pbn.DiscreteFactorType()) for j in range(n_disc)])
        ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
        pairs = [(a, b) for a in names for b in names if a != b]
        extra = {}
        if which == "c5mmhc":
            # the real restriction phase: MMPC over all 48 variables with the hybrid MutualInformation test
            from pybnesian_amd.independences import mmpc_cpcs

            t0 = time.perf_counter()
            citest = pbn.MutualInformation(df)
            t_test = time.perf_counter() - t0
            t0 = time.perf_counter()
            cpcs, ntests = mmpc_cpcs(citest, names, 0.05)
            t_mmpc = time.perf_counter() - t0
            allowed = {n: set(c) for n, c in zip(names, cpcs)}
            blacklist = [(a, b) for a, b in pairs if b not in allowed[a]]
            extra = {"mmpc_s": t_mmpc, "ci_tests": ntests, "ci_tests_per_s": ntests / t_mmpc, "ci_test_ctor_s": t_test,
                     "cpc_edges": sum(len(c) for c in cpcs) // 2, "device_passes": citest.passes()[0], "host_passes": citest.passes()[1]}
            kw = {"max_indegree": 3, "arc_blacklist": blacklist}
            label = (f"C5 (MMHC): 48-node hybrid SemiparametricBN (16 discrete), MMPC with MutualInformation (alpha 0.05) then "
                     f"ValidatedLikelihood(0.2, 10) hill-climb, arcs+node_type, max_indegree=3, {n_rows} rows fp32")
        else:
            keep = rng.random(len(pairs)) < 0.15  # stand-in for the MMPC skeleton: 85 % of the arcs are blacklisted
            kw = {"max_indegree": 3, "arc_blacklist": [p for p, k_ in zip(pairs, keep) if not k_]}
            label = (f"C5 (HC phase): 48-node hybrid SemiparametricBN (16 discrete), ValidatedLikelihood(0.2, 10), arcs+node_type, "
                     f"85% arc blacklist, max_indegree=3, {n_rows} rows fp32")
    elif which == "cv64":
        # north star: "CV-likelihood hill-climbing on 64-node synthetic data" - CKDE candidates are sharded over the
        # ranks (fixed total work: strong scaling of the delta cache); bounded to cache_scores + max_iters iterations
        n_cols = 64
        n_rows = n_rows or 100_000
        max_iters = max_iters or 1
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64, nonlinear=True)
        names = [f"x{i}" for i in range(n_cols)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
        t0 = time.perf_counter()
        score = pbn.CVLikelihood(None, 10, 0, table=table)
        t_ctor = time.perf_counter() - t0
        start = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
        ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
        label = f"64-node SemiparametricBN (all CKDE start), 10-fold CVLikelihood, arcs+node_type, max_indegree=3, {n_rows} rows fp64"
        kw = {"max_indegree": 3}
    else:
        n_cols = 32
        n_rows = n_rows or 500_000
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64, nonlinear=True)
        names = [f"x{i}" for i in range(n_cols)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
        t0 = time.perf_counter()
        score = pbn.CVLikelihood(None, 10, 0, table=table)
        t_ctor = time.perf_counter() - t0
        start = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
        ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
        label = f"C3: 32-node SemiparametricBN (all CKDE start), 10-fold CVLikelihood, arcs+node_type, max_indegree=3, {n_rows} rows fp64"
        kw = {"max_indegree": 3}
    hc = pbn.GreedyHillClimbing()
    if max_iters:
        kw["max_iters"] = max_iters
    t0 = time.perf_counter()
    res = hc.estimate(ops, score, start, **kw)
    dt = time.perf_counter() - t0
    more = dict(extra) if which == "c5mmhc" else {}
    if which == "c4" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("PBN_BENCH_NO_CPU"):
        # the CPU side of BGe is the constructor (means + covariance of all columns, one thread in the
        # reference); timed with the restatement on a row sample and scaled linearly
        try:
            from oracle import oracle

            rows = min(n_rows, 200_000)
            host = t[:, :rows].T.cpu().numpy()
            t0 = time.perf_counter()
            oracle.cov(host)
            dcpu = (time.perf_counter() - t0) * (n_rows / rows)
            more["cpu_baseline"] = {"score_ctor_s": dcpu, "kind": "port", "cores": 1,
                                    "sample": f"covariance of {rows} x {n_cols} rows on one thread, scaled to {n_rows} rows; "
                                              f"per-candidate work is O(p^3) on both sides"}
        except Exception as ex:
            more["cpu_baseline"] = {"score_ctor_s": None, "sample": f"failed: {ex}"}
    return {
        **more,
        "metric": "hill-climb candidate-arcs scored/s",
        "value": hc.last.cells_scored / dt,
        "unit": "arcs/s",
        "config": label + (f", max_iters={max_iters}" if max_iters else ""),
        "cells_scored": hc.last.cells_scored,
        "local_score_evals": hc.last.local_score_evals,
        "iterations": hc.last.iterations,
        "arcs_found": res.num_arcs(),
        "estimate_s": dt,
        "score_ctor_s": t_ctor,
    }


def pmc_traffic(args):
    """HBM-side bytes per sweep launch from the committed rocprofv3 PMC passes (profiles/r1/pmc_per_dispatch.json:
    FETCH_SIZE and WRITE_SIZE in KB, collected in separate --pmc runs of this same command; FETCH_SIZE doubled as
    MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the default workload; null otherwise."""
    if (args.n_train, args.n_test, args.dtype, args.kde) != (1_000_000, 100_000, "f64", "product"):
        return None, None
    for rnd in ("r2", "r1"):
        path = os.path.join(ROOT, "profiles", rnd, "pmc_per_dispatch.json")
        try:
            with open(path) as f:
                d = json.load(f)
            k = next(v for name, v in d.items() if "kde_sweep_kernel<double, 2, false, 4" in name)
            return (2.0 * k["FETCH_SIZE"] * 1024.0 + k["WRITE_SIZE"] * 1024.0,
                    f"committed rocprofv3 --pmc passes of this command (profiles/{rnd}/pmc_per_dispatch.json: 2 x FETCH_SIZE + "
                    f"WRITE_SIZE), not measured in this run")
        except Exception:
            continue
    return None, None


def cpu_baseline(train_np, test_np, h, budget_s=12.0):
    """Oracle (port of the reference algorithm, kde/ProductKDE.hpp:240-293) on all host cores, bounded sample."""
    from oracle import oracle

    cores = oracle.num_threads()
    probe = max(cores, 8)
    t0 = time.perf_counter()
    oracle.product_kde_logl(train_np, h, test_np[:probe])
    dt = time.perf_counter() - t0
    rows = int(min(test_np.shape[0], max(probe, probe * budget_s / max(dt, 1e-3))))
    rows = max(cores, rows // cores * cores)
    t0 = time.perf_counter()
    oracle.product_kde_logl(train_np, h, test_np[:rows])
    dt = time.perf_counter() - t0
    out = {
        "value": rows / dt / 1e6,
        "unit": "M-samples/s",
        "cores": cores,
        "kind": "port",
        "sample": f"{rows} test rows x {train_np.shape[0]} training rows, d={D}, fp64, {dt:.1f}s wall, OpenMP over test rows",
    }
    # SURVEY.md §8d also asks for the single-thread port (the reference itself is single-threaded) and for scipy's
    # gaussian_kde, the oracle of the reference's own tests (full covariance, so not the same kernel as C2's diagonal one)
    try:
        one = max(8, min(64, rows // max(cores, 1)))
        oracle.set_num_threads(1)
        t0 = time.perf_counter()
        oracle.product_kde_logl(train_np, h, test_np[:one])
        d1 = time.perf_counter() - t0
        oracle.set_num_threads(cores)
        out["single_thread"] = {"value": one / d1 / 1e6, "unit": "M-samples/s", "sample": f"{one} test rows, {d1:.1f}s wall"}
        from scipy.stats import gaussian_kde

        sub = train_np[:: max(1, train_np.shape[0] // 100_000)]          # scipy needs N x m memory: 1e5 training rows
        k = gaussian_kde(sub.T)
        m = min(200, test_np.shape[0])
        t0 = time.perf_counter()
        k.logpdf(test_np[:m].T)
        d2 = time.perf_counter() - t0
        scale = train_np.shape[0] / sub.shape[0]
        out["scipy_gaussian_kde"] = {"value": m / (d2 * scale) / 1e6, "unit": "M-samples/s",
                                     "sample": f"full-covariance gaussian_kde.logpdf, {m} test rows x {sub.shape[0]} training rows in {d2:.1f}s, "
                                               f"scaled linearly to {train_np.shape[0]} training rows"}
    except Exception as ex:
        out["extra_error"] = f"{type(ex).__name__}: {ex}"
    return out


def launch_ranks(args):
    """`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment: start the N ranks ourselves (one
    process per GPU, `python -m torch.distributed.run` as a CHILD process) and exit with its code.  This parent never
    imports torch and never touches HIP (a process that has initialised the GPU must not be replaced, and the children
    own the devices); rank 0 of the children prints the JSON line on the inherited stdout."""
    import socket
    import subprocess

    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("MASTER_ADDR", "127.0.0.1")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def e2e_host(pbn, kde, names, test_np, repeats=3):
    """the same slogl with the test table in HOST Arrow memory - upload (PCIe), query pack, sweep, finish,
    scalar back - through the reference-shaped Python call `ProductKDE.slogl(record_batch)`.  Never the headline value."""
    import pyarrow as pa

    rb = pa.RecordBatch.from_arrays([pa.array(np.ascontiguousarray(test_np[:, i])) for i in range(test_np.shape[1])], names=names)
    kde.slogl(rb.slice(0, 256))   # warm the path
    best, val = None, None
    for _ in range(repeats):
        t0 = time.perf_counter()
        val = kde.slogl(rb)
        dt = time.perf_counter() - t0
        best = dt if best is None else min(best, dt)
    return {"value": rb.num_rows / best / 1e6, "unit": "M-samples/s", "ms": best * 1e3, "slogl": val,
            "what": f"{kde.__class__.__name__}.slogl(pyarrow.RecordBatch in host memory, {rb.num_rows} rows): H2D of the test "
                    f"columns + query pack + sweep + finish + D2H of the scalar, best of {repeats}"}


def dp_issue_util():
    """Share of the FP64 issue slots the sweep kept busy, from the committed rocprofv3 PMC pass of this command:
    (MFMA busy cycles + 4 cycles per VALU wave-instruction) / (cycles x 1024 SIMDs)."""
    for rnd in ("r2", "r1"):
        try:
            with open(os.path.join(ROOT, "profiles", rnd, "pmc_per_dispatch.json")) as f:
                d = json.load(f)
            k = next(v for name, v in d.items() if "kde_sweep_kernel<double, 2, false, 4" in name)
            simd_cycles = k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
            return {"value": (k["SQ_VALU_MFMA_BUSY_CYCLES"] + 4.0 * k["SQ_INSTS_VALU"]) / simd_cycles,
                    "mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
                    "valu_insts_per_pair_value": k["SQ_INSTS_VALU"] * 64.0 / 1e11,
                    "source": f"profiles/{rnd}/pmc_per_dispatch.json (separate rocprofv3 --pmc pass, not this run)"}
        except Exception:
            continue
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--n-train", type=int, default=1_000_000)
    ap.add_argument("--n-test", type=int, default=100_000)
    ap.add_argument("--dtype", default="f64", choices=["f64", "f32"])
    ap.add_argument("--kde", default="product", choices=["product", "full"])
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--hc", default="auto", choices=["auto", "none", "c4", "c3", "cv64", "c5", "c5mmhc"],
                    help="secondary hill-climb metric: c4 = BASELINE config 4 (BGe, replicated moments), c3 = config 3 at full "
                         "size (bounded by --hc-max-iters), cv64 = 64-node CV-likelihood CKDE hill-climb whose candidates are "
                         "sharded over the ranks; auto = c4 on one GPU, cv64 on several")
    ap.add_argument("--backend", default="nccl", choices=["nccl", "gloo"],
                    help="torch.distributed backend (nccl = RCCL over xGMI; gloo only to exercise the N>1 path on one GPU)")
    ap.add_argument("--no-e2e", action="store_true", help="skip the host-Arrow end-to-end leg (profiling runs: keeps every sweep launch the full one)")
    ap.add_argument("--no-c3", action="store_true", help="skip the bounded config-3 (CKDE, 10-fold CV) hill-climb leg of the default run")
    ap.add_argument("--hc-rows", type=int, default=0)
    ap.add_argument("--hc-max-iters", type=int, default=0)
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(launch_ranks(args))   # before any torch / HIP call in this process

    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    hc_auto = args.hc == "auto"
    if hc_auto:
        args.hc = "c4" if world == 1 else "cv64"
    if not torch.cuda.is_available():
        raise RuntimeError("bench.py needs an MI355X: no HIP device visible (there is no CPU fallback)")
    if "PBN_BENCH_DEVICE" in os.environ:  # testing aid: put every rank on one device (with --backend gloo)
        local_rank = int(os.environ["PBN_BENCH_DEVICE"])
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", device_id=device)
        else:
            dist.init_process_group("gloo")

    import pybnesian_amd as pbn
    from pybnesian_amd import _lib

    ctx = pbn.Context(local_rank)
    tdtype = torch.float64 if args.dtype == "f64" else torch.float32
    pdtype = _lib.PBN_F64 if args.dtype == "f64" else _lib.PBN_F32
    names = [f"v{i}" for i in range(D)]
    # same training set on every rank (seed 0); rank-specific test rows (seed 1 + rank)
    train_t, test_t = make_tables(torch, device, args.n_train, args.n_test, 0, 1 + rank, tdtype)
    torch.cuda.synchronize()
    train = pbn.DeviceTable.from_device_pointer(ctx, train_t.data_ptr(), args.n_train, names, args.n_train, pdtype, keepalive=train_t)
    test = pbn.DeviceTable.from_device_pointer(ctx, test_t.data_ptr(), args.n_test, names, args.n_test, pdtype, keepalive=test_t)

    kde = (pbn.ProductKDE if args.kde == "product" else pbn.KDE)(names)
    kde.fit_table(train)  # bandwidth (device Gram) + whitening/packing: not part of a slogl step
    partial = torch.zeros(args.steps + args.warmup, dtype=torch.float64, device=device)

    def step(i):
        kde.slogl_table_async(test, partial.data_ptr() + 8 * i)

    def sync_all():
        ctx.sync()
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    for i in range(args.warmup):
        step(i)
    sync_all()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(args.warmup + i)
    ctx.sync()
    if dist is not None:
        dist.all_reduce(partial)  # the only exchange: K scalars
    sync_all()
    elapsed = time.perf_counter() - t0
    sweep_ms, sweep_n = ctx.kernel_time(_lib.PBN_K_SWEEP)
    pack_ms, _ = ctx.kernel_time(_lib.PBN_K_PACK)
    fin_ms, _ = ctx.kernel_time(_lib.PBN_K_FINISH)
    ctx.set_profiling(False)
    if dist is not None:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device=device)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())

    hc_out = None
    if args.hc != "none":
        if args.no_cpu_baseline:
            os.environ["PBN_BENCH_NO_CPU"] = "1"
        try:
            hc_out = bench_hill_climb(torch, pbn, _lib, ctx, device, args.hc, args.hc_rows, args.hc_max_iters)
            if dist is not None:
                hc_out["ranks"] = world
                hc_out["sharding"] = ("CKDE candidates of every delta-cache batch dealt to the ranks by variable set, one all_gather of "
                                      "the batch's scores per batch; fixed total work (strong scaling of the search)")
        except Exception as ex:  # the secondary metric must never cost the headline line
            hc_out = {"metric": "hill-climb candidate-arcs scored/s", "value": None, "error": f"{type(ex).__name__}: {ex}"}
    c3_out = None
    if hc_auto and world == 1 and not args.no_c3:
        # the CKDE path of the search, driver-timed: BASELINE config 3 at FULL size (32-node SPBN, 10-fold CV, 500 k rows),
        # bounded to the initial delta cache + one iteration
        try:
            c3_out = bench_hill_climb(torch, pbn, _lib, ctx, device, "c3", 0, 1)
        except Exception as ex:
            c3_out = {"metric": "hill-climb candidate-arcs scored/s", "value": None, "error": f"{type(ex).__name__}: {ex}"}

    total_samples = args.n_test * world * args.steps
    value = total_samples / elapsed / 1e6
    slogl = float(partial[args.warmup].item())

    if rank == 0:
        es = 8 if args.dtype == "f64" else 4
        pairs = float(args.n_train) * float(args.n_test)
        sweep_s = sweep_ms / max(sweep_n, 1) * 1e-3
        alg_bytes = (args.n_train * D + args.n_test * D + args.n_test) * es
        achieved_tf = pairs * FLOPS_PER_PAIR / sweep_s / 1e12
        if args.dtype == "f64":
            peak_tf = FP64_PEAK_TFLOPS
            peak_note = ("against the FP64 vector==matrix peak (v_mfma_f64 and the FP64 VALU share their issue slots on MI355X: "
                         "DESIGN.md §3.1)")
        else:
            # fp32 sweep: the dot products run as bf16x3 on the bf16 matrix cores and overlap with the VALU, which is
            # the binding unit: per pair value one v_exp_f32 (quarter rate: 8 issue cycles per wave64 on this part,
            # profiles/r1/microbench_r1.txt) and one v_add_f32 (4) at least.  Peak pair rate by VALU issue =
            # SIMDs x clock x 64 / 12 cycles, expressed in the same algorithmic flops (3d+2 per pair)
            peak_pairs = 256 * 4 * 2.4e9 * 64.0 / F32_VALU_CYCLES_PER_VALUE
            peak_tf = peak_pairs * FLOPS_PER_PAIR / 1e12
            peak_note = (f"against the VALU-issue bound of the fp32 sweep: one v_exp_f32 (8 issue cycles per wave64) + one v_add_f32 (4) per "
                         f"pair value -> {peak_pairs / 1e12:.2f}e12 pairs/s x (3d+2) flop; the distances run as bf16x3 on the bf16 matrix "
                         f"cores under the VALU work (the FP32 vector peak, {FP32_PEAK_TFLOPS} TFLOP/s, is not the binding unit)")
        traffic, traffic_source = pmc_traffic(args)
        out = {
            "metric": "KDE slogl M-samples/s",
            "value": value,
            "unit": "M-samples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": args.dtype,
            "data": "synthetic",
            "config": {
                "workload": f"C2: {'ProductKDE' if args.kde == 'product' else 'KDE'}.slogl {args.dtype}, N_train={args.n_train}, "
                            f"N_test={args.n_test} per GPU, d={D}, normal-reference {'diagonal' if args.kde == 'product' else 'full'} bandwidth",
                "parallelism": f"test rows sharded over {world} GPU(s), training set replicated",
                "ranks": world,
                "backend": (args.backend + (" (RCCL)" if args.backend == "nccl" else "")) if world > 1 else None,
                "pairs_per_step_per_gpu": pairs,
                "slogl_step0_rank_sum": slogl,
            },
            "roofline": {
                "kernel": "kde_sweep_kernel" if args.dtype == "f64" else "kde_sweep_bf16_kernel",
                "bound": "mfma",
                "achieved": achieved_tf,
                "peak": peak_tf,
                "unit": "TFLOP/s",
                "frac": achieved_tf / peak_tf,
                "traffic": traffic,
                "traffic_source": traffic_source,
                "note": "compute-bound sweep (SURVEY.md §8d): algorithmic flops = (3d+2) per train/test pair (exp counted as 1) "
                        + peak_note + "; sweep launch duration from HIP events on the library stream",
                "avg_launch_ms": sweep_s * 1e3,
                "gpairs_per_s": pairs / sweep_s / 1e9,
                "hbm_algorithmic_bytes": alg_bytes,
                "hbm_achieved_GBs": alg_bytes / sweep_s / 1e9,
                "hbm_frac_of_8TBs": alg_bytes / sweep_s / 1e9 / HBM_PEAK_GBS,
                "pack_ms": pack_ms / max(sweep_n, 1),
                "finish_ms": fin_ms / max(sweep_n, 1),
            },
        }
        if args.dtype == "f64" and traffic is not None:
            out["roofline"]["dp_issue_util"] = dp_issue_util()
        if hc_out is not None:
            out["secondary"] = hc_out
        if c3_out is not None:
            out["secondary_c3"] = c3_out
        if world == 1 and not args.no_e2e:
            try:
                out["e2e_host"] = e2e_host(pbn, kde, names, test_t.T.cpu().numpy())
            except Exception as ex:
                out["e2e_host"] = {"value": None, "error": f"{type(ex).__name__}: {ex}"}
        if world == 1 and not args.no_cpu_baseline:
            h = np.asarray(kde.bandwidth, dtype=np.float64)
            sample_rows = min(args.n_test, 16384)
            train_np = train_t.T.cpu().numpy().astype(np.float64)
            test_np = test_t[:, :sample_rows].T.cpu().numpy().astype(np.float64)
            try:
                out["cpu_baseline"] = cpu_baseline(train_np, test_np, h)
            except Exception as ex:  # never lose the headline line to the baseline leg
                out["cpu_baseline"] = {"value": None, "unit": "M-samples/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
