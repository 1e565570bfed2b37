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
import re
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


class EvalLog:
    """Records every local-score evaluation the search asks the engine for (variable, node type, parents - as column names),
    by wrapping the score's batched entry point: the like-for-like work list of the CPU baseline."""

    def __init__(self, score):
        self.evals = []
        self.names = list(score._names)
        inner = score._batch_raw

        def wrapped(model, var, ntype, off, par, kind):
            for i in range(len(var)):
                self.evals.append((self.names[var[i]], int(ntype[i]), tuple(self.names[j] for j in par[off[i]: off[i + 1]]), int(kind)))
            return inner(model, var, ntype, off, par, kind)

        score._batch_raw = wrapped


def cpu_quota():
    """CPUs the container's cgroup grants (None = no quota): a box may show 256 CPUs under a 16-CPU quota, and more threads than that only thrash."""
    try:
        with open("/sys/fs/cgroup/cpu.max") as f:
            q_, per_ = f.read().split()[:2]
            return None if q_ == "max" else float(q_) / float(per_)
    except Exception:
        try:
            with open("/sys/fs/cgroup/cpu/cpu.cfs_quota_us") as f, open("/sys/fs/cgroup/cpu/cpu.cfs_period_us") as g:
                q_ = float(f.read())
                return None if q_ <= 0 else q_ / float(g.read())
        except Exception:
            return None


def host_cores(visible):
    quota = cpu_quota()
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = visible
    return int(max(1, min(visible, affinity, int(np.ceil(quota)) if quota else visible)))


def cpu_arcs_baseline(which, log, cells, host, budget_s=8.0):
    """CPU side of the candidate-arcs/s metric: the SAME local-score evaluations the device run
    made (EvalLog), evaluated by the oracle's restatement of the reference arithmetic on a bounded random sample of them and on
    a row subsample of the same table, extrapolated per class (node type, number of parents) with the cost law of the
    reference algorithm: CKDE likelihoods are O(train rows x test rows) pair sums -> x (rows / sample rows)^2, LinearGaussian
    fits and likelihoods O(rows) -> x (rows / sample rows).  `host(m)` returns (columns dict of numpy arrays over the first m
    rows, dict of discrete codes / cardinalities or None).  Returns arcs/s = cells / extrapolated seconds, single thread (the
    reference's hill-climb is single-threaded) and on all cores (the oracle's OpenMP loop over test rows)."""
    from oracle import baseline, oracle

    rng = np.random.default_rng(0)
    n_rows = which["rows"]
    classes = {}
    for ev in log.evals:
        ncont = sum(1 for p_ in ev[2] if not p_.startswith("D")) if which.get("hybrid") else len(ev[2])
        classes.setdefault((ev[1], ncont), []).append(ev)

    def make_score(m):
        cols, disc = host(m)
        if which["kind"] == "cv":
            def f(ev):
                data = np.column_stack([cols[ev[0]]] + [cols[p_] for p_ in ev[2]])
                return oracle.cv_likelihood(data, "ckde" if ev[1] == 1 else "lg", which["k"], which["seed"])
            return f
        codes, cards = disc
        ratio, k, seed = which["ratio"], which["k"], which["seed"]
        tr, te = oracle.holdout_split(m, ratio, seed)
        folds = oracle.cv_folds(tr.size, k, seed)

        def f(ev):
            var, par = ev[0], ev[2]
            dpar = [p_ for p_ in par if p_ in codes]
            cpar = [p_ for p_ in par if p_ not in codes]
            if var in codes:
                fn = lambda a, b: oracle.discrete_fit_slogl(codes[var], cards[var], [codes[q] for q in dpar], [cards[q] for q in dpar], a, b)
            else:
                cont = np.column_stack([cols[var]] + [cols[q] for q in cpar]).astype(np.float64)
                fn = lambda a, b: oracle.adaptator_fit_slogl(cont, [codes[q] for q in dpar], [cards[q] for q in dpar], a, b,
                                                             "ckde" if ev[1] == 1 else "lg")
            if ev[3] == 3:   # hold-out (validation) score
                return fn(tr, te)
            return sum(fn(tr[a], tr[b]) for a, b in folds)
        return f

    def measure(m, threads, budget):
        oracle.set_num_threads(threads)
        baseline.set_num_threads(threads)
        f = make_score(m)
        est, used, t_all = 0.0, 0, time.perf_counter()
        share = budget / max(1, len(classes))
        for (nt, ncont), evs in sorted(classes.items()):
            pick = [evs[i] for i in rng.permutation(len(evs))[:6]]
            t0, done = time.perf_counter(), 0
            for ev in pick:
                f(ev)
                done += 1
                if time.perf_counter() - t0 > share and done >= 2:
                    break
            per = (time.perf_counter() - t0) / done
            scale = (n_rows / m) ** 2 if nt == 1 else (n_rows / m)   # CKDE: pairs; LinearGaussian / discrete: rows
            est += per * scale * len(evs)
            used += done
        return est, used, time.perf_counter() - t_all

    visible = oracle.num_threads()
    cores = host_cores(visible)
    out = {"unit": "arcs/s", "kind": "port", "visible_cpus": visible,
           "law": "CKDE evaluations scaled by (rows / sample rows)^2, LinearGaussian / discrete ones by rows / sample rows; per class "
                  "(node type, continuous parents): mean oracle time of up to 6 sampled evaluations x the class's evaluations",
           "ckde": "the CKDE log-likelihoods inside the sampled evaluations come from the tuned port (oracle/pbn_baseline.cpp: whitened, blocked, "
                   "vectorised exponentials), the fits / folds / LinearGaussian parts from the reference-arithmetic restatement; threads = the "
                   "container's CPU quota"}
    oracle.use_tuned_ckde(True)
    try:
        m_all = int(min(n_rows, which.get("sample_rows_all", 50000)))
        est, used, wall = measure(m_all, cores, budget_s * 0.5)
        out.update({"value": cells / est, "cores": cores, "cpu_seconds_extrapolated": est, "extrapolated": True,
                    "sample": f"{used} of the run's {len(log.evals)} local-score evaluations on the first {m_all} of {n_rows} rows, {wall:.1f}s wall"})
        m_one = int(min(n_rows, which.get("sample_rows_one", 10000)))
        est1, used1, wall1 = measure(m_one, 1, budget_s * 0.5)
        out["single_thread"] = {"value": cells / est1, "cores": 1, "cpu_seconds_extrapolated": est1,
                                "sample": f"{used1} evaluations on the first {m_one} rows, {wall1:.1f}s wall"}
    except Exception as ex:
        out["error"] = f"{type(ex).__name__}: {ex}"
    finally:
        oracle.use_tuned_ckde(False)
        oracle.set_num_threads(visible)
        baseline.set_num_threads(visible)
    return out


def eight_rank_estimate(run, one_rank_leg):
    """No 8-GPU node at hand: ONE process plays the eight ranks of a hill-climb in turn through the library's own sharding
    (pbn_scoredata_set_comm) and times every rank's share of every batch (tools/shard_emulate.py).  An 8-rank job waits per batch for
    its slowest share and repeats the unsharded work: T_8 ~ (time - sum of the shares) + sum over batches of the slowest share.  An
    ESTIMATE - no collective latency, one process's caches, the score engine's phase only - never a leg's `value`."""
    try:
        sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))
        from shard_emulate import EmulatedRanks

        with EmulatedRanks(8) as em:
            r8 = run()
        one = r8["estimate_s"] + r8.get("score_ctor_s", 0.0)
        est = em.estimate(one)
        t8 = est["per_rank_s"]
        t1 = one_rank_leg["estimate_s"] + one_rank_leg.get("score_ctor_s", 0.0)
        return {"kind": "one-process emulation, not a multi-GPU measurement", "cells_scored": r8["cells_scored"], "batches": len(em.batches),
                "one_process_s": one, "per_rank_s": t8, "slowest_over_mean_share": est["slowest_over_mean_share"],
                "arcs_per_s": r8["cells_scored"] / t8, "ratio_to_one_rank_leg": (r8["cells_scored"] / t8) / (one_rank_leg["cells_scored"] / t1),
                "method": "per batch the slowest of the eight shares (CKDE terms dealt by cost, the slices of hybrid candidates shared: the plan of "
                          "csrc/shard.hip, every rank played in turn by one process), plus the unsharded time; collective latency and per-rank "
                          "caches not modelled"}
    except Exception as ex:   # an estimate must never cost the line
        return {"error": f"{type(ex).__name__}: {ex}"}


TIE_TOL = 1e-9


def tie_accounting(pbn, hc, names, score_fn, **kw):
    """Driver-visible form of tests/test_tieflip_gpu.py (north star: "bit-exact node/arc indices"): the product's operator sequence
    is REPLAYED inside the serial restatement (oracle/hc_oracle.py, `follow=`), which takes its own greedy decision from its own
    (reference-arithmetic) deltas at every step.  Every step where the two differ is classified: a TIE of the restatement's own
    deltas (gap <= 1e-9 relative: score-equivalent orientations decided in the last ulps by find_max's unstable sort,
    operators.hpp:489-525) or a real divergence.  `non_tie_divergences` must be 0; `end_gain` is what the restatement could still
    gain where the product stopped (0 = the product's graph is a local optimum of the reference's score)."""
    from oracle import hc_oracle

    idx = {c: i for i, c in enumerate(names)}
    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    trace = [(kinds[type(op)], idx[op.source()], idx[op.target()]) for op in hc.last.trace]
    deltas = [op.delta() for op in hc.last.trace]
    t0 = time.perf_counter()
    _arcs, _types, r_trace, info = hc_oracle.estimate(len(names), 0, score_fn, follow=trace, tie_tol=float("inf"), **kw)
    flips = info["flips"]
    ties = [f for f in flips if f["gap"] <= TIE_TOL * max(1.0, abs(f["followed_delta"]))]
    end_gain = info.get("end_gain")
    return {"replayed_iterations": len(trace), "tie_flips": len(ties), "non_tie_divergences": len(flips) - len(ties),
            "max_gap": max((f["gap"] for f in flips), default=0.0),
            "max_rel_gap": max((f["gap"] / max(1.0, abs(f["followed_delta"])) for f in flips), default=0.0),
            "flip_iterations": [f["iteration"] for f in flips][:32],
            "max_delta_rel_diff": float(max((abs(a - b[3]) / max(1.0, abs(b[3])) for a, b in zip(deltas, r_trace)), default=0.0)),
            "end_gain": end_gain, "product_graph_is_oracle_local_optimum": end_gain is not None and end_gain <= TIE_TOL,
            "tie_tol_rel": TIE_TOL, "replay_s": time.perf_counter() - t0,
            "how": "the product's trace replayed in oracle/hc_oracle.estimate(follow=...) over the oracle's own scores on the same rows"}


def bench_c1(pbn):
    """BASELINE config 1 (the reference's CPU-runnable case): GaussianNetwork, 4 nodes, LinearGaussianCPD MLE fit + BIC on a
    10 k-row table - BIC hill-climb + fit + slogl through the device engine, and the SAME search by the serial restatement
    (oracle/hc_oracle.py over oracle.bic_lg) on the host: both complete, nothing sampled."""
    import pandas as pd

    rng = np.random.default_rng(0)
    n = 10_000
    a = rng.normal(size=n)
    b = 0.7 * a + rng.normal(scale=0.8, size=n)
    c = -0.5 * a + 1.2 * b + rng.normal(scale=0.6, size=n)
    d = 0.9 * c + rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c, "d": d})
    names = list(df.columns)

    def run():
        t0 = time.perf_counter()
        score = pbn.BIC(df)
        hc = pbn.GreedyHillClimbing()
        res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names))
        res.fit(df)
        sl = res.slogl(df)
        return time.perf_counter() - t0, hc, res, sl

    cold, _, _, _ = run()                 # first use of these kernel shapes in the process (code objects, arenas): reported, not the value
    dt, hc, res, sl = min((run() for _ in range(3)), key=lambda r: r[0])
    out = {"metric": "hill-climb candidate-arcs scored/s", "unit": "arcs/s", "value": hc.last.cells_scored / dt,
           "config": "C1: GaussianNetwork 4 nodes, BIC hill-climb + LinearGaussianCPD MLE fit + slogl, 10000 rows fp64 (host pandas table in, "
                     "upload + score construction + search + fit + slogl timed; best of 3 warm runs, the cold first run in `cold_seconds`)",
           "cold_seconds": cold,
           "cells_scored": hc.last.cells_scored, "iterations": hc.last.iterations, "arcs_found": res.num_arcs(), "seconds": dt, "slogl": sl}
    try:
        from oracle import hc_oracle, oracle

        data = df.to_numpy()
        cores = oracle.num_threads()
        oracle.set_num_threads(1)
        t0 = time.perf_counter()
        calls = [0]

        def sc(v, _nt, par):
            calls[0] += 1
            return oracle.bic_lg(data[:, [v] + list(par)])

        arcs, _types, trace = hc_oracle.estimate(4, 0, sc)[:3]
        for v in range(4):   # MLE fit + log-likelihood of every node, as fit() + slogl() do
            par = [s_ for s_, t_ in arcs if t_ == v]
            beta, var = oracle.lg_fit(data[:, [v] + par])
            oracle.lg_logl(data[:, [v] + par], beta, var)
        dcpu = time.perf_counter() - t0
        oracle.set_num_threads(cores)
        mine = [(names.index(s_), names.index(t_)) for s_, t_ in res.arcs()]
        out["cpu_baseline"] = {"value": hc.last.cells_scored / dcpu, "unit": "arcs/s", "cores": 1, "kind": "port", "seconds": dcpu,
                               "same_structure": sorted(arcs) == sorted(mine),
                               "same_skeleton": sorted(tuple(sorted(a_)) for a_ in arcs) == sorted(tuple(sorted(a_)) for a_ in mine),
                               "sample": f"the whole search, fit and log-likelihood by the serial restatement: {calls[0]} oracle BIC calls on all 10000 rows"}
        # 10 000 rows: milliseconds on either side - the device buys nothing at this size (a single CPU thread is within 1.5x of it)
        out["device_over_cpu"] = dcpu / out["seconds"]
        out["note"] = "C1 is the reference's own CPU-runnable case: at 10 000 rows the device is launch-bound and buys nothing over one CPU thread"
        out["tie_accounting"] = tie_accounting(pbn, hc, names, sc)
    except Exception as ex:
        out["cpu_baseline"] = {"value": None, "error": f"{type(ex).__name__}: {ex}"}
    return out


def bench_hill_climb(torch, pbn, _lib, ctx, device, which, n_rows, max_iters, n_cols=0, cpu=True):
    """Secondary metric: candidate-arcs (delta cells) scored per second during GreedyHillClimbing.estimate."""
    host = None
    cpu_cfg = None
    if which == "c4":
        n_cols = 64
        n_rows = n_rows or 2_000_000
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64)
        names = [f"x{i}" for i in range(n_cols)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
        ctx.set_profiling(2)
        t0 = time.perf_counter()
        score = pbn.BGe(None, table=table)
        t_ctor = time.perf_counter() - t0
        gram_first_ms, gram_n = ctx.kernel_time(_lib.PBN_K_GRAM)
        # the same construction once more: the first launch of a kernel in a process also pays its code-object load and finds the clocks
        # where the previous leg left them - the roofline below prices the second, `first_launch_us` keeps the first
        ctx.set_profiling(2)
        score2 = pbn.BGe(None, table=table)
        gram_ms, gram_n = ctx.kernel_time(_lib.PBN_K_GRAM)
        ctx.set_profiling(False)
        del score2
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
        cpu_cfg = {"kind": "validated", "ratio": 0.2, "k": 10, "seed": 0, "rows": n_rows, "hybrid": True, "sample_rows_all": 60000, "sample_rows_one": 16000}

        def host(m, cols=cols, disc=disc, cards=cards):
            return ({k_: np.asarray(v_[:m], dtype=np.float64) for k_, v_ in cols.items()},
                    ({k_: v_[:m] for k_, v_ in disc.items()}, {f"D{j}": int(cards[j]) for j in range(len(cards))}))
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
        n_cols = n_cols or 64
        n_rows = n_rows or 100_000
        max_iters = max_iters or 1
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64, nonlinear=True)
        cpu_cfg = {"kind": "cv", "k": 10, "seed": 0, "rows": n_rows}
        names = [f"x{i}" for i in range(n_cols)]
        torch.cuda.synchronize()
        table = pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), n_rows, names, n_rows, _lib.PBN_F64, keepalive=t)
        t0 = time.perf_counter()
        score = pbn.CVLikelihood(None, 10, 0, table=table)
        t_ctor = time.perf_counter() - t0
        start = pbn.SemiparametricBN(names, [], [(n, pbn.CKDEType()) for n in names])
        ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
        label = f"{n_cols}-node SemiparametricBN (all CKDE start), 10-fold CVLikelihood, arcs+node_type, max_indegree=3, {n_rows} rows fp64"
        kw = {"max_indegree": 3}
    else:
        n_cols = 32
        n_rows = n_rows or 500_000
        t = make_dag_table(torch, device, n_rows, n_cols, 2, torch.float64, nonlinear=True)
        cpu_cfg = {"kind": "cv", "k": 10, "seed": 0, "rows": n_rows}
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
    want_cpu = cpu and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("PBN_BENCH_NO_CPU")
    log = EvalLog(score) if (want_cpu and cpu_cfg is not None) else None
    sctx = getattr(score, "_ctx", ctx)   # the context the score's launches go to (a score built from a host frame takes the default one)
    sctx.set_profiling(2)   # HIP events around the library's sweep / Gram launches, nothing else changes (pbn_ctx_set_profiling: 2 = timing only)
    import ctypes as C
    lib = _lib.load()
    lib.pbn_debug_moment_totals(None, None, 1)   # (tile, group) pairs the tile-moment pass takes: one atomic per wave, always on
    t0 = time.perf_counter()
    res = hc.estimate(ops, score, start, **kw)
    dt = time.perf_counter() - t0
    kt = {name: sctx.kernel_time(cls) for name, cls in (("sweep", _lib.PBN_K_SWEEP), ("gram", _lib.PBN_K_GRAM), ("moment", _lib.PBN_K_MOMENT))}
    mp1, mp2 = C.c_ulonglong(0), C.c_ulonglong(0)
    lib.pbn_debug_moment_totals(C.byref(mp1), C.byref(mp2), 0)
    sctx.set_profiling(False)
    more = dict(extra) if which == "c5mmhc" else {}
    if which != "c4":
        # the leg's dominant kernel and its share of the timed search, from the library's own HIP events (no profiler)
        kname = {"c3": "kde_sweep_group_kernel<double, 1, 2, FOLD> (grouped pruned fp64 sweep, sum-only) + kde_moment_group_kernel<1 | 2> "
                       "(the tile-moment pass of its one- and two-variable terms: one event pair spans both launches)",
                 "cv64": "kde_sweep_group_kernel<double, 1, 2, FOLD> (grouped pruned fp64 sweep, sum-only)"}.get(
                     which, "kde_sweep_f16_group_kernel<1> (grouped pruned fp32 sweep on f16x2 fragments) + the per-slice kde_sweep_f16_kernel")
        kshort = {"c3": "kde_moment_group_kernel<2> + kde_sweep_group_kernel<double>", "cv64": "kde_sweep_group_kernel<double>"}.get(which, "kde_sweep_f16_group_kernel")
        more["roofline"] = {"kernel": kname, "kernel_short": kshort, "bound": "valu-issue (v_exp + add per pair value inside the pruning radius; DESIGN.md 3.1)",
                            "device_s": kt["sweep"][0] * 1e-3, "launches": kt["sweep"][1], "share_of_estimate_s": kt["sweep"][0] * 1e-3 / dt,
                            "gram_s": kt["gram"][0] * 1e-3, "gram_launches": kt["gram"][1],
                            **moment_roofline(kt["moment"][0] * 1e-3, kt["moment"][1], mp1.value, mp2.value),
                            "note": "HIP events on the launching stream around the sweep and Gram launches only (issue lanes overlap: device "
                                    "seconds may add up to more than the wall time)"}
    # The CPU side of the leg (baseline on the leg's own work list, tie accounting): host work of seconds on all cores and gigabytes of host
    # copies.  Inside the default command it runs AFTER every leg has been timed (DEFERRED_CPU), so that none of it stands between two timed
    # searches.  (Measured: the legs' host time - estimate_s minus device_s - still differs by box and process state, C5 0.9-1.4 s and
    # cv_weak 0.05-0.2 s, with or without this deferral; the device seconds are the stable part.)
    def cpu_part(more):
        nonlocal host
        if log is not None:
            if host is None:   # device-generated table (n_cols x n_rows): the first m rows, column by column
                def host(m, t=t, names=names):
                    a = t[:, :m].cpu().numpy().astype(np.float64)
                    return {nm: a[i] for i, nm in enumerate(names)}, None
            more["cpu_baseline"] = cpu_arcs_baseline(cpu_cfg, log, hc.last.cells_scored, host)
        if which == "c4" and int(os.environ.get("WORLD_SIZE", "1")) == 1 and not os.environ.get("PBN_BENCH_NO_CPU"):
            # the CPU side of BGe is the constructor (means + covariance of all columns, one thread in the
            # reference); timed with the restatement on a row sample and scaled linearly
            try:
                from oracle import oracle

                rows = min(n_rows, 200_000)
                host = t[:, :rows].T.cpu().numpy()
                cores = oracle.num_threads()
                oracle.set_num_threads(1)
                t0 = time.perf_counter()
                oracle.cov(host)
                dcpu = (time.perf_counter() - t0) * (n_rows / rows)
                oracle.set_num_threads(cores)
                more["cpu_baseline"] = {"value": hc.last.cells_scored / dcpu, "unit": "arcs/s", "score_ctor_s": dcpu, "kind": "port", "cores": 1,
                                        "sample": f"covariance of {rows} x {n_cols} rows on one thread (the reference's BGe constructor, bge.hpp:52-72), "
                                                  f"scaled to {n_rows} rows; the per-candidate work is O(p^3) on the cached covariance on both sides and is "
                                                  f"left out of the CPU time (an upper bound of the CPU rate); compare with value_with_ctor"}
            except Exception as ex:
                more["cpu_baseline"] = {"score_ctor_s": None, "sample": f"failed: {ex}"}
            try:
                # tie accounting on ALL rows: the oracle's whole-table moments once (two-pass, all cores - the sums do not depend on the
                # thread count), then the reference's BGe arithmetic per local score from them (bge.hpp:52-68 caches exactly these)
                from oracle import oracle

                t0 = time.perf_counter()
                full = np.asfortranarray(t.T.cpu().numpy())
                cov_all, means_all = oracle.cov(full)
                del full
                t_cov = time.perf_counter() - t0
                memo = {}

                def sc(v, _nt, par):
                    key = (v, tuple(par))
                    if key not in memo:
                        memo[key] = oracle.bge_cached(cov_all, means_all, n_rows, [v] + list(par), n_cols)
                    return memo[key]

                more["tie_accounting"] = dict(tie_accounting(pbn, hc, names, sc), rows=n_rows, oracle_moments_s=t_cov)
            except Exception as ex:
                more["tie_accounting"] = {"error": f"{type(ex).__name__}: {ex}"}
    if which == "c4":
        # north star: "MFMA utilisation for the covariance batch stated against gfx950 peak".  The BGe constructor's moments are ONE
        # segmented Gram launch over the whole table (scoring.hip: compute_stats_segments -> gram_glds_kernel + the segment reduce):
        # algorithmic bytes = rows x columns x 8 (every value read once), duration = HIP events of THIS run's constructor
        gbytes = float(n_rows) * n_cols * 8.0
        gsec = gram_ms * 1e-3 / max(gram_n, 1)
        rec, why = pmc_record("gram_glds_kernel", "gram_pmc.json", "stats_kernels.hip")
        roof = {"kernel": "gram_glds_kernel<4> (segmented form) + segment reduce", "bound": "hbm", "achieved": gbytes / gsec / 1e9, "peak": HBM_PEAK_GBS,
                "unit": "GB/s", "frac": gbytes / gsec / 1e9 / HBM_PEAK_GBS, "frac_of_measured_copy": gbytes / gsec / 1e9 / 6290.0,
                "algorithmic_bytes": gbytes, "launch_us": gsec * 1e6, "launches": gram_n, "first_launch_us": gram_first_ms * 1e3 / max(gram_n, 1),
                "note": "HIP events around the score constructor's Gram launch pair in this run (the second construction; first_launch_us = the "
                        "first, which `score_ctor_s` times); 6.29 TB/s = the best device-to-device copy measured on "
                        "this part (profiles/r2/gram_floors.txt); mfma_tflops = EXECUTED flops (upper-triangle tile pairs) on v_mfma_f64_16x16x4_f64"}
        # EXECUTED flops: the kernel computes the upper triangle of 16 x 16 column-tile pairs (stats_kernels.hip: 10 of 16 at 64 columns), one
        # v_mfma_f64_16x16x4_f64 (2 048 flop) per tile pair and 4 rows - the count SQ_INSTS_MFMA reports; the full n x n product the
        # triangle stands for is kept under its own name
        nct = (n_cols + 15) // 16
        roof["mfma_tflops"] = float(n_rows) / 4.0 * (nct * (nct + 1) // 2) * 2048.0 / gsec / 1e12
        roof["mfma_frac_of_fp64_peak"] = roof["mfma_tflops"] / FP64_PEAK_TFLOPS
        roof["algorithmic_full_matrix_tflops"] = float(n_rows) * n_cols * n_cols * 2.0 / gsec / 1e12
        roof["kernel_short"] = "gram_glds_kernel<4>"
        if rec is not None:
            cyc = rec["GRBM_GUI_ACTIVE"] / 8.0
            roof["mfma_busy"] = rec["SQ_VALU_MFMA_BUSY_CYCLES"] / 1024.0 / cyc
            roof["traffic"] = 2.0 * rec["FETCH_SIZE"] * 1024.0 if "FETCH_SIZE" in rec else None
            roof["pmc_source"] = (f"profiles/{rec['_round']}/gram_pmc.json (separate rocprofv3 --pmc passes of tools/gram_bench.py on stats_kernels.hip blob "
                                  f"{rec['_blob'][:12]} = the working tree's; MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / 1024 SIMDs / GPU cycles; traffic = 2 x FETCH_SIZE)")
        else:
            roof["mfma_busy"], roof["traffic"], roof["pmc_source"] = None, None, why
        more["roofline"] = roof
    out = {
        **more,
        "metric": "hill-climb candidate-arcs scored/s",
        "which": {"c5mmhc": "c5"}.get(which, which),
        "value": hc.last.cells_scored / dt,
        "unit": "arcs/s",
        "config": label + (f", max_iters={max_iters}" if max_iters else ""),
        "cells_scored": hc.last.cells_scored,
        "local_score_evals": hc.last.local_score_evals,
        "iterations": hc.last.iterations,
        "near_tie_redos": getattr(hc.last, "near_tie_redos", None),
        "arcs_found": res.num_arcs(),
        "estimate_s": dt,
        "score_ctor_s": t_ctor,
        "value_with_ctor": hc.last.cells_scored / (dt + t_ctor),
    }
    if DEFERRED_CPU is not None:
        DEFERRED_CPU.append(lambda: cpu_part(out))
    else:
        cpu_part(out)
    return out


DEFERRED_CPU = None   # a list while the default command collects its legs: the CPU parts of the hill-climb legs, run after the last leg


def moment_roofline(moment_s, launches, pairs_d1, pairs_d2):
    """The tile-moment pass (kde_moment_group_kernel<1 | 2>) against the FP64 vector peak: a (tile, group) pair is 16 queries x
    the Horner scheme of the order-8 polynomial - 8 FMAs in one dimension, 44 in two - so its algorithmic flops are pairs x 16 x FMAs x 2; the
    device seconds are the library's own HIP events around the pass (they lie INSIDE the sweep class's bracket: `device_s` includes them)."""
    if not launches or moment_s <= 0:
        return {}
    flops = (pairs_d1 * 8.0 + pairs_d2 * 44.0) * 16.0 * 2.0
    return {"moment_s": moment_s, "moment_launches": launches, "moment_pairs": pairs_d1 + pairs_d2, "moment_pairs_d2": pairs_d2,
            "moment_frac": flops / moment_s / (FP64_PEAK_TFLOPS * 1e12),
            "moment_cycles_per_pair": moment_s * 2.4e9 * 1024.0 / max(pairs_d1 + pairs_d2, 1),
            "moment_note": "FMAs of the Horner schemes alone against the FP64 vector peak (the kernel issues ~65 instructions per 44 FMAs at two "
                           "dimensions: its own instruction-count bound is 0.68 of this peak); cycles per (tile, group) pair at 2.4 GHz on 1 024 SIMDs"}


def git_blob_sha1(path):
    """`git hash-object` of a working-tree file (sha1 of "blob <size>\\0" + content), without calling git."""
    import hashlib

    with open(path, "rb") as f:
        data = f.read()
    return hashlib.sha1(b"blob %d\0" % len(data) + data).hexdigest()


def pmc_record(kernel_substr, fname="pmc_per_dispatch.json", src_name="kde_kernels.hip"):
    """The committed rocprofv3 PMC record of one kernel (profiles/r*/<fname>, newest round first) - but only if it was taken on THIS
    revision of the kernel source: the file stores the git blob hash of pybnesian_amd/csrc/<src_name> it was collected on
    (`_source_blob`), and a record whose hash differs from the working tree's is stale evidence -> (None, reason)."""
    src = os.path.join(ROOT, "pybnesian_amd", "csrc", src_name)
    try:
        now = git_blob_sha1(src)
    except OSError as ex:
        return None, f"kernel source not readable: {ex}"
    reason = f"no profiles/r*/{fname}"
    rounds = sorted((r for r in os.listdir(os.path.join(ROOT, "profiles")) if re.fullmatch(r"r\d+", r)), key=lambda r: -int(r[1:]))
    for rnd in rounds:
        path = os.path.join(ROOT, "profiles", rnd, fname)
        try:
            with open(path) as f:
                d = json.load(f)
        except Exception:
            continue
        blob = d.get("_source_blob", {}).get(src_name)
        if blob != now:
            reason = (f"stale: profiles/{rnd}/{fname} was collected on {src_name} blob {str(blob)[:12]}, the working tree "
                      f"has {now[:12]} - re-run the profiling script (tools/profile_bench.sh / tools/gram_pmc.sh)")
            return None, reason
        try:
            k = next(v for name, v in d.items() if kernel_substr in name)
        except StopIteration:
            return None, f"profiles/{rnd}/{fname} has no record of {kernel_substr}"
        return dict(k, _round=rnd, _blob=now), None
    return None, reason


SWEEP_KERNEL = "kde_sweep_kernel<double, 2, false, 4"


def pmc_traffic(args):
    """HBM-side bytes per sweep launch from the committed rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE in KB, collected in
    separate --pmc runs of this same command; FETCH_SIZE doubled as MI355X_MICROARCH.md prescribes for gfx950).  Only valid for the
    default workload and for the kernel source the passes were taken on (pmc_record); null with the reason otherwise."""
    if (args.n_train, args.n_test, args.dtype, args.kde) != (1_000_000, 100_000, "f64", "product"):
        return None, "not the default workload"
    k, why = pmc_record(SWEEP_KERNEL)
    if k is None:
        return None, why
    return (2.0 * k["FETCH_SIZE"] * 1024.0 + k["WRITE_SIZE"] * 1024.0,
            f"committed rocprofv3 --pmc passes of this command (profiles/{k['_round']}/pmc_per_dispatch.json: 2 x FETCH_SIZE + "
            f"WRITE_SIZE; collected on kde_kernels.hip blob {k['_blob'][:12]} = the working tree's), not measured in this run")


def parity_rows(torch, train_t, test_t, h_diag, n_first=1024, n_far=8, n_near=8):
    """Rows of the test table the oracle is run on: the first `n_first` (random rows of the table) plus, for each of the `n_far`
    training rows farthest from the centre in bandwidth units (the rows whose norms take the sweep's rare paths: WMUL weights that
    underflow, redone chunks - DESIGN.md 3.1), the `n_near` test rows nearest to it."""
    sd = torch.sqrt(torch.as_tensor(h_diag, dtype=torch.float64, device=train_t.device))[:, None]
    zt = (train_t.double() - train_t.double().mean(dim=1, keepdim=True)) / sd
    far = torch.topk((zt * zt).sum(dim=0), n_far).indices
    zq = (test_t.double() - train_t.double().mean(dim=1, keepdim=True)) / sd
    rows = [torch.arange(min(n_first, test_t.shape[1]), device=train_t.device)]
    for i in far.tolist():
        d2 = ((zq - zt[:, i:i + 1]) ** 2).sum(dim=0)
        rows.append(torch.topk(d2, n_near, largest=False).indices)
    rows = torch.unique(torch.cat(rows))
    return rows.cpu().numpy(), float((zt * zt).sum(dim=0).max().item())


def parity_block(torch, kde, test_table, train_t, test_t, slogl_step, tol, abs_tol=None):
    """Oracle numbers against the BASELINE-size run itself: per-row logl of the WHOLE test table through the same device sweep the
    timed steps ran (its sum must reproduce the timed step's slogl), and the oracle's logl (pbn_oracle.cpp: the reference's
    arithmetic, fp64, all N_train rows) on `parity_rows`.  `ok` = relative slogl difference over those rows <= tol and every
    per-row logl within tol relative (fp32 tables: within abs_tol absolute, the reference tests' own fp32 tolerance)."""
    from oracle import oracle

    t0 = time.perf_counter()
    got_all = kde.logl_table(test_table)
    bw = np.asarray(kde.bandwidth, dtype=np.float64)
    h_diag = bw if bw.ndim == 1 else np.diag(bw)
    rows, max_z2 = parity_rows(torch, train_t, test_t, h_diag)
    train_np = train_t.T.cpu().numpy().astype(np.float64)
    test_np = test_t.T.cpu().numpy().astype(np.float64)[rows]
    fn = oracle.product_kde_logl if bw.ndim == 1 else oracle.kde_logl
    want = fn(train_np, bw, test_np)
    got = got_all[rows]
    rel = np.abs(got - want) / np.maximum(1.0, np.abs(want))
    # the north star's quantity - slogl - through the kernel the timed steps ran (sum-only evaluations take 2^f on the fp32
    # transcendental unit, per-row logl outputs the fp64 polynomial: DESIGN.md 3.1): the same rows as one slogl call
    sub = test_table.take(rows.astype(np.int32))
    slogl_rows = kde.slogl_table(sub)
    rel_slogl = abs(slogl_rows - want.sum()) / abs(want.sum())
    rel_slogl_from_logl = abs(got.sum() - want.sum()) / abs(want.sum())
    sum_rel = abs(got_all.sum() - slogl_step) / abs(slogl_step)
    ok = bool(rel_slogl <= tol and rel_slogl_from_logl <= tol and sum_rel <= tol and np.isfinite(got_all).all()
              and (np.abs(got - want).max() <= abs_tol if abs_tol is not None else rel.max() <= tol))
    return {"ok": ok, "rows": int(rows.size), "n_train": int(train_np.shape[0]), "max_rel_logl": float(rel.max()), "max_abs_logl": float(np.abs(got - want).max()),
            "rel_slogl": float(rel_slogl), "rel_slogl_of_per_row_logl": float(rel_slogl_from_logl), "tol": tol, "abs_tol": abs_tol,
            "sum_of_device_logl_vs_timed_slogl_rel": float(sum_rel),
            "max_whitened_norm2_of_training_rows": max_z2, "seconds": time.perf_counter() - t0,
            "how": "oracle/pbn_oracle.cpp (reference arithmetic, fp64) on the first 1024 test rows + the 8 test rows nearest each of the 8 farthest-out "
                   "training rows, against ALL training rows; device side: slogl of exactly those rows (`rel_slogl`, the timed kernel) and the per-row "
                   "logl of all test rows (`max_rel_logl`; its sum against the timed step's slogl)"}


def cpu_baseline(train_np, test_np, h, budget_s=12.0):
    """CPU side of the headline metric on the host cores of the GPU box, bounded sample.  `value` = the tuned CPU form
    (oracle/pbn_baseline.cpp: whitened once, training tiles reused by blocks of 64 test rows, vectorised exponentials, OpenMP over
    query blocks; built here with -Ofast -march=native) on all cores; beside it the same on ONE thread (the reference's KDE path is
    single-threaded on the host side), the scaling efficiency, and `reference_arithmetic_port` = the checker (pbn_oracle.cpp: the
    reference's per-pair arithmetic, kde/ProductKDE.hpp:240-293, scalar exp, -O2) that round 1-3 reported as the baseline."""
    from oracle import baseline, oracle

    full = np.ndim(h) == 2
    fast = baseline.kde_logl if full else baseline.product_kde_logl
    slow = oracle.kde_logl if full else oracle.product_kde_logl
    visible = baseline.num_threads()
    quota = cpu_quota()   # a container may see every core and still be held to a CPU-time quota: more threads than that only thrash
    try:
        affinity = len(os.sched_getaffinity(0))
    except Exception:
        affinity = None
    cores = host_cores(visible)
    baseline.set_num_threads(cores)
    oracle.set_num_threads(cores)
    n_train = train_np.shape[0]

    def timed(fn, rows):
        t0 = time.perf_counter()
        r = fn(train_np, h, test_np[:rows])
        return time.perf_counter() - t0, r

    def sized(fn, threads, budget):
        probe = min(test_np.shape[0], 64 * threads)             # one block of 64 test rows per thread
        dt, _ = timed(fn, probe)
        rows = int(min(test_np.shape[0], max(probe, probe * budget / max(dt, 1e-3))))
        rows = max(probe, rows // (64 * threads) * (64 * threads))
        dt, r = timed(fn, rows)
        return rows, dt, r

    fast(train_np[:4096], h, test_np[:16])                      # build / load outside the timed calls
    rows, dt, r_fast = sized(fast, cores, budget_s * 0.45)
    chk = slow(train_np, h, test_np[:min(rows, 4 * oracle.num_threads())])   # the tuned form against the checker
    try:
        phys = len(set(zip(*[[l_.split(":")[1].strip() for l_ in open("/proc/cpuinfo") if l_.startswith(key)] for key in ("physical id", "core id")])))
    except Exception:
        phys = None
    out = {
        "value": rows / dt / 1e6, "unit": "M-samples/s", "cores": cores, "kind": "port",
        "pairs_per_s": rows * n_train / dt, "pairs_per_s_per_thread": rows * n_train / dt / cores,
        "max_rel_vs_checker": float(np.max(np.abs(r_fast[:chk.size] - chk) / np.maximum(1.0, np.abs(chk)))),
        "cpu": baseline.cpu_model(), "physical_cores": phys, "visible_cpus": visible, "cpu_quota_cores": quota, "cpu_affinity": affinity,
        "cores_note": "threads = min(visible CPUs, affinity, the container's cgroup CPU quota): the GPU boxes of this pool show 256 CPUs under a 16-CPU quota",
        "sample": f"{rows} test rows x {n_train} training rows, d={D}, fp64, {dt:.1f}s wall; oracle/pbn_baseline.cpp (whitened, blocked 64 x 1024, "
                  f"libmvec exponentials, -Ofast -march=native), OpenMP over blocks of 64 test rows",
    }
    try:
        baseline.set_num_threads(1)
        rows1, d1, _ = sized(fast, 1, budget_s * 0.2)
        baseline.set_num_threads(cores)
        out["single_thread"] = {"value": rows1 / d1 / 1e6, "unit": "M-samples/s", "pairs_per_s": rows1 * n_train / d1,
                                "sample": f"{rows1} test rows, {d1:.1f}s wall, same code on one thread"}
        out["all_cores_over_one_thread"] = (rows / dt) / (rows1 / d1)
        out["scaling_efficiency"] = (rows / dt) / (rows1 / d1) / cores
        # the checker's own speed (what rounds 1-3 reported as the CPU baseline): reference arithmetic, scalar exp
        ocores = oracle.num_threads()
        rows_o, d_o, _ = sized(slow, ocores, budget_s * 0.2)
        oracle.set_num_threads(1)
        one = max(8, min(32, rows_o // max(ocores, 1)))
        d_o1, _ = timed(slow, one)
        oracle.set_num_threads(ocores)
        out["reference_arithmetic_port"] = {"value": rows_o / d_o / 1e6, "unit": "M-samples/s", "cores": ocores,
                                            "single_thread": one / d_o1 / 1e6,
                                            "sample": f"{rows_o} test rows in {d_o:.1f}s on {ocores} threads, {one} rows in {d_o1:.1f}s on one: pbn_oracle.cpp, "
                                                      f"the reference's operation order (difference, divide, square per pair; max -> exp -> sum -> log), -O2"}
        from scipy.stats import gaussian_kde

        sub = train_np[:: max(1, n_train // 100_000)]          # scipy needs N x m memory: 1e5 training rows
        k = gaussian_kde(sub.T)
        m = min(200, test_np.shape[0])
        t0 = time.perf_counter()
        k.logpdf(test_np[:m].T)
        d2 = time.perf_counter() - t0
        scale = n_train / sub.shape[0]
        out["scipy_gaussian_kde"] = {"value": m / (d2 * scale) / 1e6, "unit": "M-samples/s",
                                     "sample": f"full-covariance gaussian_kde.logpdf, {m} test rows x {sub.shape[0]} training rows in {d2:.1f}s, "
                                               f"scaled linearly to {n_train} training rows"}
    except Exception as ex:
        out["extra_error"] = f"{type(ex).__name__}: {ex}"
    finally:
        baseline.set_num_threads(visible)
        oracle.set_num_threads(visible)
    return out


def _num(v, sig=6):
    """Numbers of the compact line: 6 significant digits (the verbose object keeps the full doubles)."""
    if isinstance(v, bool) or v is None or isinstance(v, (int, str)):
        return v
    if isinstance(v, float):
        if v != v or v in (float("inf"), float("-inf")):
            return None
        return float(f"{v:.{sig}g}")
    return v


def _pick(src, *keys, **renamed):
    """{key: number} of the keys `src` holds (dict values and prose are left to the verbose object)."""
    out = {}
    if not isinstance(src, dict):
        return out
    for k in keys:
        if k in src and not isinstance(src[k], (dict, list)):
            out[k] = _num(src[k])
    for new, old in renamed.items():
        if old in src and not isinstance(src[old], (dict, list)):
            out[new] = _num(src[old])
    return out


def compact_line(out, full_name):
    """The ONE stdout line, numbers only and under 5 KB (the driver's record keeps the parsed headline keys and the last ~6 KB of the line:
    round 5's 25 KB line lost its C3 / C4 / cv_weak legs there).  Everything else - prose, methods, samples, per-flip lists - goes to the
    verbose object `full_name` beside it."""
    head = {k: _num(out[k]) for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                                     "vs_baseline", "dtype", "data") if k in out}
    cfg = out.get("config", {})
    head["config"] = {k: _num(cfg.get(k), 13) for k in ("workload", "parallelism", "ranks", "backend", "ranks_seen", "rccl_world1_child",
                                                    "pairs_per_step_per_gpu", "slogl_step0_rank_sum") if k in cfg}
    if cfg.get("rank_devices"):
        head["config"]["rank_devices"] = [list(d[:2]) for d in cfg["rank_devices"] if d]
    r = out.get("roofline", {})
    head["roofline"] = _pick(r, "kernel", "bound", "achieved", "peak", "unit", "frac", "traffic", "avg_launch_ms", "exp_unit")
    if isinstance(r.get("dp_issue_util"), dict):
        head["roofline"].update(_pick(r["dp_issue_util"], dp_issue_util="value", mfma_busy="mfma_busy_frac"))
    c = out.get("cpu_baseline")
    if isinstance(c, dict):
        head["cpu_baseline"] = _pick(c, "value", "unit", "cores", "kind", "pairs_per_s_per_thread", "scaling_efficiency")
        head["cpu_baseline"]["sample"] = str(c.get("sample", ""))[:96]
        if isinstance(c.get("single_thread"), dict):
            head["cpu_baseline"]["single_thread"] = _num(c["single_thread"].get("value"))
        if isinstance(c.get("reference_arithmetic_port"), dict):
            head["cpu_baseline"]["reference_arithmetic"] = _num(c["reference_arithmetic_port"].get("value"))
    par = ("ok", "rows", "rel_slogl", "max_rel_logl", "max_abs_logl", "tol", "error")
    if isinstance(out.get("parity"), dict):
        head["parity"] = _pick(out["parity"], *par)

    def search_leg(v):
        o = _pick(v, "value", "unit", "estimate_s", "score_ctor_s", "cells_scored", "local_score_evals", "iterations", "arcs_found", "ranks", "nodes",
                  "scaling", "mmpc_s", "ci_tests", "cpc_edges", "seconds", "cold_seconds", "near_tie_redos", "device_over_cpu", "error")
        rf = v.get("roofline")
        if isinstance(rf, dict):
            o["roofline"] = _pick(rf, "achieved", "peak", "unit", "frac", "frac_of_measured_copy", "launch_us", "mfma_tflops", "mfma_frac_of_fp64_peak",
                                  "mfma_busy", "traffic", "device_s", "launches", "share_of_estimate_s", "moment_s", "moment_pairs", "moment_frac",
                                  "moment_cycles_per_pair", kernel="kernel_short")
            o["roofline"]["bound"] = str(rf.get("bound", "")).split(" ")[0]
        cb = v.get("cpu_baseline")
        if isinstance(cb, dict):
            o["cpu"] = _pick(cb, "value", "cores", "kind", "extrapolated", "same_structure", "same_skeleton")
            if isinstance(cb.get("single_thread"), dict):
                o["cpu"]["single_thread"] = _num(cb["single_thread"].get("value"))
        ta = v.get("tie_accounting")
        if isinstance(ta, dict):
            o["ties"] = _pick(ta, "replayed_iterations", "tie_flips", "non_tie_divergences", "max_rel_gap", "end_gain", "error")
        er = v.get("eight_rank_estimate")
        if isinstance(er, dict):
            o["eight_rank_emulation"] = _pick(er, "per_rank_s", "slowest_over_mean_share", "ratio_to_one_rank_leg")
        if v.get("per_rank_estimate_s"):
            o["per_rank_estimate_s"] = [_num(x, 4) for x in v["per_rank_estimate_s"]]
        return o

    legs = {}
    sec = out.get("secondary") or {}
    for name, key in ((sec.get("which") or ("c4" if "C4" in str(sec.get("config", "")) else "cv64"), "secondary"), ("c1", "secondary_c1"),
                      ("c3", "secondary_c3"), ("c5", "secondary_c5"), ("cv_weak", "secondary_cv_weak")):
        if isinstance(out.get(key), dict):
            legs[name] = search_leg(out[key])
    f = out.get("secondary_f32")
    if isinstance(f, dict):
        legs["f32"] = _pick(f, "value", "unit", "ms_per_step", "rel_diff_vs_f64", "error")
        if isinstance(f.get("roofline"), dict):
            legs["f32"]["roofline"] = _pick(f["roofline"], "kernel", "bound", "avg_launch_ms", "frac", "frac_of_issue_bound")
        if isinstance(f.get("parity"), dict):
            legs["f32"]["parity"] = _pick(f["parity"], *par)
    w = out.get("rccl_world1")
    if isinstance(w, dict):
        legs["rccl_world1"] = _pick(w, "ok", "world", "rccl_ranks_seen", "collectives", "all_gather_identity", "hip_runtime_of_torch", "hip_runtime_pairing", "seconds")
        if isinstance(w.get("bit_identical"), dict):
            legs["rccl_world1"]["bit_identical"] = all(bool(x) for x in w["bit_identical"].values())
        if "error" in w:
            legs["rccl_world1"]["error"] = str(w["error"])[:160]
    e = out.get("e2e_host")
    if isinstance(e, dict):
        legs["e2e_host"] = _pick(e, "value", "unit", "ms", "fit_ms", "fit_first_ms")
    if legs:
        head["legs"] = legs
    head["full"] = full_name
    return head


def write_full(out, name="bench_full.json"):
    """The verbose object (every leg with its prose) beside the line: ./bench_full.json, and gpurun_out/ when that directory exists."""
    text = json.dumps(out, indent=1)
    for d in (os.getcwd(), os.path.join(ROOT, "gpurun_out")):
        try:
            if os.path.isdir(d):
                with open(os.path.join(d, name), "w") as fh:
                    fh.write(text)
        except OSError:
            pass
    return name


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


def rccl_world1():
    """Runs tools/rccl_world1.py as a child process (it initialises its own GPU context and a one-rank RCCL communicator) and returns its
    JSON; a failure is reported in the object, it never costs the line."""
    import socket
    import subprocess

    try:
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), OMP_NUM_THREADS="4")
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        env.pop("PBN_FORCE_DIST", None)
        p = subprocess.run([sys.executable, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools", "rccl_world1.py")], env=env,
                           capture_output=True, text=True, timeout=600)
        lines = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")]
        if not lines:
            return {"ok": False, "error": f"exit code {p.returncode}: {p.stderr[-600:]}"}
        res = json.loads(lines[-1][len("RESULT "):])
        res["what"] = ("a fresh child process: torch.distributed 'nccl' (RCCL) with ONE rank on this GPU, the one-process-per-GPU mode forced "
                       "(PBN_FORCE_DIST): CV-likelihood CKDE / hybrid / BGe / BIC hill-climbs and a sharded KDE slogl with every batch's all-gather "
                       "through RCCL, compared bit for bit with the plain calls; `mapped` = the HIP / RCCL libraries that process loaded")
        return res
    except Exception as ex:
        return {"ok": False, "error": f"{type(ex).__name__}: {ex}"}


def e2e_host(pbn, kde, names, test_np, repeats=3, train_np=None):
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
    out = {"value": rb.num_rows / best / 1e6, "unit": "M-samples/s", "ms": best * 1e3, "slogl": val,
           "what": f"{kde.__class__.__name__}.slogl(pyarrow.RecordBatch in host memory, {rb.num_rows} rows): H2D of the test "
                   f"columns + query pack + sweep + finish + D2H of the scalar, best of {repeats}"}
    if train_np is not None:
        # the fit a reference user pays before the first slogl (KDE::fit: upload, covariance, bandwidth, whitening + packing of the training rows)
        fits = []
        for rep in range(3):
            # a NEW record batch per repetition: model-level calls share the upload of one and the same frame (dataset.shared_upload)
            trb = pa.RecordBatch.from_arrays([pa.array(np.ascontiguousarray(train_np[:, i] + 0.0)) for i in range(train_np.shape[1])], names=names)
            k2 = kde.__class__(names)
            t0 = time.perf_counter()
            k2.fit(trb)
            pbn.default_context().sync()
            fits.append(time.perf_counter() - t0)
        out["fit_ms"] = min(fits[1:]) * 1e3
        out["fit_first_ms"] = fits[0] * 1e3
        out["fit_what"] = (f"{kde.__class__.__name__}.fit(pyarrow.RecordBatch in host memory, {trb.num_rows} rows x {trb.num_columns}): H2D of the "
                           f"training columns + covariance (device Gram) + bandwidth + whitening / packing; best of 2 after the first call (fit_first_ms: "
                           f"the first, which also grows the context's arenas)")
    return out


def dp_issue_util():
    """Share of the FP64 issue slots the sweep kept busy, from the committed rocprofv3 PMC pass of this command:
    (MFMA busy cycles + 4 cycles per VALU wave-instruction, 8 for the quarter-rate v_exp_f32) / (cycles x 1024 SIMDs).  Null with the reason when the pass was
    taken on another revision of the kernel source (pmc_record)."""
    k, why = pmc_record(SWEEP_KERNEL)
    if k is None:
        return {"value": None, "reason": why}
    simd_cycles = k["GRBM_GUI_ACTIVE"] / 8.0 * 1024.0
    # rounds 4-6: the sum-only sweep takes 2^f on the fp32 transcendental unit (round 6: from the accumulator's own words, exp2_magic) - one v_exp_f32 per pair value, which holds the issue port
    # for 8 cycles (a quarter-rate instruction: two slots of 4) - so the slot count is the instruction count plus one per value
    exp32 = 1e11 / 64.0 if k["SQ_INSTS_VALU"] * 64.0 / 1e11 < 9.5 else 0.0
    return {"value": (k["SQ_VALU_MFMA_BUSY_CYCLES"] + 4.0 * (k["SQ_INSTS_VALU"] + exp32)) / simd_cycles,
            "mfma_busy_frac": k["SQ_VALU_MFMA_BUSY_CYCLES"] / simd_cycles,
            "valu_insts_per_pair_value": k["SQ_INSTS_VALU"] * 64.0 / 1e11,
            "quarter_rate_insts_per_pair_value": exp32 * 64.0 / 1e11,
            "source": f"profiles/{k['_round']}/pmc_per_dispatch.json (separate rocprofv3 --pmc pass on kde_kernels.hip blob {k['_blob'][:12]}, not this run)"}


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
    ap.add_argument("--no-extra-legs", action="store_true", help="skip the secondary_* legs of the default run (C3, C5, C1, f32, weak-scaling CV)")
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
    ranks_seen, devs = None, None
    if world > 1:
        import torch.distributed as dist

        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # the backends' own chatter ("[Gloo] Rank 0 is connected to ...", NCCL_DEBUG lines) goes to C-level stdout: send it to stderr while
        # the group comes up, so that stdout carries the ONE JSON line only
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            if args.backend == "nccl":
                dist.init_process_group("nccl", device_id=device)
            else:
                dist.init_process_group("gloo")
            # the collective really runs over `world` ranks of this backend: one all_reduce of ones (nccl: on the device = RCCL over xGMI)
            ones = torch.ones(1, dtype=torch.float64, device=device if args.backend == "nccl" else "cpu")
            dist.all_reduce(ones)
            ranks_seen = int(round(float(ones.item())))
            devs = [None] * world
            dist.all_gather_object(devs, (rank, local_rank, torch.cuda.get_device_properties(local_rank).name))
            dist.barrier()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

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
    legs = {}
    exit_code = 0

    def leg(name, fn):
        try:
            legs[name] = fn()
        except Exception as ex:  # a secondary leg must never cost the headline line
            legs[name] = {"metric": "hill-climb candidate-arcs scored/s", "value": None, "error": f"{type(ex).__name__}: {ex}"}
        if dist is not None:     # every rank's own time for the leg (the value is rank 0's; all ranks run the same search in lock step)
            per = [None] * world
            dist.all_gather_object(per, legs[name].get("estimate_s"))
            legs[name]["per_rank_estimate_s"] = per

    if args.no_cpu_baseline:
        os.environ["PBN_BENCH_NO_CPU"] = "1"
    global DEFERRED_CPU
    DEFERRED_CPU = []   # the legs' CPU parts run after the last leg has been timed (bench_hill_climb: cpu_part)
    if args.hc != "none":
        # N > 1: the north star's scaling workload in its fixed-work form (64 nodes on every world size: strong scaling of the
        # search) with update batches inside the timed region (max_iters >= 5)
        iters = args.hc_max_iters or (5 if (args.hc == "cv64" and world > 1) else 0)
        leg("secondary", lambda: bench_hill_climb(torch, pbn, _lib, ctx, device, args.hc, args.hc_rows, iters))
        hc_out = legs.pop("secondary")
        if dist is not None and hc_out.get("value") is not None:
            hc_out["ranks"] = world
            hc_out["scaling"] = "strong"
            hc_out["sharding"] = ("the unknown CKDE terms A(joint), A(marginal) of every delta-cache batch dealt to the ranks (longest processing time first), "
                                  "one all_gather of their totals per batch, every rank assembles the deltas from the same doubles; fixed total work "
                                  "(strong scaling of the search)")
    if hc_auto and not args.no_extra_legs:
        # the north star's weak form: delta cells per rank constant - n = round(64 sqrt(N / 8)) nodes (23 on one GPU ... 64 on
        # eight), 100 k rows, cache_scores + 5 iterations
        wn = int(round(64.0 * (world / 8.0) ** 0.5))
        leg("secondary_cv_weak", lambda: bench_hill_climb(torch, pbn, _lib, ctx, device, "cv64", args.hc_rows, 5, n_cols=wn))
        if legs["secondary_cv_weak"].get("value") is not None:
            legs["secondary_cv_weak"].update({"ranks": world, "scaling": "weak", "nodes": wn,
                                              "note": "cells scale with nodes^2: nodes = round(64 sqrt(ranks / 8)) keeps cells per rank constant"})
        if world == 1 and dist is None and legs["secondary_cv_weak"].get("value") is not None:
            legs["secondary_cv_weak"]["eight_rank_estimate"] = eight_rank_estimate(
                lambda: bench_hill_climb(torch, pbn, _lib, ctx, device, "cv64", args.hc_rows, 5, n_cols=64, cpu=False), legs["secondary_cv_weak"])
    if hc_auto and world == 1 and not args.no_extra_legs:
        if not args.no_c3:
            # the CKDE path of the search, driver-timed: BASELINE config 3 at FULL size (32-node SPBN, 10-fold CV, 500 k rows),
            # bounded to the initial delta cache + one iteration
            leg("secondary_c3", lambda: bench_hill_climb(torch, pbn, _lib, ctx, device, "c3", 0, 1))
        # BASELINE config 5 end to end on one GPU: MMPC (hybrid MutualInformation) + ValidatedLikelihood hill-climb to convergence
        leg("secondary_c5", lambda: bench_hill_climb(torch, pbn, _lib, ctx, device, "c5mmhc", 0, 1_000_000))
        if dist is None and legs["secondary_c5"].get("value") is not None:
            # BASELINE config 5 is an 8-GPU configuration: the same estimate for its hill-climb phase (fixed work: strong scaling; the
            # stand-in skeleton of `--hc c5`, the MMPC phase in front of it is not part of the estimate)
            def c5_one():
                return bench_hill_climb(torch, pbn, _lib, ctx, device, "c5", 0, 1_000_000, cpu=False)

            base5 = c5_one()
            legs["secondary_c5"]["eight_rank_estimate"] = dict(eight_rank_estimate(c5_one, base5), one_rank_s=base5["estimate_s"],
                                                               phase="hill-climb over a random 15 % skeleton (--hc c5), fixed work")
        leg("secondary_c1", lambda: bench_c1(pbn))

        def f32_leg():
            tr32, te32 = make_tables(torch, device, args.n_train, args.n_test, 0, 1, torch.float32)
            torch.cuda.synchronize()
            a = pbn.DeviceTable.from_device_pointer(ctx, tr32.data_ptr(), args.n_train, names, args.n_train, _lib.PBN_F32, keepalive=tr32)
            b = pbn.DeviceTable.from_device_pointer(ctx, te32.data_ptr(), args.n_test, names, args.n_test, _lib.PBN_F32, keepalive=te32)
            k32 = (pbn.ProductKDE if args.kde == "product" else pbn.KDE)(names)
            k32.fit_table(a)
            buf = torch.zeros(8, dtype=torch.float64, device=device)
            for i in range(2):
                k32.slogl_table_async(b, buf.data_ptr() + 8 * i)
            ctx.sync()
            ctx.set_profiling(True)
            t0 = time.perf_counter()
            for i in range(5):
                k32.slogl_table_async(b, buf.data_ptr() + 8 * (2 + i))
            ctx.sync()
            el = time.perf_counter() - t0
            ms, nl = ctx.kernel_time(_lib.PBN_K_SWEEP)
            ctx.set_profiling(False)
            pairs_ = float(args.n_train) * float(args.n_test)
            peak_pairs = 256 * 4 * 2.4e9 * 64.0 / F32_VALU_CYCLES_PER_VALUE
            par32 = None
            if not args.no_cpu_baseline:
                try:   # north star: fp32 slogl within 1e-3 relative; per row the reference tests' own fp32 tolerance (atol 5e-4; 5e-3 ProductKDE)
                    par32 = parity_block(torch, k32, b, tr32, te32, float(buf[2].item()), 1e-3, abs_tol=5e-3 if args.kde == "product" else 5e-4)
                except Exception as ex:
                    par32 = {"ok": False, "error": f"{type(ex).__name__}: {ex}"}
            # the whole instruction stream of the W32 kernel at d = 8: per 1 024 pair values (16 per lane) 16 v_exp_f32 (8 issue cycles) + 16 v_add_f32 (4)
            # + two v_mfma_f32_32x32x16_f16 (8 each) = 13 issue cycles per value
            issue_pairs = 256 * 4 * 2.4e9 * 64.0 / 13.0
            return {"metric": "KDE slogl M-samples/s", "dtype": "f32", "parity": par32, "value": args.n_test * 5 / el / 1e6, "unit": "M-samples/s", "ms_per_step": el / 5 * 1e3,
                    "slogl": float(buf[2].item()), "rel_diff_vs_f64": abs(float(buf[2].item()) - slogl) / abs(slogl),
                    "roofline": {"kernel": "kde_sweep_f16_w32_kernel", "bound": "valu-issue", "avg_launch_ms": ms / max(nl, 1),
                                 "frac": pairs_ / (ms / max(nl, 1) * 1e-3) / peak_pairs,
                                 "frac_of_issue_bound": pairs_ / (ms / max(nl, 1) * 1e-3) / issue_pairs,
                                 "note": "against the VALU-issue bound of the fp32 sweep (one v_exp_f32 + one v_add_f32 per pair value); the f32 "
                                         "coordinates run as two f16 pieces on the 16-bit matrix cores (DESIGN.md 3.1)"}}

        slogl = float(partial[args.warmup].item())
        leg("secondary_f32", f32_leg)

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
            # fp32 sweep: the dot products run as f16x2 on the 16-bit matrix cores and overlap with the VALU, which is
            # the binding unit: per pair value one v_exp_f32 (quarter rate: 8 issue cycles per wave64 on this part,
            # profiles/r1/microbench_r1.txt) and one v_add_f32 (4) at least.  Peak pair rate by VALU issue =
            # SIMDs x clock x 64 / 12 cycles, expressed in the same algorithmic flops (3d+2 per pair)
            peak_pairs = 256 * 4 * 2.4e9 * 64.0 / F32_VALU_CYCLES_PER_VALUE
            peak_tf = peak_pairs * FLOPS_PER_PAIR / 1e12
            peak_note = (f"against the VALU-issue bound of the fp32 sweep: one v_exp_f32 (8 issue cycles per wave64) + one v_add_f32 (4) per "
                         f"pair value -> {peak_pairs / 1e12:.2f}e12 pairs/s x (3d+2) flop; the distances run as f16x2 on the 16-bit matrix "
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
                "rccl_ranks_seen": ranks_seen if args.backend == "nccl" else None,
                "ranks_seen": ranks_seen,
                "rank_devices": devs,
                "pairs_per_step_per_gpu": pairs,
                "slogl_step0_rank_sum": slogl,
            },
            "roofline": {
                "kernel": "kde_sweep_kernel" if args.dtype == "f64" else "kde_sweep_f16_w32_kernel",
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
        if args.dtype == "f64":
            out["roofline"]["dp_issue_util"] = dp_issue_util()
        if hc_out is not None:
            out["secondary"] = hc_out
        out.update(legs)
        if world == 1 and hc_auto and not args.no_extra_legs:
            # the RCCL path of the sharded delta cache, executed on this one GPU by a fresh child process (tools/rccl_world1.py): a
            # one-rank "nccl" group, the product's one-process-per-GPU mode forced, every batch's all-gather through RCCL on device
            # buffers, results bit-identical to the plain calls; never part of a timed region
            out["rccl_world1"] = rccl_world1()
            out["config"]["rccl_world1_child"] = True   # the collective evidence of this one-GPU line is that child; the timed region ran with dist = None
        if world == 1 and not args.no_e2e:
            try:
                out["e2e_host"] = e2e_host(pbn, kde, names, test_t.T.cpu().numpy(), train_np=train_t.T.cpu().numpy())
            except Exception as ex:
                out["e2e_host"] = {"value": None, "error": f"{type(ex).__name__}: {ex}"}
        for fn in DEFERRED_CPU or []:   # every leg is timed: now the CPU sides (baselines on the legs' own work lists, tie accounting)
            try:
                fn()
            except Exception as ex:      # a CPU part must never cost the line
                print(f"bench.py: a leg's CPU part failed: {type(ex).__name__}: {ex}", file=sys.stderr, flush=True)
        DEFERRED_CPU = None
        if world == 1 and not args.no_cpu_baseline:
            # the oracle against THIS run (BASELINE size): north star bar 1e-6 relative in fp64, 1e-3 in fp32
            try:
                out["parity"] = parity_block(torch, kde, test, train_t, test_t, slogl, 1e-6 if args.dtype == "f64" else 1e-3,
                                             abs_tol=None if args.dtype == "f64" else (5e-3 if args.kde == "product" else 5e-4))
            except Exception as ex:
                out["parity"] = {"ok": False, "error": f"{type(ex).__name__}: {ex}"}
            h = np.asarray(kde.bandwidth, dtype=np.float64)
            sample_rows = min(args.n_test, 65536)
            train_np = train_t.T.cpu().numpy().astype(np.float64)
            test_np = test_t[:, :sample_rows].T.cpu().numpy().astype(np.float64)
            try:
                out["cpu_baseline"] = cpu_baseline(train_np, test_np, h)
            except Exception as ex:  # never lose the headline line to the baseline leg
                out["cpu_baseline"] = {"value": None, "unit": "M-samples/s", "cores": 0, "kind": "port", "sample": f"failed: {ex}"}
        print(json.dumps(compact_line(out, write_full(out)), separators=(",", ":")), flush=True)
        bad = [k_ for k_, v_ in [("headline", out)] + list(legs.items()) if isinstance(v_.get("parity"), dict) and not v_["parity"].get("ok")]
        bad += [k_ for k_, v_ in [("secondary", hc_out or {})] + list(legs.items())
                if isinstance(v_.get("tie_accounting"), dict) and v_["tie_accounting"].get("non_tie_divergences", 0) > 0]
        if bad:
            print(f"bench.py: PARITY FAILED in {bad} (see the `parity` / `tie_accounting` objects of the line above)", file=sys.stderr, flush=True)
            exit_code = 3
    if dist is not None:
        dist.destroy_process_group()
    sys.exit(exit_code)


if __name__ == "__main__":
    main()
