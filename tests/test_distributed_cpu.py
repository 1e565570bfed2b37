"""CPU tier, world_size 2 over gloo: the N>1 path of the delta-score cache.  Every rank scores its share of
each batch (candidate i belongs to rank i % world), one all_gather per batch rebuilds the full result in
rank order, and both ranks must take exactly the decisions of a single-process run."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from test_hc_cpu import TableScore, run_product


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n, seed, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        ts = TableScore(n, seed, validated=True)
        arcs, types, trace, last = run_product(ts, "spbn", n, [0] * n, op_types=True, patience=1)
        queue.put((rank, arcs, types, trace, last.cells_scored, ts.calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("seed", [0, 3])
def test_sharded_delta_cache_world2(ensure_built, seed):
    n, world = 9, 2
    ts = TableScore(n, seed, validated=True)
    ref = run_product(ts, "spbn", n, [0] * n, op_types=True, patience=1)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=120) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    calls = []
    for rank, arcs, types, trace, cells, ncalls in results:
        assert trace == ref[2] and sorted(arcs) == sorted(ref[0]) and types == ref[1]
        assert cells == ref[3].cells_scored
        calls.append(ncalls)
    # CKDE-typed candidates are split over the ranks, LinearGaussian ones are recomputed by every rank
    assert max(calls) < ts.calls and sum(calls) >= ts.calls


def test_shard_indices_partition():
    from pybnesian_amd.distributed import shard_indices

    for n in (0, 1, 7, 64):
        for world in (1, 2, 8):
            parts = [shard_indices(n, r, world) for r in range(world)]
            assert sorted(i for p in parts for i in p) == list(range(n))


def _mmpc_worker(rank, world, port, n, seed, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pybnesian_amd.independences import mmpc_cpcs
        from test_mmpc_cpu import TableTest

        t = TableTest(n, seed, 500)
        cpcs, ntests = mmpc_cpcs(t, t.names, 0.05)
        queue.put((rank, cpcs, ntests, t.calls))
    finally:
        dist.destroy_process_group()


def test_sharded_mmpc_world2(ensure_built):
    """The independence tests of every batched MMPC step are dealt over the ranks and gathered: both ranks end with the
    single-process CPCs (same members, same libstdc++ set order, same number of tests) while evaluating fewer tests."""
    from pybnesian_amd.independences import mmpc_cpcs
    from test_mmpc_cpu import TableTest

    n, seed, world = 10, 4, 2
    ref_t = TableTest(n, seed, 500)
    ref, ref_tests = mmpc_cpcs(ref_t, ref_t.names, 0.05)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_mmpc_worker, args=(r, world, port, n, seed, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = [q.get(timeout=180) for _ in range(world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    for rank, cpcs, ntests, calls in results:
        assert cpcs == ref and ntests == ref_tests
        assert calls < ref_t.calls
    assert sum(r[3] for r in results) >= ref_t.calls


class _FailingScore:
    """Stands in for a device score: rank `bad` raises while computing its share."""

    def __init__(self, bad):
        self.bad = bad

    def _batch_raw(self, model, var, ntype, off, par, kind):
        if dist.get_rank() == self.bad:
            raise ValueError("boom on purpose")
        return np.asarray([float(v) for v in var])


def _failing_worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pybnesian_amd.distributed import sharded_batch

        n = 8
        try:
            sharded_batch(_FailingScore(1), None, list(range(n)), [1] * n, list(range(n + 1)), [(i + 1) % n for i in range(n)], 0, shard_all=True)
            queue.put((rank, "no error"))
        except ValueError as ex:
            queue.put((rank, "ValueError: " + str(ex)))
        except RuntimeError as ex:
            queue.put((rank, "RuntimeError: " + str(ex)))
        # the group is still usable: nobody was left behind in the collective
        ok = sharded_batch(_FailingScore(-1), None, list(range(n)), [1] * n, list(range(n + 1)), [(i + 1) % n for i in range(n)], 0, shard_all=True)
        queue.put((rank, list(ok)))
    finally:
        dist.destroy_process_group()


def test_failed_rank_does_not_hang_the_collective(ensure_built):
    """A rank that raises while scoring its share still enters the all_gather (with an error flag); every rank raises
    afterwards - the failing one its own exception, the others a RuntimeError naming it - and the next batch works."""
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_failing_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    got = [q.get(timeout=120) for _ in range(2 * world)]
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    first = {r: v for r, v in got if isinstance(v, str)}
    second = {r: v for r, v in got if not isinstance(v, str)}
    assert first[1].startswith("ValueError: boom") and first[0].startswith("RuntimeError") and "[1]" in first[0]
    assert second[0] == second[1] == [float(i) for i in range(8)]


def test_deal_sets_longest_first_is_balanced_and_deterministic(ensure_built):
    """pbn_shard_deal on pbn_shard_term_cost (distributed.deal_sets): CKDE variable sets are dealt longest-processing-time first on the
    cost model (rows a sweep of the term's dimension meets x test rows): no rank carries more than the mean load + one set, and the
    dealing is a pure function of the batch (every rank computes the same owners)."""
    from pybnesian_amd.distributed import deal_sets, term_cost

    rng = np.random.default_rng(5)
    keys = []
    for _ in range(126):
        d = int(rng.integers(1, 6))
        keys.append((tuple(sorted(rng.choice(64, size=d, replace=False).tolist())), int(rng.integers(1, 3))))
    cost = [term_cost(len(k)) + n * term_cost(len(k) - 1) for k, n in keys]
    assert term_cost(5) > 2.5 * term_cost(2) > 0 and term_cost(0) == 0.0
    # pruned sweeps (d <= 4 on >= 32 768 training rows) meet ~ N^(4/(d+4)) rows per query: sub-linear in the training rows
    assert term_cost(2, 4_000_000, 1000) < 3.0 * term_cost(2, 1_000_000, 1000) and term_cost(6, 4_000_000, 1000) == 4.0 * term_cost(6, 1_000_000, 1000)
    for world in (2, 3, 8):
        owner = deal_sets(keys, world)
        assert owner == deal_sets(list(keys), world)
        load = [sum(c for c, o in zip(cost, owner) if o == r) for r in range(world)]
        assert all(0 <= o < world for o in owner)
        assert max(load) <= sum(cost) / world + max(cost)
        # round-robin in order of first appearance (round 2) is worse or equal on this mix
        rr = [sum(c for i, c in enumerate(cost) if i % world == r) for r in range(world)]
        assert max(load) <= max(rr) + 1e-9


class _FakeEngineScore:
    """Stands in for a device Score in distributed.sharded_batch: 4 continuous columns (ids 0-3), 2 discrete (4, 5); deterministic
    closed-form 'sweeps' with the engine's term / part interfaces, and a record of what was evaluated where."""

    class _T:
        names = ["a", "b", "c", "d"]

    def __init__(self, regions=1):
        self._table = self._T()
        self.regions = regions
        self.totals, self.evaluated, self.region_items, self.part_calls, self.raw_calls = {}, [], [], [], 0

    def _term_regions(self, kind):
        return self.regions

    @staticmethod
    def _a_region(kind, term, f):
        m, vs = term[0], sorted(term[1:])
        return float(np.sin(1.0 + kind + 0.37 * m + 1.7 * f + sum((i + 1) * 0.913 * (v + 1) for i, v in enumerate(vs))) * 1e3)

    def _a(self, kind, term):
        acc = 0.0
        for f in range(self.regions):   # a term's total: its regions added in region order
            acc += self._a_region(kind, term, f)
        return acc

    @staticmethod
    def _part(kind, v, ps, q):
        return float(np.cos(0.1 * q + kind + v + 0.77 * sum(ps))) if q % 5 else 0.0

    def _terms(self, what, kind, terms, values=None, regions=None):
        if what == "missing":
            return [0 if (kind,) + tuple([t[0]] + sorted(t[1:])) in self.totals else 1 for t in terms]
        if what == "eval_regions":
            self.region_items += [(kind,) + tuple([t[0]] + sorted(t[1:])) + (-1, f) for t, f in zip(terms, regions)]
            return np.array([self._a_region(kind, t, f) for t, f in zip(terms, regions)])
        if what == "eval":
            self.evaluated += [(kind,) + tuple([t[0]] + sorted(t[1:])) for t in terms]
            return np.array([self._a(kind, t) for t in terms])
        for t, v in zip(terms, values):
            self.totals[(kind,) + tuple([t[0]] + sorted(t[1:]))] = float(v)
        return None

    def _batch_parts(self, model, var, ntype, off, par, kind, part, n_parts):
        self.part_calls.append((part, n_parts, len(var)))
        out = np.zeros((len(var), 64))
        for i, v in enumerate(var):
            for q in range(part, 64, n_parts):
                out[i, q] = self._part(kind, v, par[off[i]: off[i + 1]], q)
        return out

    def _batch_raw(self, model, var, ntype, off, par, kind):
        self.raw_calls += 1
        out = np.zeros(len(var))
        for i, v in enumerate(var):
            ps = list(par[off[i]: off[i + 1]])
            if ntype[i] != 1:                                   # LinearGaussian / discrete: host arithmetic
                out[i] = 0.5 * v - 0.25 * sum(ps) + kind
            elif any(q >= 4 for q in ps):                       # hybrid CKDE: the 64 parts in order
                acc = 0.0
                for q in range(64):
                    acc += self._part(kind, v, ps, q)
                out[i] = acc
            else:                                               # continuous CKDE: joint term - marginal term
                d = len(ps) + 1
                key = lambda t: self.totals.get((kind,) + tuple([t[0]] + sorted(t[1:])), self._a(kind, t))
                out[i] = key((d, v) + tuple(ps)) - (key((d,) + tuple(ps)) if ps else 0.0)
        return out


_BATCH = dict(var=[0, 1, 2, 3, 0, 1, 2, 3, 0, 4, 2, 1], ntype=[1, 1, 1, 1, 1, 1, 0, 1, 1, 2, 1, 0],
              off=[0, 0, 1, 3, 4, 6, 8, 9, 11, 12, 13, 16, 17], par=[0, 0, 1, 0, 1, 4, 2, 5, 3, 4, 5, 2, 5, 0, 4, 3, 4])


def _term_worker(rank, world, port, queue, regions=1):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pybnesian_amd import _lib
        from pybnesian_amd.distributed import sharded_batch

        s = _FakeEngineScore(regions)
        first = sharded_batch(s, None, _BATCH["var"], _BATCH["ntype"], _BATCH["off"], _BATCH["par"], _lib.PBN_SCORE_CVLIK)
        again = sharded_batch(s, None, _BATCH["var"], _BATCH["ntype"], _BATCH["off"], _BATCH["par"], _lib.PBN_SCORE_CVLIK)   # every term known now
        queue.put((rank, first.tolist(), again.tolist(), s.evaluated + s.region_items, s.part_calls))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("regions,world", [(1, 2), (3, 3)])
def test_sharded_batch_deals_terms_and_slices_world2(ensure_built, regions, world):
    """pbn_shard_batch (through distributed.sharded_batch) with an engine score: continuous CKDE candidates by TERM (each unknown term
    evaluated on exactly one rank, totals installed everywhere, nothing evaluated again) - or, for a batch of few terms over several
    folds (regions = 3 on three ranks: the batch's 8 terms are fewer than 4 per rank), by (term, fold): each pair on exactly one rank, the
    folds added in fold order everywhere; hybrid CKDE candidates by SLICE (every rank its parts of every candidate), the rest
    redundantly - and every rank returns the one-process values bit for bit."""
    from pybnesian_amd import _lib

    ref = _FakeEngineScore(regions)._batch_raw(None, _BATCH["var"], _BATCH["ntype"], _BATCH["off"], _BATCH["par"], _lib.PBN_SCORE_CVLIK).tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_term_worker, args=(r, world, port, q, regions)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    evaluated = []
    for rank, first, again, ev, part_calls in results:
        assert first == ref and again == ref
        evaluated += ev
        assert part_calls == [(rank, world, 4), (rank, world, 4)]          # the four hybrid CKDE candidates, in both batches
    terms = set()
    for i, v in enumerate(_BATCH["var"]):
        ps = _BATCH["par"][_BATCH["off"][i]: _BATCH["off"][i + 1]]
        if _BATCH["ntype"][i] == 1 and v < 4 and all(q < 4 for q in ps):
            terms.add((_lib.PBN_SCORE_CVLIK, len(ps) + 1) + tuple(sorted([v] + ps)))
            if ps:
                terms.add((_lib.PBN_SCORE_CVLIK, len(ps) + 1) + tuple(sorted(ps)))
    assert len(terms) >= 6
    if regions > 1:
        terms = {t + (-1, f) for t in terms for f in range(regions)}
        per_rank = [len(ev) for _, _, _, ev, _ in results]
        assert max(per_rank) - min(per_rank) <= 2                                 # (term, fold) pairs of equal cost: dealt evenly
    assert len(evaluated) == len(set(evaluated)) and set(evaluated) == terms      # each term (or pair) once, on one rank, in the first batch only


def _lone_worker(rank, world, port, queue):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from pybnesian_amd import _lib, distributed
        from pybnesian_amd.distributed import sharded_batch

        calls = {"n": 0}
        real = distributed.Comm._all_gather if hasattr(distributed.Comm, "_all_gather") else None
        s = _FakeEngineScore(regions=5)
        # (1) ONE continuous CKDE candidate (2 | 0, 1): its two terms are dealt fold by fold - 10 (term, fold) pairs over the job, each once
        lone = sharded_batch(s, None, [2], [1], [0, 2], [0, 1], _lib.PBN_SCORE_CVLIK)
        ev_lone = list(s.region_items) + list(s.evaluated)
        # (2) a batch of LinearGaussian / discrete candidates only: nothing is dealt, no collective is made
        before = distributed.collective_calls() if hasattr(distributed, "collective_calls") else None
        light = sharded_batch(s, None, [0, 1, 4], [0, 0, 2], [0, 1, 3, 4], [1, 0, 2, 5], _lib.PBN_SCORE_CVLIK)
        after = distributed.collective_calls() if hasattr(distributed, "collective_calls") else None
        queue.put((rank, lone.tolist(), light.tolist(), ev_lone, s.raw_calls, None if before is None else after - before))
    finally:
        dist.destroy_process_group()


def test_a_lone_candidate_is_evaluated_once_in_the_job_and_light_batches_make_no_collective(ensure_built):
    """Round 6: a batch with exactly ONE heavy candidate used to be evaluated by every rank (both dealers wanted at least two); now its terms
    go round the ranks (term, fold) by (term, fold) like any small batch, every pair evaluated once across the job.  And a batch in which the
    plan deals nothing (LinearGaussian / discrete candidates, or every term already installed) makes no collective call at all."""
    from pybnesian_amd import _lib

    world = 3
    ref = _FakeEngineScore(regions=5)
    want_lone = ref._batch_raw(None, [2], [1], [0, 2], [0, 1], _lib.PBN_SCORE_CVLIK).tolist()
    want_light = ref._batch_raw(None, [0, 1, 4], [0, 0, 2], [0, 1, 3, 4], [1, 0, 2, 5], _lib.PBN_SCORE_CVLIK).tolist()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_lone_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    results = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    evaluated = []
    for rank, lone, light, ev, raw_calls, ncoll in results:
        assert lone == want_lone and light == want_light
        evaluated += ev
        if ncoll is not None:
            assert ncoll == 0, "a batch of light candidates made a collective call"
    k = _lib.PBN_SCORE_CVLIK
    pairs = {(k, 3, 0, 1, 2, -1, f) for f in range(5)} | {(k, 3, 0, 1, -1, f) for f in range(5)}
    assert len(evaluated) == 10 and set(evaluated) == pairs          # ten (term, fold) pairs, each on exactly one rank
    per_rank = [len(ev) for _, _, _, ev, _, _ in results]
    assert max(per_rank) - min(per_rank) <= 1
