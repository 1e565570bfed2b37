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


def test_deal_sets_longest_first_is_balanced_and_deterministic():
    """distributed.deal_sets: CKDE variable sets are dealt longest-processing-time first on the cost model (a d = 5 candidate
    costs 3-4x a d = 2 one): no rank carries more than the mean load + one set, and the dealing is a pure function of the
    batch (every rank computes the same owners)."""
    from pybnesian_amd.distributed import _kde_cost, deal_sets

    rng = np.random.default_rng(5)
    keys = []
    for _ in range(126):
        d = int(rng.integers(1, 6))
        keys.append((tuple(sorted(rng.choice(64, size=d, replace=False).tolist())), int(rng.integers(1, 3))))
    cost = [_kde_cost(len(k)) + n * _kde_cost(len(k) - 1) for k, n in keys]
    for world in (2, 3, 8):
        owner = deal_sets(keys, world)
        assert owner == deal_sets(list(keys), world)
        load = [sum(c for c, o in zip(cost, owner) if o == r) for r in range(world)]
        assert all(0 <= o < world for o in owner)
        assert max(load) <= sum(cost) / world + max(cost)
        # round-robin in order of first appearance (round 2) is worse or equal on this mix
        rr = [sum(c for i, c in enumerate(cost) if i % world == r) for r in range(world)]
        assert max(load) <= max(rr) + 1e-9
