"""GPU tier: the run-time switches that select between implementations or partitions of the same quantity must not change it.
Round 5 cut the switchboard to the documented knobs (tuning constants and the alternative paths of past experiments are
compiled to their defaults unless the library is built with `make EXPERIMENTS=1`) - what is left to hold is:
 * PBN_SCORE_LANES=1 (one issue lane) - the sums do not depend on the issue order: identical to the last bit;
 * PBN_SWEEP_PRUNE=0 (no tile pruning), PBN_SCORE_GROUPED=0 (one launch chain per (set, fold) instead of the grouped evaluation of
   kde_group.hip), PBN_GROUP_SPLIT_TILES (another partition of the same work), PBN_FAR_SPAN=0 (no fp32 tail for far tiles) - other
   partitions / paths of the same sums: equal to rounding, the fp32 sweeps to their own precision;
 * PBN_MI_FULL_BUDGET_MB=0 (per-test moment kernels instead of the per-grouping moments), PBN_MI_THREADS=1;
 * PBN_HYBRID_BATCH=0 (the hybrid candidates of a batch one by one instead of in one chain), PBN_HYBRID_BATCH_SLOTS, PBN_GROUP_ARENA_MB,
   PBN_HYBRID_GROUPINGS (batches cut by their slots, their arena, a reset of the grouping cache): bit-identical."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

HERE = os.path.dirname(os.path.abspath(__file__))


# The switches select between implementations of the SAME sums; what a pruned sweep drops depends on which tiles it visits, so the
# comparison is made with the pruning margins pinned where the dropped mass is below rounding-level tolerances (52: 2.2e-10 of a sum;
# fp32 40: 9e-7).  The default margins of the sum-only sweeps (43 / 36, prune_margin) are held against these in
# test_default_margins_stay_inside_their_bound.
PINNED = {"PBN_PRUNE_MARGIN": "52", "PBN_PRUNE_MARGIN_F32": "40"}


def run(env_extra, pinned=True):
    env = dict(os.environ)
    if pinned:
        env.update(PINNED)
    env.update(env_extra)
    p = subprocess.run([sys.executable, os.path.join(HERE, "switch_worker_gpu.py")], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0, p.stderr[-2000:]
    line = [ln for ln in p.stdout.splitlines() if ln.startswith("RESULT ")][-1]
    return json.loads(line[len("RESULT "):])


@pytest.fixture(scope="module")
def default():
    return run({})


def close(a, b, rtol):
    a, b = np.asarray(a, dtype=np.float64), np.asarray(b, dtype=np.float64)
    return np.allclose(a, b, rtol=rtol, atol=rtol, equal_nan=True)


def test_one_issue_lane_is_bit_identical(default):
    got = run({"PBN_SCORE_LANES": "1"})
    assert got == default


@pytest.mark.parametrize("env", [{"PBN_SWEEP_PRUNE": "0"}, {"PBN_SCORE_GROUPED": "0"}, {"PBN_GROUP_SPLIT_TILES": "64"}])
def test_sweep_switches(default, env):
    got = run(env)
    assert close(got["cv_ckde_float64"], default["cv_ckde_float64"], 1e-10)
    assert close(got["cv_ckde_float32"], default["cv_ckde_float32"], 1e-4)
    assert got["hc_arcs_float64"] == default["hc_arcs_float64"]
    assert got["hc_arcs_float32"] == default["hc_arcs_float32"]


def test_default_margins_stay_inside_their_bound(default):
    """The shipped margins (fp64 sum-only sweeps 43, fp32 36 at 10^6 rows: at most 1.1e-7 / 1.5e-5 of a sum dropped) against the
    pinned ones: same searches, scores within the bound."""
    got = run({}, pinned=False)
    assert close(got["cv_ckde_float64"], default["cv_ckde_float64"], 3e-7)
    assert close(got["cv_ckde_float32"], default["cv_ckde_float32"], 3e-5)
    assert close(got["hybrid_float64"], default["hybrid_float64"], 3e-7)
    assert got["hc_arcs_float64"] == default["hc_arcs_float64"] and got["hc_arcs_float32"] == default["hc_arcs_float32"]


def test_far_tiles_through_the_fp32_unit_stay_inside_their_bound(default):
    """PBN_FAR_SPAN=0: every visited tile through the full 2^f path instead of the fp32 tail for tiles 26+ bits below the sum bound
    (at most 8e-8 of a sum): same searches, scores within that bound."""
    got = run({"PBN_FAR_SPAN": "0"})
    assert close(got["cv_ckde_float64"], default["cv_ckde_float64"], 1e-7)
    assert close(got["hybrid_float64"], default["hybrid_float64"], 1e-7)
    assert got["hc_arcs_float64"] == default["hc_arcs_float64"]


@pytest.mark.parametrize("env", [{"PBN_MI_THREADS": "1"}, {"PBN_MI_FULL_BUDGET_MB": "0"}])
def test_mi_switches(default, env):
    got = run(env)
    for key in ("mi_plain", "mi_nulls"):
        assert close(got[key], default[key], 1e-9), key


def test_hybrid_candidates_batched_or_one_by_one_are_bit_identical(default):
    """PBN_HYBRID_BATCH=0 finishes every hybrid candidate before the next is prepared; the default enqueues the candidates of a
    pbn_score_batch call together (one grouped chain, one wait, terms shared inside the batch).  Every term is the same evaluation
    either way - and the same as when each candidate is asked for on its own."""
    got = run({"PBN_HYBRID_BATCH": "0"})
    for dtype in ("float64", "float32"):
        assert got[f"hybrid_batch_{dtype}"] == default[f"hybrid_batch_{dtype}"], dtype
        n = len(default[f"hybrid_{dtype}"])
        assert default[f"hybrid_batch_{dtype}"][:n] == default[f"hybrid_{dtype}"], dtype
        assert default[f"hybrid_batch_{dtype}"][0] == default[f"hybrid_batch_{dtype}"][n + 1], dtype   # c2 | {c1, d1} twice (parents in another order)


@pytest.mark.parametrize("env", [{"PBN_HYBRID_BATCH_SLOTS": "8"}, {"PBN_GROUP_ARENA_MB": "64"}, {"PBN_HYBRID_GROUPINGS": "1"}])
def test_hybrid_batch_cut_by_its_slots_or_its_arena_is_bit_identical(default, env):
    """A batch that runs out of result slots finishes what is in flight and starts over; one whose pools exceed the arena budget hands
    them over early.  Neither changes a bit of any score (the 64 MB arena also cuts the plain engine's chains).  PBN_HYBRID_GROUPINGS=1:
    the cache of row groupings (one per set of discrete parents, 256 by default) starts over at every new parent set - inside the batch,
    while earlier candidates' pools still read their grouping's device row list (the batch keeps those alive until it has flushed)."""
    got = run(env)
    for dtype in ("float64", "float32"):
        assert got[f"hybrid_batch_{dtype}"] == default[f"hybrid_batch_{dtype}"], dtype
        assert got[f"hybrid_{dtype}"] == default[f"hybrid_{dtype}"], dtype
        assert got[f"cv_ckde_{dtype}"] == default[f"cv_ckde_{dtype}"], dtype
