"""GPU tier: the run-time switches that select between implementations of the same quantity must not change it.  Each
configuration runs tests/switch_worker_gpu.py in its own process (the switches are read once per process) and the outputs are
compared with the default run:
 * PBN_SCORE_LANES=1 (one issue lane) - the sums do not depend on the issue order: identical to the last bit;
 * PBN_SWEEP_QLB=0 (offsets from the split's first tile instead of the prepass bounds) and PBN_SWEEP_PRUNE=0 (no tile
   pruning) - other offsets / other partitions of the same sums: equal to rounding, the fp32 sweeps to their own precision;
 * PBN_SCORE_GROUPED=0 (one launch chain per (set, fold) as in round 2 instead of the grouped evaluation of kde_group.hip),
   PBN_PRUNE_GROUP_MASKS=0 (one visit mask per wave instead of one per 16-query group), PBN_GROUP_SUM_BOUND=0 (pruning
   threshold on the largest known term instead of the known part of the sum), PBN_GROUP_SPLIT_TILES / PBN_GROUP_MAX_POOLS
   (other partitions of the same work);
 * PBN_MI_FULLGRAM=0 (per-test moment kernels instead of the per-grouping moments), PBN_MI_FULL_BUDGET_MB=0 (their cache
   budget exhausted: the same fallback), PBN_MI_THREADS=1;
 * PBN_GRAM_LDS=1 / 0 (the older Gram kernels), PBN_MI_GRAM_ORDER=0 / 1 (launch order of a grouping's Gram pieces);
 * PBN_HYBRID_FULLMOMENTS=0, PBN_HYBRID_FUSED=1, PBN_HYBRID_SEGMENTED=0, PBN_SCORE_MEMO=0 (hybrid candidates); PBN_HYBRID_BATCH=0
   (the hybrid candidates of a batch one by one instead of in one chain)."""
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


@pytest.mark.parametrize("env", [{"PBN_SWEEP_QLB": "0"}, {"PBN_SWEEP_PRUNE": "0"}, {"PBN_SCORE_GROUPED": "0"}, {"PBN_PRUNE_GROUP_MASKS": "0"},
                                 {"PBN_GROUP_SUM_BOUND": "0"}, {"PBN_GROUP_SPLIT_TILES": "64"}, {"PBN_GROUP_MAX_POOLS": "1"}])
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


@pytest.mark.parametrize("env", [{"PBN_MI_FULLGRAM": "0"}, {"PBN_MI_THREADS": "1"}, {"PBN_MI_FULL_BUDGET_MB": "0"}])
def test_mi_switches(default, env):
    got = run(env)
    for key in ("mi_plain", "mi_nulls"):
        assert close(got[key], default[key], 1e-9), key


@pytest.mark.parametrize("env", [{"PBN_HYBRID_FULLMOMENTS": "0"}, {"PBN_HYBRID_FUSED": "1"}, {"PBN_HYBRID_SEGMENTED": "0"}, {"PBN_SCORE_MEMO": "0"},
                                 {"PBN_HYBRID_FULLMOMENTS": "0", "PBN_HYBRID_SEGMENTED": "0"},
                                 {"PBN_HYBRID_FULLMOMENTS": "0", "PBN_HYBRID_SEGMENTED": "0", "PBN_HYBRID_CELLWISE_GRAM": "1"}])
def test_hybrid_score_switches(default, env):
    """Hybrid candidates: moments from the per-grouping Gram or from per-candidate launches (the register kernel for up to 8 columns;
    without it - as for wider candidates - the candidate's columns through one segmented MFMA Gram for all cells, or one launch per cell
    as in round 3), slices fused or split, with and without the local-score memo - the same scores to rounding (the fp32 tables to
    their own precision)."""
    got = run(env)
    # (PBN_HYBRID_FUSED=1 evaluates a slice by the fused joint + marginal kernel - fp64 polynomial - instead of two plain sum-only
    #  sweeps - 2^f on the fp32 transcendental unit: 1e-8, both far inside the 1e-6 bar)
    assert close(got["hybrid_float64"], default["hybrid_float64"], 1e-8 if env == {"PBN_HYBRID_FUSED": "1"} else 1e-10)
    assert close(got["hybrid_float32"], default["hybrid_float32"], 1e-4)
    if env == {"PBN_SCORE_MEMO": "0"}:
        assert got["hybrid_float64"] == default["hybrid_float64"]      # the memo returns what a fresh evaluation gives


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


@pytest.mark.parametrize("order", ["0", "1"])
def test_grouping_gram_launch_order_is_bit_identical(default, order):
    """PBN_MI_GRAM_ORDER: the pieces of a grouping's Gram in configuration-major / stripe-major order instead of the XCD-aligned
    stripe-major default - the partial slots and the order of every sum are the same, so are the bits."""
    got = run({"PBN_MI_GRAM_ORDER": order})
    assert got["mi_plain"] == default["mi_plain"] and got["mi_nulls"] == default["mi_nulls"]


@pytest.mark.parametrize("variant", ["1", "0"])
def test_gram_kernel_variants(default, variant):
    got = run({"PBN_GRAM_LDS": variant})
    assert close(got["bic"], default["bic"], 1e-11)
    assert close(got["mi_plain"], default["mi_plain"], 1e-7)   # p-values of sums taken in another order: rounding times the statistic
