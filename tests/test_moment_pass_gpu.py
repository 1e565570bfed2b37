"""GPU tier, fp64: the tile-moment pass of the grouped sum-only sweeps (kde_moment_group_kernel, round 5) - CV-likelihood CKDE terms of one
and two variables, where the pass takes the (tile, group) pairs whose order-8 expansion error is proved below the pruning bound and the sweep
keeps the rest.  What the numbers are held against is the reference's arithmetic (kde/opencl_kernels/KDE.cl.src:115-121,227-233 through
oracle/): the pass on, off, and the oracle must agree; the pass must really have taken pairs (debug counters), or the test proves nothing.
The shipped rule switches the pass on from 250 000 training rows (PBN_MOMENT_MIN_ROWS); the small cases lower it to 0."""
import ctypes as C

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _oracle():
    from oracle import oracle

    return oracle


def _pairs(lib, reset=False):
    p, b = C.c_ulonglong(0), C.c_ulonglong(0)
    lib.pbn_debug_moment_pairs(C.byref(p), C.byref(b), 1 if reset else 0)
    return p.value, b.value


def _score(df, cols, k, seed):
    import pybnesian_amd as pbn

    names = list(df.columns)
    s = pbn.CVLikelihood(df, k, seed)
    return s.local_score_node_type(pbn.SemiparametricBN(names), pbn.CKDEType(), cols[0], cols[1:])


def _tables(rng, n):
    a = rng.normal(size=n)
    b = np.tanh(a) + 0.4 * rng.normal(size=n)
    out = {"normal": np.column_stack([a, b])}
    t = rng.standard_t(2.5, size=n)
    out["heavy"] = np.column_stack([t, 0.5 * t + rng.standard_t(3, size=n)])              # far-out rows: offsets far below their neighbours'
    base = rng.normal(size=(n // 40, 2))
    out["dups"] = base[rng.integers(0, len(base), size=n)]                                 # tiles of identical rows: radius 0
    c = rng.uniform(-40, 40, size=(4, 2))
    out["clusters"] = c[rng.integers(0, 4, size=n)] + rng.normal(scale=0.5, size=(n, 2))   # empty space between dense regions
    out["offset"] = 1e4 + np.column_stack([a, 3.0 * b])
    return out


@pytest.mark.parametrize("kind", ["normal", "heavy", "dups", "clusters", "offset"])
def test_moment_pass_against_the_oracle_and_the_sweep(kind, monkeypatch):
    from pybnesian_amd import _lib

    oracle = _oracle()
    lib = _lib.load()
    rng = np.random.default_rng(21)
    n, k, seed = 9000, 3, 4
    x = _tables(rng, n)[kind]
    df = pd.DataFrame(x, columns=["a", "b"])
    monkeypatch.setenv("PBN_PRUNE_MIN_ROWS", "256")        # the grouped, pruned evaluation at this size
    monkeypatch.setenv("PBN_SWEEP_COUNT_REDO", "1")        # debug counters on
    for cols, ocols in ((["a"], [0]), (["b"], [1]), (["b", "a"], [1, 0]), (["a", "b"], [0, 1])):
        want = oracle.cv_likelihood(x[:, ocols], "ckde", k, seed)
        monkeypatch.setenv("PBN_MOMENT_MIN_ROWS", "0")
        _pairs(lib, reset=True)
        on = _score(df, cols, k, seed)
        taken, _ = _pairs(lib)
        monkeypatch.setenv("PBN_MOMENT_PASS", "0")
        _pairs(lib, reset=True)
        off = _score(df, cols, k, seed)
        assert _pairs(lib)[0] == 0
        monkeypatch.delenv("PBN_MOMENT_PASS")
        assert taken > 0, (kind, cols)
        assert abs(on - want) <= 1e-8 * abs(want), (kind, cols, on, want)
        assert abs(off - want) <= 1e-8 * abs(want), (kind, cols, off, want)
        assert abs(on - off) <= 3e-7 * abs(off)            # the documented budget of either form; measured 1e-9


def test_pinned_margin_the_two_passes_partition_the_pairs(monkeypatch):
    """With the pruning margin pinned at 52 nothing visible is dropped by either form and the expansion error bound is 2^-54 of a sum: the
    pass on and off then differ by rounding only - a pair taken twice, or by neither kernel, would show at 1e-5."""
    rng = np.random.default_rng(3)
    n = 60000
    a = rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": np.tanh(a) + 0.4 * rng.normal(size=n)})
    monkeypatch.setenv("PBN_PRUNE_MARGIN", "52")
    monkeypatch.setenv("PBN_MOMENT_MIN_ROWS", "0")
    for cols in (["a"], ["b", "a"]):
        on = _score(df, cols, 3, 1)
        monkeypatch.setenv("PBN_MOMENT_PASS", "0")
        off = _score(df, cols, 3, 1)
        monkeypatch.delenv("PBN_MOMENT_PASS")
        assert abs(on - off) <= 1e-10 * abs(off), (cols, on, off)


def test_shipped_rule_takes_dense_folds_only(monkeypatch):
    """Default knobs: folds of 450 000 training rows take the pass, folds of 125 000 do not; on and off agree inside the budget."""
    from pybnesian_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(8)
    monkeypatch.setenv("PBN_SWEEP_COUNT_REDO", "1")
    for n, expect in ((900_000, True), (250_000, False)):
        a = rng.normal(size=n)
        df = pd.DataFrame({"a": a, "b": 0.6 * a + rng.normal(scale=0.7, size=n)})
        _pairs(lib, reset=True)
        on = _score(df, ["b", "a"], 2, 0)
        assert (_pairs(lib)[0] > 0) == expect, n
        monkeypatch.setenv("PBN_MOMENT_PASS", "0")
        off = _score(df, ["b", "a"], 2, 0)
        monkeypatch.delenv("PBN_MOMENT_PASS")
        assert abs(on - off) <= 3e-7 * abs(off)


@pytest.mark.parametrize("moments", ["0", "1"])
def test_splits_longer_than_one_super_batch(moments, monkeypatch):
    """The two-level walk classifies 64 batches (4 096 tiles) per ballot.  A split of 9 375 tiles is three such super-batches, the last one partly
    filled: the same sums in another partition - the scores of the shipped split size and of ONE split per unit agree to rounding, with and
    without the moment pass, for one, two and three variables."""
    rng = np.random.default_rng(17)
    n = 300_000
    a = rng.normal(size=n)
    b = np.tanh(a) + 0.4 * rng.normal(size=n)
    c = 0.5 * a - 0.3 * b + rng.normal(scale=0.8, size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c})
    monkeypatch.setenv("PBN_MOMENT_PASS", moments)
    monkeypatch.setenv("PBN_MOMENT_MIN_ROWS", "0")
    for cols in (["a"], ["b", "a"], ["c", "a", "b"]):
        base = _score(df, cols, 2, 1)
        monkeypatch.setenv("PBN_GROUP_SPLIT_TILES", "16384")
        one = _score(df, cols, 2, 1)
        monkeypatch.setenv("PBN_GROUP_SPLIT_TILES", "5000")    # two splits, the second one 4 375 tiles: a second super-batch with 5 batches
        two = _score(df, cols, 2, 1)
        monkeypatch.delenv("PBN_GROUP_SPLIT_TILES")
        assert abs(one - base) <= 1e-11 * abs(base), (cols, one, base)
        assert abs(two - base) <= 1e-11 * abs(base), (cols, two, base)
