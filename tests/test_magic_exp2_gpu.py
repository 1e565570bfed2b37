"""GPU tier, fp64 sum-only sweeps: the 2^x that reads the accumulator's own words (kde_kernels.hip: exp2_magic) at the edges of what it
assumes.  The form needs the accumulator inside [2^20, 2^21) - |exponent| < 2^19 - and an exponent inside +-1023 for the bare
five-instruction form; a clamp (v_med3_i32 on the high word) makes it total, and the unpruned sweeps drop the clamp for chunks of training
tiles whose radii PROVE the exponents inside +-1022 (SweepArgs::tile_r).  These tables put rows and queries where those assumptions fail:
training rows thousands of bandwidths out (accumulators outside the binade: terms that must come out as 0), queries whose nearest row is
tens of bandwidths away (offsets far below 0: the guard's limit turns negative), both in one 64-query block with ordinary queries - and hold
the sum to the reference arithmetic (oracle/pbn_oracle.cpp after kde/opencl_kernels/KDE.cl.src:115-121,227-233) and to the per-row path
(the fp64 polynomial, an independent 2^x)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


def _fit(train, h2):
    import pybnesian_amd as pbn

    d = train.shape[1]
    names = [f"v{i}" for i in range(d)]
    k = pbn.KDE(names)
    k.fit(pd.DataFrame(train, columns=names))
    k.bandwidth = np.eye(d) * h2
    return k, names


@pytest.mark.parametrize("d", [3, 4, 5, 8, 10])      # norm in a K slot (KS = 1, 2, 3) and as weights (KS = 1, 2)
def test_rows_and_queries_beyond_the_guard(d, monkeypatch):
    from oracle import oracle

    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    rng = np.random.default_rng(100 + d)
    n, m, h = 6000, 640, 0.35
    train = rng.normal(size=(n, d))
    # outliers inside ordinary chunks: 40 / 400 / 4000 bandwidths out along random directions (weights that underflow, tile radii beyond every
    # limit, accumulators beyond the binade)
    for i, r in zip(rng.choice(n, 12, replace=False), np.repeat([40.0, 400.0, 4000.0, 30000.0], 3)):
        u = rng.normal(size=d)
        train[i] = u / np.linalg.norm(u) * r * h
    test = rng.normal(size=(m, d))
    # queries 12 ... 60 bandwidths from the bulk (their largest exponents lie hundreds of units below 0), a few next to the outliers, all mixed
    # into the 64-row blocks of ordinary queries
    far = rng.choice(m, 48, replace=False)
    for j, r in zip(far, np.tile([12.0, 25.0, 45.0, 60.0], 12)):
        u = rng.normal(size=d)
        test[j] = u / np.linalg.norm(u) * r * h * np.sqrt(d)
    near_out = rng.choice(np.setdiff1d(np.arange(m), far), 6, replace=False)
    test[near_out] = train[np.argsort(-np.linalg.norm(train, axis=1))[:6]] + rng.normal(scale=0.2 * h, size=(6, d))
    k, names = _fit(train, h * h)
    tdf = pd.DataFrame(test, columns=names)
    want_rows = oracle.kde_logl(train, np.eye(d) * h * h, test)
    rows = k.logl(tdf)                                 # the per-row path: fp64 polynomial
    s = k.slogl(tdf)                                   # the sum-only sweep: exp2_magic
    assert np.all(np.isfinite(want_rows)) and np.all(np.isfinite(rows)) and np.isfinite(s)
    # Gram-form distances: eps |z|^2 on an exponent (|z| up to 30 000 here for the queries next to the outliers)
    z2 = (np.vstack([test]) ** 2).sum(axis=1) / (h * h)
    tol = 1e-8 + 32.0 * 2.0 ** -52 * z2
    assert np.all(np.abs(rows - want_rows) <= tol * np.maximum(1.0, np.abs(want_rows))), np.max(np.abs(rows - want_rows))
    assert abs(s - rows.sum()) <= 2e-7 * m + float(np.sum(tol * np.maximum(1.0, np.abs(want_rows))))
    # ordinary queries alone (their blocks' chunks pass the guard except where an outlier row sits): 1e-8 of the sum
    keep = np.setdiff1d(np.arange(m), np.r_[far, near_out])
    s0 = k.slogl(tdf.iloc[keep])
    assert abs(s0 - want_rows[keep].sum()) <= 1e-8 * abs(want_rows[keep].sum())


def test_exponents_at_the_clamp(monkeypatch):
    """Training rows sorted along a line, 64 bandwidths long: the offsets follow the first tile of a split, so a split that starts far from a
    query and ends next to it walks through accumulators from below -1023 (the clamp's lower end: a subnormal) to beyond +1023 (its upper
    end: NaN / inf) - the blind chunks overflow and come back through the checked redo with raised offsets.  The sum must be the oracle's."""
    from oracle import oracle

    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    rng = np.random.default_rng(7)
    d, h = 4, 1.0
    n = 4096
    # training rows sorted along a line: tile t sits at distance ~ t / 4 bandwidths from the origin
    train = rng.normal(scale=0.05, size=(n, d))
    train[:, 0] += np.arange(n) / 64.0
    test = rng.normal(scale=0.05, size=(256, d))
    test[:, 0] += rng.uniform(0.0, n / 64.0, size=256)      # each query next to SOME tile, up to 64 bandwidths (x = -2950) from tile 0
    k, names = _fit(train, h * h)
    tdf = pd.DataFrame(test, columns=names)
    want = oracle.kde_logl(train, np.eye(d), test)
    s = k.slogl(tdf)
    assert np.isfinite(s) and abs(s - want.sum()) <= 1e-8 * abs(want.sum())
    assert np.allclose(k.logl(tdf), want, rtol=1e-9, atol=1e-9)


@pytest.mark.parametrize("d", [1, 2, 3])
def test_pruned_batches_beyond_the_proof(d):
    """Pruned sweeps (>= 32 768 training rows, d <= 3: the norm in a K slot, boxes over every dimension) drop the clamp per 64-tile batch when
    the LARGEST distance between the batch's box and the groups' boxes proves every exponent inside +-1022.  A bandwidth of 1/200 of the
    spread makes most of that fail - batch boxes hundreds of bandwidths wide, queries whose nearest row is dozens of bandwidths away
    (offsets below -890: the group's gate closes) - so proven and unproven batches, the checked loop and the overflow redo all meet in one
    sum, which must be the oracle's."""
    from oracle import oracle

    rng = np.random.default_rng(300 + d)
    n, m, h = 40_000, 1500, 0.005
    train = rng.normal(size=(n, d))
    train[:64] *= 40.0                                   # a sparse halo: tiles whose boxes span thousands of bandwidths
    test = rng.normal(size=(m, d))
    test[:96] = rng.normal(size=(96, d)) * 6.0           # queries in empty space
    k, names = _fit(train, h * h)
    tdf = pd.DataFrame(test, columns=names)
    want = oracle.kde_logl(train, np.eye(d) * h * h, test)
    s = k.slogl(tdf)
    assert np.isfinite(want).all() and np.isfinite(s)
    assert abs(s - want.sum()) <= 1e-8 * abs(want.sum())
    # the per-row path (fp64 polynomial, not this file's subject) as a second witness: Gram-form distances, eps |z|^2 on an exponent
    z2 = (test ** 2).sum(axis=1) / (h * h)
    tol = 1e-8 + 32.0 * 2.0 ** -52 * z2
    rows = k.logl(tdf)
    assert np.all(np.abs(rows - want) <= tol * np.maximum(1.0, np.abs(want))), np.max(np.abs(rows - want))


@pytest.mark.parametrize("n", [6000, 40_000])            # unpruned (tile-radius guard) and pruned (batch-box guard)
def test_null_rows_of_the_test_table(n):
    """The reference's slogl is the sum over the rows WITHOUT nulls (kde/KDE.hpp:451-478, 592-640: the test frame is filtered before the
    kernel), its logl carries NaN at the null rows: the mirror must do both whichever 2^x its sweeps take - the per-row path marks exactly
    that row, the sum-only path (exp2_magic) gives the sum of the others."""
    rng = np.random.default_rng(n)
    d = 2
    train = rng.normal(size=(n, d))
    test = rng.normal(size=(300, d))
    test[137, 1] = np.nan
    k, names = _fit(train, 0.04)
    tdf = pd.DataFrame(test, columns=names)
    rows = k.logl(tdf)
    assert np.isnan(rows[137]) and np.isfinite(np.delete(rows, 137)).all()
    s, clean = k.slogl(tdf), k.slogl(tdf.drop(index=137))
    assert np.isfinite(s) and abs(s - clean) <= 1e-10 * abs(clean)
    assert abs(s - np.delete(rows, 137).sum()) <= 1e-8 * abs(s)


def _cv_score(df, cols, k=3, seed=1):
    import pybnesian_amd as pbn

    s = pbn.CVLikelihood(df, k, seed)
    return s.local_score_node_type(pbn.SemiparametricBN(list(df.columns)), pbn.CKDEType(), cols[0], cols[1:])


def test_the_guards_change_no_bit(monkeypatch):
    """PBN_MAGIC_GUARD=0 keeps the clamp of exp2_magic everywhere; with the guards on, chunks / batches whose exponents are proven inside
    +-1022 run without it.  Inside that range the clamp is a no-op, so the two must agree BIT FOR BIT - unpruned sweeps (norms as weights and
    in a K slot, ordinary and heavy-tailed tables), pruned stand-alone handles and the grouped CV-likelihood terms of one to three variables."""
    rng = np.random.default_rng(77)

    def both(fn):
        monkeypatch.setenv("PBN_MAGIC_GUARD", "1")
        a = fn()
        monkeypatch.setenv("PBN_MAGIC_GUARD", "0")
        b = fn()
        monkeypatch.delenv("PBN_MAGIC_GUARD")
        return a, b

    for d, n in ((8, 20_000), (4, 20_000), (3, 20_000), (5, 20_000), (2, 60_000), (3, 60_000), (1, 40_000)):
        for heavy in (False, True):
            train = rng.standard_t(2.5, size=(n, d)) if heavy else rng.normal(size=(n, d))
            test = pd.DataFrame(rng.normal(size=(2000, d)) * (3.0 if heavy else 1.0), columns=[f"v{i}" for i in range(d)])

            def handle(train=train, test=test, d=d):
                import pybnesian_amd as pbn

                k = pbn.KDE(list(test.columns))
                k.fit(pd.DataFrame(train, columns=list(test.columns)))
                return k.slogl(test)

            a, b = both(handle)
            assert np.isfinite(a) and a == b, (d, n, heavy, a, b)
    a0 = rng.normal(size=45_000)
    df = pd.DataFrame({"a": a0, "b": np.tanh(a0) + 0.4 * rng.normal(size=45_000), "c": rng.standard_t(3, size=45_000)})
    for cols in (["a"], ["b", "a"], ["c", "a", "b"]):
        a, b = both(lambda cols=cols: _cv_score(df, cols))
        assert np.isfinite(a) and a == b, (cols, a, b)
