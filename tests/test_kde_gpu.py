"""GPU tier: KDE / ProductKDE through the Python mirror -> C ABI -> HIP kernels, against
(1) the golden scipy recipes of the reference tests, (2) the CPU oracle on seeded inputs,
(3) size-independent properties at larger sizes.  Tolerances: BASELINE.json north_star —
1e-6 relative for fp64, 1e-3 for fp32 (the reference tests themselves use atol 5e-4 per value for fp32)."""
import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

from helpers import COLS, RTOL_F32, RTOL_F64, VARSETS, frame, rel_err

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_bandwidth_golden(pbn, golden, variables):
    key = "".join(variables)
    df = frame(golden["train500"])
    for n in (50, 500):
        k = pbn.KDE(variables)
        k.fit(df.iloc[:n])
        assert np.allclose(k.bandwidth, golden[f"kde_bw_nr_{key}_{n}"], rtol=1e-8)
        k = pbn.KDE(variables, pbn.ScottsBandwidth())
        k.fit(df.iloc[:n])
        assert np.allclose(k.bandwidth, golden[f"kde_bw_scott_{key}_{n}"], rtol=1e-8)
        k.fit(df.iloc[:n].astype("float32"))
        assert np.allclose(k.bandwidth, golden[f"kde_bw_scott_{key}_{n}"], rtol=1e-4)
    for n in (50, 150, 500):
        k = pbn.ProductKDE(variables)
        k.fit(df.iloc[:n])
        assert np.allclose(k.bandwidth, golden[f"pkde_bw_nr_{key}_{n}"], rtol=1e-8)
        k = pbn.ProductKDE(variables, pbn.ScottsBandwidth())
        k.fit(df.iloc[:n])
        assert np.allclose(k.bandwidth, golden[f"pkde_bw_scott_{key}_{n}"], rtol=1e-8)


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_logl_slogl_golden_f64(pbn, golden, variables):
    key = "".join(variables)
    k = pbn.KDE(variables)
    k.fit(frame(golden["train500"]))
    test = frame(golden["test50"])
    want = golden[f"kde_logl_{key}_f64"]
    got = k.logl(test)
    assert got.shape == want.shape
    assert rel_err(got, want) < RTOL_F64
    assert abs(k.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    assert k.num_instances() == 500 and k.num_variables() == len(variables) and k.fitted()
    assert k.data_type() == pa.float64()


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_logl_golden_f32(pbn, golden, variables):
    key = "".join(variables)
    k = pbn.KDE(variables)
    k.fit(frame(golden["train500"], "float32"))
    test = frame(golden["test50"], "float32")
    want = golden[f"kde_logl_{key}_f32"]
    got = k.logl(test)
    assert np.allclose(got, want, atol=5e-4)  # KDE_test.py:181-182
    assert abs(k.slogl(test) - want.sum()) <= RTOL_F32 * abs(want.sum())
    assert k.data_type() == pa.float32()


@pytest.mark.parametrize("variables", VARSETS)
def test_product_kde_golden(pbn, golden, variables):
    key = "".join(variables)
    k = pbn.ProductKDE(variables)
    k.fit(frame(golden["train500"]))
    test = frame(golden["test50"])
    want = golden[f"pkde_logl_{key}_f64"]
    assert rel_err(k.logl(test), want) < RTOL_F64
    assert abs(k.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    k.fit(frame(golden["train500"], "float32"))
    got32 = k.logl(frame(golden["test50"], "float32"))
    assert np.allclose(got32, want, atol=5e-3)  # ProductKDE_test.py:222-223


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_logl_null_rows(pbn, golden, variables):
    key = "".join(variables)
    k = pbn.KDE(variables)
    k.fit(frame(golden["train500"]))
    test = frame(golden["test50_null"])
    want = golden[f"kde_logl_null_{key}_f64"]
    got = k.logl(test)
    assert np.array_equal(np.isnan(got), np.isnan(want))
    ok = ~np.isnan(want)
    assert rel_err(got[ok], want[ok]) < RTOL_F64
    assert abs(k.slogl(test) - np.nansum(want)) <= RTOL_F64 * abs(np.nansum(want))


def test_fit_with_null_rows(pbn, golden, oracle):
    df = frame(golden["train500"]).copy()
    rng = np.random.RandomState(0)
    for c in COLS:
        df.loc[df.index[rng.randint(0, 500, size=30)], c] = np.nan
    variables = ["c", "a", "b"]
    k = pbn.KDE(variables)
    k.fit(df)
    clean = df[variables].dropna()
    assert k.num_instances() == clean.shape[0]
    cov, _ = oracle.cov(clean.to_numpy())
    H = oracle.bandwidth(0, 0, cov, clean.shape[0])
    assert np.allclose(k.bandwidth, H, rtol=1e-9)
    test = frame(golden["test50"])
    want = oracle.kde_logl(clean.to_numpy(), H, test[variables].to_numpy())
    assert rel_err(k.logl(test), want) < RTOL_F64


def test_dtype_mismatch_and_unfitted(pbn, golden):
    df, dff = frame(golden["train500"]), frame(golden["train500"], "float32")
    k = pbn.KDE(["a"])
    with pytest.raises(ValueError, match="not fitted"):
        k.slogl(df)
    k.fit(df)
    for fn in (k.logl, k.slogl):
        with pytest.raises(ValueError, match="Data type of training and test datasets is different."):
            fn(dff)
    k.fit(dff)
    for fn in (k.logl, k.slogl):
        with pytest.raises(ValueError, match="Data type of training and test datasets is different."):
            fn(df)
    with pytest.raises(ValueError):
        pbn.KDE([])


def test_singular_covariance(pbn, golden):
    df = frame(golden["train500"])
    with pytest.raises(pbn.SingularCovarianceData):
        pbn.KDE(["a", "b", "c"]).fit(df.iloc[:3])  # N <= d
    dup = df.copy()
    dup["b"] = 2.0 * dup["a"]
    with pytest.raises(pbn.SingularCovarianceData):
        pbn.KDE(["a", "b"]).fit(dup)  # not positive definite
    assert issubclass(pbn.SingularCovarianceData, ValueError)


def test_variable_order_invariance(pbn, golden):
    df, test = frame(golden["train500"]), frame(golden["test50"])
    k1, k2 = pbn.KDE(["d", "a", "b", "c"]), pbn.KDE(["a", "c", "d", "b"])
    k1.fit(df)
    k2.fit(df)
    assert np.allclose(k1.logl(test), k2.logl(test), rtol=1e-7)


def test_set_bandwidth_refits(pbn, golden, oracle):
    df, test = frame(golden["train500"]), frame(golden["test50"])
    k = pbn.KDE(["a", "b"])
    k.fit(df)
    H = np.array([[0.5, 0.1], [0.1, 2.0]])
    k.bandwidth = H
    want = oracle.kde_logl(df[["a", "b"]].to_numpy(), H, test[["a", "b"]].to_numpy())
    assert rel_err(k.logl(test), want) < RTOL_F64
    with pytest.raises(ValueError):
        k.bandwidth = np.eye(3)


def test_custom_selector(pbn, golden, oracle):
    class Fixed(pbn.BandwidthSelector):
        def bandwidth(self, df, variables):
            return np.eye(len(variables)) * 0.3

        def diag_bandwidth(self, df, variables):
            return np.full(len(variables), 0.3)

    df, test = frame(golden["train500"]), frame(golden["test50"])
    v = ["a", "c"]
    k = pbn.KDE(v, Fixed())
    k.fit(df)
    want = oracle.kde_logl(df[v].to_numpy(), np.eye(2) * 0.3, test[v].to_numpy())
    assert rel_err(k.logl(test), want) < RTOL_F64
    p = pbn.ProductKDE(v, Fixed())
    p.fit(df)
    assert rel_err(p.logl(test), want) < RTOL_F64


@pytest.mark.parametrize("dtype,rtol", [("float64", RTOL_F64), ("float32", RTOL_F32)])
@pytest.mark.parametrize("d", [1, 2, 3, 4, 5, 6, 7, 8, 9, 11, 12, 13, 15, 16])   # every KS with and without a free K slot
def test_oracle_parity_random(pbn, oracle, d, dtype, rtol):
    """Seeded correlated Gaussian tables; ragged sizes (not multiples of 16) on both sides.
    float64: against the oracle, 1e-6 relative.  float32 is judged TWICE and the names of the bars say which is which:
    (a) `f64 truth`: the f64 oracle on the f32-ROUNDED data - what the exact answer for these inputs is - at the reference tests'
    own f32 tolerance (atol 5e-4, rtol 1e-4 per logl): the product is MORE accurate than the reference's f32 arithmetic, so this is
    the tight bar; (b) `f32 restatement`: the oracle run in f32 like the reference (itself only ~1e-4 accurate), at the north star's
    fp32 bar (1e-3 relative on slogl) - the bar a drop-in replacement of the reference's f32 path has to meet."""
    rng = np.random.default_rng(100 + d)
    n, m = 3001, 257
    mix = np.tril(rng.uniform(-0.5, 0.5, size=(d, d)), -1) + np.eye(d)
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame((rng.normal(size=(n, d)) @ mix.T) * 3.0 + 10.0, columns=names).astype(dtype)
    test = pd.DataFrame((rng.normal(size=(m, d)) @ mix.T) * 3.0 + 10.0, columns=names).astype(dtype)
    for cls, fn in ((pbn.KDE, oracle.kde_logl), (pbn.ProductKDE, oracle.product_kde_logl)):
        k = cls(names)
        k.fit(train)
        # the oracle works in the data dtype like the reference; its f32 path is itself only ~1e-4 accurate,
        # so the f32 comparison uses the f64 oracle on the f32-rounded data as the truth
        want = fn(train.to_numpy().astype(np.float64), k.bandwidth, test.to_numpy().astype(np.float64))
        got = k.logl(test)
        if dtype == "float64":
            assert rel_err(got, want) < rtol
        else:
            assert np.allclose(got, want, atol=5e-4, rtol=1e-4), "f32 against the f64 truth on the rounded data"
            want32 = np.asarray(fn(train.to_numpy(), k.bandwidth, test.to_numpy()), dtype=np.float64)   # the reference's own arithmetic type
            assert abs(k.slogl(test) - want32.sum()) <= 1e-3 * abs(want32.sum()), "f32 against the f32 restatement"
        assert abs(k.slogl(test) - want.sum()) <= rtol * abs(want.sum())


def test_far_queries_trigger_rescale(pbn, oracle):
    """Queries tens of bandwidths away from every training point, and a training set whose first tile is
    far from the queries' neighbourhood: exercises the offset (m) raise path and the no-underflow logic."""
    rng = np.random.default_rng(7)
    far = rng.normal(loc=500.0, scale=0.01, size=(16, 2))     # first tile: very far cluster
    near = rng.normal(loc=0.0, scale=1.0, size=(5000, 2))
    train = pd.DataFrame(np.vstack([far, near]), columns=["x", "y"])
    q = np.vstack([rng.normal(size=(40, 2)), rng.normal(loc=60.0, size=(10, 2)), rng.normal(loc=-300.0, size=(7, 2))])
    test = pd.DataFrame(q, columns=["x", "y"])
    H = np.array([[0.05, 0.01], [0.01, 0.08]])
    k = pbn.KDE(["x", "y"])
    k.fit(train)
    k.bandwidth = H
    want = oracle.kde_logl(train.to_numpy(), H, test.to_numpy())
    got = k.logl(test)
    assert np.all(np.isfinite(got))
    assert rel_err(got, want) < RTOL_F64


def test_tiny_sizes(pbn, oracle):
    rng = np.random.default_rng(3)
    train = pd.DataFrame(rng.normal(size=(3, 1)), columns=["x"])
    test = pd.DataFrame(rng.normal(size=(1, 1)), columns=["x"])
    k = pbn.KDE(["x"])
    k.fit(train)
    want = oracle.kde_logl(train.to_numpy(), k.bandwidth, test.to_numpy())
    assert rel_err(k.logl(test), want) < RTOL_F64
    empty = test.iloc[:0]
    assert k.logl(empty).shape == (0,)
    assert k.slogl(empty) == 0.0


def test_large_properties(pbn):
    """Size-independent properties at a size the oracle cannot finish quickly (2e5 x 2e4, d=8):
    slogl == sum(logl); splitting the test rows is additive; duplicating the training set leaves logl
    unchanged when the bandwidth is held fixed; permuting training rows changes nothing beyond rounding."""
    rng = np.random.default_rng(11)
    d, n, m = 8, 200_000, 20_000
    names = [f"v{i}" for i in range(d)]
    mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
    train = pd.DataFrame(rng.normal(size=(n, d)) @ mix.T, columns=names)
    test = pd.DataFrame(rng.normal(size=(m, d)) @ mix.T, columns=names)
    k = pbn.ProductKDE(names)
    k.fit(train)
    h = k.bandwidth.copy()
    ll = k.logl(test)
    s = k.slogl(test)
    assert np.all(np.isfinite(ll))
    assert abs(s - ll.sum()) <= 1e-9 * abs(s)
    s1, s2 = k.slogl(test.iloc[:7777]), k.slogl(test.iloc[7777:])
    assert abs((s1 + s2) - s) <= 1e-9 * abs(s)
    perm = rng.permutation(n)
    k2 = pbn.ProductKDE(names)
    k2.fit(train.iloc[perm])
    k2.bandwidth = h
    assert np.allclose(k2.logl(test.iloc[:2000]), ll[:2000], rtol=1e-9, atol=1e-9)
    k3 = pbn.ProductKDE(names)
    k3.fit(pd.concat([train, train], ignore_index=True))
    k3.bandwidth = h
    assert np.allclose(k3.logl(test.iloc[:2000]), ll[:2000], rtol=1e-9, atol=1e-9)


def test_pickle_roundtrip(pbn, golden):
    """tests/serialization/serialize_factor_test.py: a pickled fitted KDE / ProductKDE / CKDE scores identically."""
    import pickle

    df, test = frame(golden["train500"]), frame(golden["test50"])
    for obj in (pbn.KDE(["a", "c"]), pbn.ProductKDE(["b", "d", "a"]), pbn.CKDE("d", ["a", "b"]), pbn.KDE(["a"], pbn.ScottsBandwidth())):
        unfitted = pickle.loads(pickle.dumps(obj))
        assert not unfitted.fitted()
        obj.fit(df)
        clone = pickle.loads(pickle.dumps(obj))
        assert clone.fitted() and clone.num_instances() == 500
        assert np.array_equal(clone.bandwidth, obj.bandwidth)
        assert np.allclose(clone.logl(test), obj.logl(test), rtol=1e-12, atol=1e-12)
    f32 = pbn.KDE(["a", "b"])
    f32.fit(frame(golden["train500"], "float32"))
    c32 = pickle.loads(pickle.dumps(f32))
    assert c32.data_type() == pa.float32()
    assert np.allclose(c32.logl(frame(golden["test50"], "float32")), f32.logl(frame(golden["test50"], "float32")), rtol=1e-6)


def test_pickle_state_is_the_reference_tuple(pbn, golden):
    """The on-disk format of this path: KDE::__getstate__ (kde/KDE.hpp:642-666) =
    (variables, fitted, selector, bandwidth, flat column-major training vector, lognorm_const, N, arrow type id);
    ProductKDE the same with a bandwidth vector and a list of columns (ProductKDE.hpp:310-334); CKDE =
    (variable, evidence, fitted, joint KDE tuple) (CKDE.hpp:737-745); unfitted models carry the reference's placeholders."""
    df = frame(golden["train500"])
    k = pbn.KDE(["a", "c"])
    st = k.__getstate__()
    assert len(st) == 8 and st[0] == ["a", "c"] and st[1] is False and st[3].shape == (0, 0) and len(st[4]) == 0
    assert st[5:] == (-1.0, -1, -1)
    k.fit(df)
    st = k.__getstate__()
    assert len(st) == 8 and st[1] is True and isinstance(st[2], pbn.NormalReferenceRule)
    assert st[3].shape == (2, 2) and np.array_equal(st[3], k.bandwidth)
    assert st[4].dtype == np.float64 and np.array_equal(st[4], df[["a", "c"]].to_numpy().reshape(-1, order="F"))
    d, n = 2, 500
    want_lognorm = -0.5 * np.linalg.slogdet(k.bandwidth)[1] - 0.5 * d * np.log(2 * np.pi) - np.log(n)   # KDE.hpp:476-477
    assert abs(st[5] - want_lognorm) <= 1e-12 * abs(want_lognorm) and st[6] == 500 and st[7] == 12            # arrow::Type::DOUBLE
    k32 = pbn.KDE(["a", "b"])
    k32.fit(frame(golden["train500"], "float32"))
    s32 = k32.__getstate__()
    assert s32[7] == 11 and s32[4].dtype == np.float32
    pk = pbn.ProductKDE(["b", "d", "a"])
    assert pk.__getstate__()[3].shape == (0,) and pk.__getstate__()[4] == []
    pk.fit(df)
    sp = pk.__getstate__()
    assert len(sp) == 8 and sp[3].shape == (3,) and isinstance(sp[4], list) and len(sp[4]) == 3
    assert all(np.array_equal(c, df[v].to_numpy()) for c, v in zip(sp[4], ["b", "d", "a"]))
    want = -0.5 * 3 * np.log(2 * np.pi) - 0.5 * np.log(pk.bandwidth).sum() - np.log(500)                     # ProductKDE.hpp:188-189
    assert abs(sp[5] - want) <= 1e-12 * abs(want)
    c = pbn.CKDE("d", ["a", "b"])
    assert c.__getstate__() == ("d", ["a", "b"], False, ())
    c.fit(df)
    sc = c.__getstate__()
    assert len(sc) == 4 and sc[:3] == ("d", ["a", "b"], True) and len(sc[3]) == 8
    assert sc[3][0] == ["d", "a", "b"] and sc[3][3].shape == (3, 3) and sc[3][6] == 500
    assert np.array_equal(sc[3][4], df[["d", "a", "b"]].to_numpy().reshape(-1, order="F"))
    with pytest.raises(RuntimeError, match="Not valid KDE"):
        pbn.KDE(["a"]).__setstate__((["a"], False))


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("d", [1, 2, 3, 5])
def test_pruned_handles_match_oracle(pbn, oracle, d, dtype, monkeypatch):
    """Fitted KDE / ProductKDE / CKDE handles with <= 5 (marginal) dimensions and >= 32768 training rows pack their rows in
    Morton order and skip the tile pairs that cannot contribute: logl comes back in the caller's row
    order, equal to the oracle and to the unpruned sweep, on clustered data with far-away queries."""
    rng = np.random.default_rng(900 + d)
    n, m = 40_003, 1501
    centres = rng.uniform(-30.0, 30.0, size=(4, d))
    mix = np.tril(rng.uniform(-0.4, 0.4, size=(d, d)), -1) + np.eye(d)
    def draw(k):
        return centres[rng.integers(0, 4, size=k)] + rng.normal(size=(k, d)) @ mix.T
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame(draw(n), columns=names).astype(dtype)
    q = draw(m)
    q[:9] += 400.0                                   # queries far from every training row
    q[9:20] = 0.5 * (centres[0] + centres[1])        # between clusters
    test = pd.DataFrame(q, columns=names).astype(dtype)
    t64, q64 = train.to_numpy().astype(np.float64), test.to_numpy().astype(np.float64)
    cases = [(lambda: pbn.KDE(names), oracle.kde_logl), (lambda: pbn.ProductKDE(names), oracle.product_kde_logl)]
    if d >= 2:
        cases.append((lambda: pbn.CKDE(names[0], names[1:]), oracle.ckde_logl))
    for make, fn in cases:
        k = make()
        k.fit(train)
        got, s = k.logl(test), k.slogl(test)
        want = fn(t64, k.bandwidth, q64)
        assert np.all(np.isfinite(got))
        if dtype == "float64":
            assert rel_err(got, want) < RTOL_F64
            assert abs(s - want.sum()) <= RTOL_F64 * abs(want.sum())
        else:
            assert np.allclose(got, want, atol=5e-4, rtol=1e-4)
            assert abs(s - want.sum()) <= RTOL_F32 * abs(want.sum())
        # slogl: a sum-only sweep (2^f on the fp32 unit, pruning margin 43: 1.4e-7 + 1.1e-7 of a sum at most) against the per-row logl
        # (polynomial, margin 52)
        assert abs(s - got.sum()) <= (3e-7 if dtype == "float64" else 1e-5) * abs(s)
        monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
        plain = make()
        plain.fit(train)
        monkeypatch.delenv("PBN_SWEEP_PRUNE")
        ref = plain.logl(test)
        assert np.allclose(got, ref, rtol=1e-9 if dtype == "float64" else 1e-4, atol=1e-9 if dtype == "float64" else 5e-4)
        assert np.allclose(k.logl(test.iloc[100:777]), got[100:777], rtol=1e-9 if dtype == "float64" else 1e-4,
                           atol=1e-9 if dtype == "float64" else 5e-4)
        assert np.allclose(k.logl(test.iloc[5:6]), got[5:6], rtol=1e-9 if dtype == "float64" else 1e-4, atol=5e-4)   # one far query alone
        assert k.logl(test.iloc[:0]).shape == (0,) and k.slogl(test.iloc[:0]) == 0.0


@pytest.mark.parametrize("what,d", [("kde", 2), ("ckde", 3), ("ckde", 4)])
def test_whole_splits_pruned_far_cluster(pbn, oracle, monkeypatch, what, d):
    """Two clusters ~1000 bandwidths apart, queries next to one of them: every tile of the other cluster - whole training
    splits of the sweep - is pruned, so their partial sums are the empty-sum rule's single term (kde_sweep_kernel epilogue).
    fp64, plain pruned sweep (KDE d = 2; CKDE d = 3 as two plain sweeps) and the fused pruned CKDE sweep (d = 4): equal to
    the unpruned sweep and to the oracle."""
    rng = np.random.default_rng(17)
    n = 48_000
    mix = np.tril(np.full((d, d), 0.3), -1) + np.eye(d)
    a = rng.normal(size=(n, d)) @ mix.T
    b = rng.normal(size=(n, d)) @ mix.T + 300.0
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame(np.vstack([a, b]), columns=names)
    test = pd.DataFrame(np.vstack([rng.normal(size=(700, d)) @ mix.T, rng.normal(size=(300, d)) @ mix.T + 300.0]), columns=names)
    make = (lambda: pbn.KDE(names)) if what == "kde" else (lambda: pbn.CKDE(names[0], names[1:]))
    k = make()
    k.fit(train)
    got = k.logl(test)
    assert np.all(np.isfinite(got))
    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    plain = make()
    plain.fit(train)
    monkeypatch.delenv("PBN_SWEEP_PRUNE")
    # (Gram-form distances around a centre between the clusters: eps |z|^2 ~ 1e-10 on every exponent, in both sweeps)
    assert np.allclose(got, plain.logl(test), rtol=1e-8, atol=1e-8)
    sub = test.iloc[np.r_[0:40, 700:740]]
    fn = oracle.kde_logl if what == "kde" else oracle.ckde_logl
    want = fn(train.to_numpy(), k.bandwidth, sub.to_numpy())
    assert np.allclose(k.logl(sub), want, rtol=1e-7, atol=1e-7)


@pytest.mark.parametrize("d", [4, 8])
@pytest.mark.parametrize("cls", ["KDE", "ProductKDE"])
def test_weighted_norm_sweep_with_outliers(pbn, oracle, d, cls):
    """d mod 4 = 0: the sweep takes the training norms as weights 2^(-|z|^2/2) and accumulates blind over chunks of 32 tiles
    (kde_sweep_kernel WMUL / unchecked passes).  Heavy-tailed rows (|z|^2 far beyond 2000: NaN weights by design), queries next
    to them (2^x overflows) and queries far from everything (offsets hundreds of bandwidths off) must all come through the
    checked redo with the oracle's values."""
    rng = np.random.default_rng(100 + d)
    n, m = 6000, 400
    names = [f"v{i}" for i in range(d)]
    train = rng.standard_t(1.5, size=(n, d))
    train[:40] *= 50.0                                   # a block of extreme rows in the first tiles
    test = np.vstack([rng.standard_t(1.5, size=(m, d)), train[:8] + 0.01, train[-8:] * 1.001, rng.normal(size=(8, d)) * 1e4])
    tdf, qdf = pd.DataFrame(train, columns=names), pd.DataFrame(test, columns=names)
    k = getattr(pbn, cls)(names)
    k.fit(tdf)
    got = k.logl(qdf)
    fn = oracle.kde_logl if cls == "KDE" else oracle.product_kde_logl
    want = fn(train, k.bandwidth, test)
    assert np.all(np.isfinite(got)) == np.all(np.isfinite(want))
    fin = np.isfinite(want)
    # Gram-form distances around the pilot centre: eps |z|^2 on an exponent, |z|^2 up to ~1e9 for the far queries here
    H = np.asarray(k.bandwidth, dtype=np.float64)
    X = np.vstack([train, test]) - train[:1024].mean(axis=0)
    Z2 = ((X * X) / H).sum(axis=1) if H.ndim == 1 else (np.linalg.solve(np.linalg.cholesky(H), X.T) ** 2).sum(axis=0)
    tol = 1e-8 + 16.0 * 2.0 ** -52 * Z2[n:]
    assert np.all(np.abs(got[fin] - want[fin]) <= tol[fin] * np.maximum(1.0, np.abs(want[fin]))), np.max(np.abs(got[fin] - want[fin]))
    assert abs(k.slogl(qdf.iloc[:m]) - want[:m].sum()) <= 1e-7 * abs(want[:m].sum())


@pytest.mark.parametrize("d", [1, 2, 3])
def test_unchecked_passes_with_stale_offsets(pbn, oracle, monkeypatch, d):
    """Unpruned fp64 sweeps over a SORTED training table: the offsets a split takes from its first tile are thousands of
    bandwidths off for most queries, so the blind passes overflow all the time - including the nasty case of a lane's
    partial sum that is finite but so large that adding the four lanes of a query column overflows - and every such chunk
    must come back through the checked redo: finite everywhere, equal to the pruned handle and to the oracle."""
    rng = np.random.default_rng(40 + d)
    n, m = 150_000, 3000
    names = [f"v{i}" for i in range(d)]
    x = rng.normal(size=(n, d))
    x = x[np.argsort(x[:, 0])]
    train = pd.DataFrame(x, columns=names)
    test = pd.DataFrame(rng.normal(size=(m, d)) * 1.3, columns=names)
    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    plain = pbn.KDE(names)
    plain.fit(train)
    got = plain.logl(test)
    s = plain.slogl(test)
    monkeypatch.delenv("PBN_SWEEP_PRUNE")
    # slogl (a sum: 2^f on the fp32 transcendental unit, <= 1.4e-7 per term) against the sum of the per-row logl (fp64 polynomial,
    # 2.2e-9): 1e-8 relative - the north star's bar is 1e-6
    assert np.all(np.isfinite(got)) and np.isfinite(s) and abs(s - got.sum()) <= 1e-8 * abs(s)
    pruned = pbn.KDE(names)
    pruned.fit(train)
    assert np.allclose(pruned.logl(test), got, rtol=1e-10, atol=1e-10)
    sub = np.r_[0:60, m - 60:m]
    want = oracle.kde_logl(x, plain.bandwidth, test.to_numpy()[sub])
    assert np.allclose(got[sub], want, rtol=1e-8, atol=1e-8)


def test_pruned_handles_full_size_properties(pbn, monkeypatch):
    """BASELINE sizes (1e6 training x 1e5 test rows) at d = 2, where the fitted handle prunes: the sum equals the unpruned
    sweep's and the sum of the per-row values, and splitting the test rows is additive (each slice is Morton-ordered on its
    own, so this also checks that the scatter back to caller order does not depend on the slice)."""
    rng = np.random.default_rng(5)
    n, m, names = 1_000_000, 100_000, ["a", "b"]
    mix = np.array([[1.0, 0.0], [0.3, 1.0]])
    train = pa.RecordBatch.from_arrays([pa.array(c) for c in (rng.normal(size=(n, 2)) @ mix.T).T], names=names)
    q = rng.normal(size=(m, 2)) @ mix.T
    test = pd.DataFrame(q, columns=names)
    k = pbn.KDE(names)
    k.fit(train)
    ll, s = k.logl(test), k.slogl(test)
    assert np.all(np.isfinite(ll)) and abs(s - ll.sum()) <= 1e-8 * abs(s)     # sum-only sweeps take the fp32-unit 2^f, per-row logl the polynomial
    s1, s2 = k.slogl(test.iloc[:33_333]), k.slogl(test.iloc[33_333:])
    assert abs((s1 + s2) - s) <= 1e-10 * abs(s)
    assert np.allclose(k.logl(test.iloc[50_000:50_100]), ll[50_000:50_100], rtol=1e-11, atol=1e-11)
    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    plain = pbn.KDE(names)
    plain.fit(train)
    monkeypatch.delenv("PBN_SWEEP_PRUNE")
    assert abs(plain.slogl(test) - s) <= 3e-7 * abs(s)       # sum-only sweeps prune at the margin whose bound is 1.1e-7 of a sum (prune_margin)
    monkeypatch.setenv("PBN_PRUNE_MARGIN", "52")             # ... and pinned at 52 the pruned sum is the unpruned one to rounding
    assert abs(k.slogl(test) - plain.slogl(test)) <= 1e-10 * abs(s)
    monkeypatch.delenv("PBN_PRUNE_MARGIN")
    assert np.allclose(plain.logl(test.iloc[:4096]), ll[:4096], rtol=1e-10, atol=1e-10)


@pytest.mark.parametrize("d", [17, 20, 24, 29, 32])   # KS = 5 ... 8: the whitening matrix travels through device memory
def test_more_than_16_dimensions(pbn, oracle, d):
    """KDE / ProductKDE / CKDE over 17-32 variables in fp64 (the reference has no limit: KDE.hpp is dimension-agnostic) against the
    oracle, and in fp32 against the f64 oracle on the rounded data; the cdf of such a CKDE against the oracle, too."""
    rng = np.random.default_rng(500 + d)
    n, m = 2501, 133
    mix = np.tril(rng.uniform(-0.3, 0.3, size=(d, d)), -1) + np.eye(d)
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame((rng.normal(size=(n, d)) @ mix.T) * 2.0 - 3.0, columns=names)
    test = pd.DataFrame((rng.normal(size=(m, d)) @ mix.T) * 2.0 - 3.0, columns=names)
    for cls, fn in ((pbn.KDE, oracle.kde_logl), (pbn.ProductKDE, oracle.product_kde_logl)):
        k = cls(names)
        k.fit(train)
        want = fn(train.to_numpy(), k.bandwidth, test.to_numpy())
        assert rel_err(k.logl(test), want) < RTOL_F64, cls.__name__
        assert abs(k.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    cpd = pbn.CKDE(names[0], names[1:])
    cpd.fit(train)
    want = oracle.ckde_logl(train.to_numpy(), cpd.bandwidth, test.to_numpy())
    assert rel_err(cpd.logl(test), want) < RTOL_F64
    if d - 1 > 16:   # the cdf beyond 16 evidence variables: runtime-sized fp64 kernels (round 4)
        assert np.allclose(cpd.cdf(test), oracle.ckde_cdf(train.to_numpy(), cpd.bandwidth, test.to_numpy()), rtol=1e-8, atol=1e-13)
    # float32 tables: the f16x2 sweep with 4-7 MFMAs per tile pair; the f64 oracle on the f32-rounded data is the truth
    tr32, te32 = train.astype("float32"), test.astype("float32")
    k32 = pbn.KDE(names)
    k32.fit(tr32)
    want32 = oracle.kde_logl(tr32.to_numpy().astype(np.float64), k32.bandwidth, te32.to_numpy().astype(np.float64))
    assert np.allclose(k32.logl(te32), want32, atol=5e-4, rtol=1e-4)
    c32 = pbn.CKDE(names[0], names[1:])
    c32.fit(tr32)
    wantc = oracle.ckde_logl(tr32.to_numpy().astype(np.float64), c32.bandwidth, te32.to_numpy().astype(np.float64))
    assert np.allclose(c32.logl(te32), wantc, atol=2e-3, rtol=1e-3)


@pytest.mark.parametrize("d", [33, 40, 64, 80])
def test_more_than_32_dimensions(pbn, oracle, d):
    """The reference's KDE kernels loop over any number of variables (kde/KDE.hpp:592-640, KDE.cl.src:123-135); beyond the templated
    shapes (32 whitened dimensions) the library takes a generic runtime-sized pack + sweep in fp64 fragments, a CKDE of that size is
    evaluated as joint - marginal (CKDE.hpp:256-287), and the covariance of more than 64 columns is assembled from 32-column block
    pairs.  KDE / ProductKDE / CKDE in fp64 against the oracle; fp32 tables (packed into doubles) against the fp64 oracle on the
    rounded data; ragged sizes, several training splits."""
    rng = np.random.default_rng(900 + d)
    n, m = 3001, 77
    mix = np.tril(rng.uniform(-0.2, 0.2, size=(d, d)), -1) + np.eye(d)
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame((rng.normal(size=(n, d)) @ mix.T) * 1.5 + 2.0, columns=names)
    test = pd.DataFrame((rng.normal(size=(m, d)) @ mix.T) * 1.5 + 2.0, columns=names)
    for cls, fn in ((pbn.KDE, oracle.kde_logl), (pbn.ProductKDE, oracle.product_kde_logl)):
        k = cls(names)
        k.fit(train)
        cov, _ = oracle.cov(train.to_numpy())
        want_bw = oracle.bandwidth(0, 0 if cls is pbn.KDE else 1, cov, n)
        assert np.allclose(k.bandwidth, want_bw, rtol=1e-8), cls.__name__          # incl. the block-pair covariance (d > 64)
        want = fn(train.to_numpy(), k.bandwidth, test.to_numpy())
        assert rel_err(k.logl(test), want) < RTOL_F64, cls.__name__
        assert abs(k.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    cpd = pbn.CKDE(names[0], names[1:])           # d - 1 >= 32 evidence variables
    cpd.fit(train)
    want = oracle.ckde_logl(train.to_numpy(), cpd.bandwidth, test.to_numpy())
    got = cpd.logl(test)
    assert np.allclose(got, want, rtol=RTOL_F64, atol=1e-7)
    assert abs(cpd.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    tr32, te32 = train.astype("float32"), test.astype("float32")
    k32 = pbn.KDE(names)
    k32.fit(tr32)
    want32 = oracle.kde_logl(tr32.to_numpy().astype(np.float64), k32.bandwidth, te32.to_numpy().astype(np.float64))
    assert np.allclose(k32.logl(te32), want32, atol=5e-4, rtol=1e-4)
    # pickled and restored, the wide model evaluates the same
    import pickle

    k2 = pickle.loads(pickle.dumps(k))
    assert np.array_equal(k2.logl(test), k.logl(test))


def test_fp32_tables_with_tiny_bandwidths_take_fp64_fragments(pbn, monkeypatch):
    """fp32 tables whose whitened rows lie hundreds of bandwidths from the centre (diagonal bandwidths of nearly collinear columns;
    a user-set bandwidth far below the spread): the fp32 Gram form z_t.z_q - |z_t|^2/2 - |z_q|^2/2 loses ~2^-24 |z|^2 on every
    exponent - 1e-3...1e-2 on a logl - where the reference's fp32 path subtracts BEFORE squaring (KDE.cl.src:173-226) and holds
    1e-5.  Such models are packed into fp64 fragments and swept by the fp64 kernels (KdeModel::widen, decided at fit time from the
    farthest whitened training row).  Checked against the oracle in fp64 arithmetic on the same float data at the reference
    tests' fp32 tolerance (atol 5e-4 per logl, KDE_test.py:181-182), with the oracle's own fp32 arithmetic beside it; the switch
    PBN_F32_WIDEN_AT=inf shows what the fp32 fragments would have given."""
    from oracle import oracle

    rng = np.random.default_rng(77)
    n, m, d = 40_000, 600, 3
    def draw(k):
        t = rng.normal(size=(k, 1))
        return (t @ np.ones((1, d)) + rng.normal(scale=0.02, size=(k, d))).astype(np.float32)
    names = [f"v{i}" for i in range(d)]
    train, test = pd.DataFrame(draw(n), columns=names), pd.DataFrame(draw(m), columns=names)
    tr64, te64 = train.to_numpy().astype(np.float64), test.to_numpy().astype(np.float64)

    def models():
        a = pbn.ProductKDE(names)                  # the diagonal rule on collinear columns: bandwidths ~1e-5 of the variances
        a.fit(train)
        b = pbn.KDE(names)                         # a user-set bandwidth far below the spread of the data
        b.fit(train)
        b.bandwidth = np.eye(d) * 4e-5
        c = pbn.CKDE(names[0], names[1:])
        c.fit(train)
        Hc = np.eye(d) * 4e-5 + 1e-5
        c.kde_joint().bandwidth = Hc               # the reference's way of setting a CKDE's bandwidth: through its member KDEs
        c.kde_marg().bandwidth = Hc[1:, 1:]
        return [(a, oracle.product_kde_logl), (b, oracle.kde_logl), (c, oracle.ckde_logl)]

    worst_widened = 0.0
    for k, fn in models():
        bw = np.asarray(k.bandwidth, dtype=np.float64)
        truth = fn(tr64, bw, te64)
        ref32 = fn(train.to_numpy(), bw, test.to_numpy())          # the reference's fp32 arithmetic (differences first)
        got = k.logl(test)
        fin = np.isfinite(truth)
        assert np.abs(ref32[fin] - truth[fin]).max() <= 5e-4 * np.maximum(1.0, np.abs(truth[fin])).max()
        err = np.abs(got[fin] - truth[fin]).max()
        worst_widened = max(worst_widened, err)
        assert err <= 5e-4, (type(k).__name__, err)
        assert abs(k.slogl(test) - truth[fin].sum()) <= 1e-6 * abs(truth[fin].sum()) or not fin.all()
    # the same models on fp32 fragments: the error this routing removes (otherwise the test above proves nothing)
    monkeypatch.setenv("PBN_F32_WIDEN_AT", "inf")
    worst_f32 = 0.0
    for k, fn in models():
        truth = fn(tr64, np.asarray(k.bandwidth, dtype=np.float64), te64)
        fin = np.isfinite(truth)
        worst_f32 = max(worst_f32, np.abs(k.logl(test)[fin] - truth[fin]).max())
    assert worst_f32 > 10 * max(worst_widened, 1e-6), (worst_f32, worst_widened)
    monkeypatch.delenv("PBN_F32_WIDEN_AT")
    # ordinary fp32 tables stay on the fp32 (bf16x3) path: same numbers with the test switched off
    g = pd.DataFrame(rng.normal(size=(20_000, 3)).astype(np.float32), columns=names)
    q = pd.DataFrame(rng.normal(size=(300, 3)).astype(np.float32), columns=names)
    k1 = pbn.KDE(names); k1.fit(g); l1 = k1.logl(q)
    monkeypatch.setenv("PBN_F32_WIDEN_AT", "inf")
    k2 = pbn.KDE(names); k2.fit(g)
    assert np.array_equal(l1, k2.logl(q))
