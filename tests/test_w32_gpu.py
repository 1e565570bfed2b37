"""GPU tier: the W32 form of the plain unpruned fp32 sweep (kde_sweep_f16_w32_kernel: v_mfma_f32_32x32x16_bf16 on the f16x2 fragments, the
stream placed by hand) against the f64 oracle on the f32-rounded data and against the 16x16 form it replaces (PBN_F32_W32=0, read per call).
Covers: every dimension the form applies to (5...9), full and diagonal bandwidths, training sizes with an odd number of 16-row tiles, fewer
tiles than one blind run, splits that end inside a chunk, and a table whose probe tiles all lie far from the queries (the blind chunks overflow
and are redone checked: the offset-raise path).  Replaces kde/opencl_kernels/KDE.cl.src:115-121,143-170 for float tables."""
import os

import numpy as np
import pandas as pd
import pytest

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


def _both(fn):
    """fn() under the W32 form and under the 16x16 form."""
    out = []
    for v in ("1", "0"):
        os.environ["PBN_F32_W32"] = v
        try:
            out.append(fn())
        finally:
            os.environ.pop("PBN_F32_W32", None)
    return out


@pytest.mark.parametrize("d", [5, 6, 7, 8, 9])
@pytest.mark.parametrize("n,m", [(17, 5), (1000, 33), (3010, 257), (70001, 1025)])
def test_w32_matches_the_oracle_and_the_16x16_form(pbn, oracle, d, n, m):
    rng = np.random.default_rng(1000 * d + n)
    mix = np.tril(rng.uniform(-0.5, 0.5, size=(d, d)), -1) + np.eye(d)
    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame((rng.normal(size=(n, d)) @ mix.T) * 2.0 + 5.0, columns=names).astype("float32")
    test = pd.DataFrame((rng.normal(size=(m, d)) @ mix.T) * 2.0 + 5.0, columns=names).astype("float32")
    for cls, fn in ((pbn.KDE, oracle.kde_logl), (pbn.ProductKDE, oracle.product_kde_logl)):
        k = cls(names)
        k.fit(train)
        want = fn(train.to_numpy().astype(np.float64), k.bandwidth, test.to_numpy().astype(np.float64))
        w32, w16 = _both(lambda: (k.logl(test), k.slogl(test)))
        assert np.allclose(w32[0], want, atol=5e-4, rtol=1e-4), "W32 against the f64 truth on the rounded data"
        assert np.allclose(w32[0], w16[0], atol=2e-4, rtol=2e-5), "W32 against the 16x16 form"
        assert abs(w32[1] - want.sum()) <= 1e-4 * abs(want.sum())
        assert abs(w32[1] - w16[1]) <= 2e-5 * abs(w16[1])


def test_w32_counts_the_sweeps_it_takes(pbn):
    """The W32 kernel is the one that runs for unpruned float tables whose contraction leaves its last three slots free (3 d + 6 <= 32 blocks,
    one or two blocks: d <= 8 and 10 ... 19) and is not for d = 9, 20, 24: the library's launch counter says which."""
    import ctypes as C

    from pybnesian_amd import _lib

    lib = _lib.load()
    rng = np.random.default_rng(5)
    for d, expect in ((4, 1), (8, 1), (9, 0), (10, 1), (19, 1), (20, 0), (24, 0)):
        names = [f"v{i}" for i in range(d)]
        train = pd.DataFrame(rng.normal(size=(5000, d)), columns=names).astype("float32")
        k = pbn.ProductKDE(names)
        k.fit(train)
        c = C.c_ulonglong(0)
        lib.pbn_debug_w32_launches(C.byref(c), 1)
        k.slogl(train.iloc[:500])
        lib.pbn_debug_w32_launches(C.byref(c), 0)
        assert (c.value > 0) == bool(expect), (d, c.value)


def test_w32_offsets_are_raised_when_every_probe_is_far(pbn, oracle):
    """4 000 training rows 15 bandwidths away from the queries and 24 rows among them (between two probe tiles) right beside the queries:
    the probes see only far rows, the near rows' exponents (~ +1 300) overflow against those offsets, the chunk is redone checked and the offsets
    rise.  (Inside the fp32-fragment criterion: |z|^2 ~ 2 600 < 4 194, PBN_F32_WIDEN_AT.)"""
    rng = np.random.default_rng(12)
    d = 8
    names = [f"v{i}" for i in range(d)]
    far = rng.normal(loc=3.0, scale=0.05, size=(4000, d))
    near = rng.normal(loc=0.0, scale=0.05, size=(24, d))
    rows = np.vstack([far[:1100], near, far[1100:]])
    train = pd.DataFrame(rows, columns=names).astype("float32")
    test = pd.DataFrame(rng.normal(loc=0.0, scale=0.05, size=(300, d)), columns=names).astype("float32")
    k = pbn.ProductKDE(names)
    k.fit(train)
    k.bandwidth = np.full(d, 0.04)
    import ctypes as C

    from pybnesian_amd import _lib

    c = C.c_ulonglong(0)
    _lib.load().pbn_debug_w32_launches(C.byref(c), 1)
    want = oracle.product_kde_logl(train.to_numpy().astype(np.float64), k.bandwidth, test.to_numpy().astype(np.float64))
    w32, w16 = _both(lambda: k.logl(test))
    _lib.load().pbn_debug_w32_launches(C.byref(c), 0)
    assert c.value == 1, "the model was not widened to fp64 fragments: the W32 form ran once"
    assert np.all(np.isfinite(w32))
    assert np.allclose(w32, want, atol=5e-4, rtol=1e-4)
    assert np.allclose(w32, w16, atol=2e-4, rtol=2e-5)
    # queries far from everything: the sums stay finite and equal to the oracle's
    lost = pd.DataFrame(rng.normal(loc=-3.0, scale=0.05, size=(40, d)), columns=names).astype("float32")
    want = oracle.product_kde_logl(train.to_numpy().astype(np.float64), k.bandwidth, lost.to_numpy().astype(np.float64))
    got = _both(lambda: k.logl(lost))[0]
    assert np.all(np.isfinite(got)) and np.allclose(got, want, rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize("cls", ["KDE", "ProductKDE", "CKDE"])
def test_queries_beyond_the_f16_range_are_evaluated_in_fp64(pbn, cls):
    """The f16x2 fragments clamp a whitened query coordinate beyond +-65504 (tens of thousands of bandwidths out); the pack flags such rows and
    kde_far_fix_kernel evaluates them in fp64 against the decoded training fragments: the result is the fp64 model's, to 1e-6 relative."""
    rng = np.random.default_rng(3)
    names = ["a", "b", "c"]
    train = pd.DataFrame(rng.normal(size=(5000, 3)), columns=names)
    q = rng.normal(size=(100, 3))
    q[7] = [9e4, 0.0, 0.0]; q[31] = [-2e5, 3e5, 1.0]; q[99] = [0.0, 0.0, -7e4]; q[50] = [40.0, -40.0, 40.0]
    test = pd.DataFrame(q, columns=names)
    def make():
        return pbn.CKDE("a", ["b", "c"]) if cls == "CKDE" else getattr(pbn, cls)(names)
    k32, k64 = make(), make()
    k32.fit(train.astype("float32"))
    k64.fit(train.astype("float32").astype("float64"))
    got, want = k32.logl(test.astype("float32")), k64.logl(test.astype("float32").astype("float64"))
    assert np.all(np.isfinite(got))
    far = [7, 31, 99]
    assert np.allclose(got[far], want[far], rtol=1e-6 if cls != "CKDE" else 1e-5)
    rest = [i for i in range(100) if i not in far]
    assert np.allclose(got[rest], want[rest], rtol=1e-4, atol=5e-4)


@pytest.mark.parametrize("d", [1, 2, 3, 5, 6])
def test_pruned_w32_form_matches_the_oracle_and_the_16x16_form(pbn, oracle, d):
    """Pruned fp32 handles (>= 32 768 training rows, <= 6 dimensions) and the grouped launches of the score engine take their kept tiles two at a
    time on 32x32x16 MFMAs (kde_sweep_f16_w32p_body); PBN_F32_W32=0 brings the 16x16 form back.  Clustered data with far queries: odd numbers of
    kept tiles per split, splits that are pruned whole, offsets from the prepass bounds."""
    import ctypes as C

    from pybnesian_amd import _lib

    rng = np.random.default_rng(40 + d)
    n, m = 50_021, 1501
    centres = rng.uniform(-20.0, 20.0, size=(3, d))
    mix = np.tril(rng.uniform(-0.4, 0.4, size=(d, d)), -1) + np.eye(d)

    def draw(k):
        return centres[rng.integers(0, 3, size=k)] + rng.normal(size=(k, d)) @ mix.T

    names = [f"v{i}" for i in range(d)]
    train = pd.DataFrame(draw(n), columns=names).astype("float32")
    q = draw(m)
    q[:7] += 300.0
    test = pd.DataFrame(q, columns=names).astype("float32")
    for cls, fn in ((pbn.KDE, oracle.kde_logl), (pbn.ProductKDE, oracle.product_kde_logl)):
        k = cls(names)
        k.fit(train)
        want = fn(train.to_numpy().astype(np.float64), k.bandwidth, test.to_numpy().astype(np.float64))
        c = C.c_ulonglong(0)
        _lib.load().pbn_debug_w32_launches(C.byref(c), 1)
        w32, w16 = _both(lambda: (k.logl(test), k.slogl(test)))
        _lib.load().pbn_debug_w32_launches(C.byref(c), 0)
        assert c.value >= 2, "the pruned W32 form ran (logl and slogl of the first pass)"
        assert np.all(np.isfinite(w32[0]))
        assert np.allclose(w32[0], want, atol=5e-4, rtol=1e-4)
        assert np.allclose(w32[0], w16[0], atol=2e-4, rtol=2e-5)
        assert abs(w32[1] - want.sum()) <= 1e-4 * abs(want.sum())


def test_grouped_fp32_terms_on_paired_tiles(pbn, oracle, monkeypatch):
    """The score engine's grouped fp32 launches (C5's path in miniature): CV-likelihood local scores of a float table through the paired-tile
    kernel against the 16x16 form and the oracle."""
    monkeypatch.setenv("PBN_PRUNE_MIN_ROWS", "256")
    rng = np.random.default_rng(9)
    n = 6000
    a = rng.normal(size=n)
    b = np.tanh(a) + 0.4 * rng.normal(size=n)
    c = 0.5 * a - 0.3 * b + 0.5 * rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c}).astype("float32")
    model = pbn.SemiparametricBN(list(df.columns))

    def scores():
        s = pbn.CVLikelihood(df, 4, 1)
        return [s.local_score_node_type(model, pbn.CKDEType(), v, ps) for v, ps in (("a", []), ("b", ["a"]), ("c", ["a", "b"]))]

    w32, w16 = _both(scores)
    data = df.to_numpy().astype(np.float64)
    wants = [oracle.cv_likelihood(data[:, cols], "ckde", 4, 1) for cols in ([0], [1, 0], [2, 0, 1])]
    for got32, got16, want in zip(w32, w16, wants):
        assert abs(got32 - want) <= 1e-4 * abs(want)
        assert abs(got32 - got16) <= 2e-5 * abs(want)
