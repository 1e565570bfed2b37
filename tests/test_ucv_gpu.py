"""GPU tier: UCV bandwidth selector (kde/UCV.{hpp,cpp}).  The objective N * UCV(H) is compared with the oracle's
all-pairs restatement; the Nelder-Mead search (NLopt in the reference, absent here) is checked for what any correct
minimiser must deliver: a score no worse than the start, agreement with scipy's Nelder-Mead optimum, the reference's
guards, and a usable KDE."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def table(n, d, seed, dtype="float64"):
    rng = np.random.default_rng(seed)
    x = rng.normal(size=(n, d)) @ (np.eye(d) + np.tril(rng.uniform(-0.5, 0.5, (d, d)), -1)).T
    x[:, 0] = np.where(rng.random(n) < 0.4, x[:, 0] + 3.0, x[:, 0])       # bimodal: UCV moves away from the normal rule
    return pd.DataFrame(x.astype(dtype), columns=[f"v{i}" for i in range(d)])


@pytest.mark.parametrize("n,d", [(257, 1), (700, 2), (1203, 4), (500, 7), (611, 16), (903, 17), (700, 24)])   # d > 16: runtime-sized fp64 kernel
def test_ucv_score_matches_oracle(pbn, n, d):
    from oracle import oracle

    df = table(n, d, 3 + d)
    names = list(df.columns)
    ucv = pbn.UCV()
    H = pbn.NormalReferenceRule().bandwidth(df, names)
    h = pbn.NormalReferenceRule().diag_bandwidth(df, names)
    for bw in (H, 0.5 * H, 2.0 * H, h, 0.3 * h):
        got, want = ucv.score(df, names, bw), oracle.ucv_score(df.to_numpy(), bw)
        assert got == pytest.approx(want, rel=1e-9, abs=1e-14)
    df32 = df.astype("float32")
    assert ucv.score(df32, names, H) == pytest.approx(oracle.ucv_score(df.to_numpy(), H), rel=2e-4)
    with pytest.raises(ValueError, match="Wrong dimension"):
        ucv.score(df, names, np.eye(d + 1))


@pytest.mark.parametrize("d", [1, 2, 3])
def test_ucv_bandwidth_minimises(pbn, d):
    from scipy.optimize import minimize

    from oracle import oracle

    df = table(600, d, 40 + d)
    names = list(df.columns)
    ucv = pbn.UCV()
    nr = pbn.NormalReferenceRule()
    X = df.to_numpy()
    # diagonal
    h0 = nr.diag_bandwidth(df, names)
    h = ucv.diag_bandwidth(df, names)
    assert h.shape == (d,) and np.all(h > 0) and ucv.last_evaluations > d + 1
    s0, s1 = ucv.score(df, names, h0), ucv.score(df, names, h)
    assert s1 <= s0 + 1e-12
    ref = minimize(lambda v: oracle.ucv_score(X, v * v), np.sqrt(h0), method="Nelder-Mead", options={"xatol": 1e-5, "fatol": 1e-9})
    assert s1 == pytest.approx(ref.fun, rel=2e-3)
    assert np.allclose(np.sqrt(h), np.abs(ref.x), rtol=0.05)
    # unconstrained
    H0 = nr.bandwidth(df, names)
    H = ucv.bandwidth(df, names)
    assert H.shape == (d, d) and np.allclose(H, H.T) and np.all(np.linalg.eigvalsh(H) > 0)
    assert ucv.score(df, names, H) <= ucv.score(df, names, H0) + 1e-12
    assert 1e-3 <= np.linalg.det(H) / np.linalg.det(H0) <= 1e3              # the reference's determinant guard
    kde = pbn.KDE(names, pbn.UCV())
    kde.fit(df)
    assert np.allclose(kde.bandwidth, H) and np.isfinite(kde.slogl(df.iloc[:50]))
    assert str(pbn.UCV()) == "UCV" and pbn.UCV().bandwidth(df, []).shape == (0, 0)
