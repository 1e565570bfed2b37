"""GPU tier: CKDE (fused joint+marginal sweep) against the golden scipy recipes of
/root/reference/tests/factors/continuous/CKDE_test.py and against the CPU oracle."""
import numpy as np
import pandas as pd
import pytest

from helpers import CKDE_SETS, COLS, RTOL_F32, RTOL_F64, frame, rel_err

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


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
@pytest.mark.parametrize("tag", ["10k", "10"])
def test_ckde_golden_f64(pbn, golden, variable, evidence, tag):
    key = variable + "_" + "".join(evidence)
    train = frame(golden["train10k" if tag == "10k" else "train10"])
    test = frame(golden["test50"])
    cpd = pbn.CKDE(variable, evidence)
    cpd.fit(train)
    assert np.allclose(cpd.bandwidth, golden[f"ckde_bw_{key}_{tag}"], rtol=1e-8)
    want = golden[f"ckde_logl_{key}_{tag}"]
    got = cpd.logl(test)
    assert np.allclose(got, want, rtol=RTOL_F64, atol=1e-9)
    assert abs(cpd.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    assert cpd.num_instances() == train.shape[0]
    assert cpd.variable() == variable and cpd.evidence() == evidence


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_ckde_golden_f32(pbn, golden, variable, evidence):
    key = variable + "_" + "".join(evidence)
    cpd = pbn.CKDE(variable, evidence)
    cpd.fit(frame(golden["train10k"], "float32"))
    test = frame(golden["test50"], "float32")
    want = golden[f"ckde_logl_{key}_10k"]
    got = cpd.logl(test)
    assert np.allclose(got, want, atol=5e-4)  # CKDE_test.py:231-232
    assert abs(cpd.slogl(test) - want.sum()) <= 0.0005 * 10000  # CKDE_test.py:325-327


def test_ckde_evidence_order_and_nulls(pbn, golden):
    train, test = frame(golden["train10k"]), frame(golden["test50"])
    c1, c2 = pbn.CKDE("d", ["a", "b", "c"]), pbn.CKDE("d", ["c", "b", "a"])
    c1.fit(train)
    c2.fit(train)
    assert np.allclose(c1.logl(test), c2.logl(test), rtol=1e-7)
    tn = frame(golden["test50_null"])
    l1, l2 = c1.logl(tn), c2.logl(tn)
    nulls = np.any(np.isnan(golden["test50_null"]), axis=1)
    assert np.array_equal(np.isnan(l1), nulls) and np.array_equal(np.isnan(l2), nulls)
    assert np.allclose(l1[~nulls], l2[~nulls], rtol=1e-7)
    assert abs(c1.slogl(tn) - np.nansum(l1)) <= 1e-9 * abs(np.nansum(l1))


def test_ckde_dtype_mismatch(pbn, golden):
    cpd = pbn.CKDE("a", ["b"])
    with pytest.raises(ValueError, match="not fitted"):
        cpd.slogl(frame(golden["test50"]))
    cpd.fit(frame(golden["train500"]))
    with pytest.raises(ValueError, match="Data type of training and test datasets is different."):
        cpd.logl(frame(golden["test50"], "float32"))


@pytest.mark.parametrize("p", [1, 2, 3, 4, 5, 8, 10, 12, 14, 16])
def test_ckde_oracle_parity_random(pbn, oracle, p):
    """Seeded non-linear tables, ragged sizes; includes conditional outliers (joint far, marginal near)."""
    rng = np.random.default_rng(40 + p)
    n, m = 4099, 301
    ev = rng.normal(size=(n + m, p)) @ (np.tril(rng.uniform(-0.4, 0.4, size=(p, p)), -1) + np.eye(p)).T
    y = np.tanh(ev[:, 0]) * 2.0 + 0.3 * ev.sum(axis=1) + rng.normal(scale=0.4, size=n + m)
    data = np.column_stack([y, ev])
    names = ["y"] + [f"e{i}" for i in range(p)]
    train = pd.DataFrame(data[:n], columns=names)
    test = pd.DataFrame(data[n:], columns=names)
    test.iloc[:5, 0] += 40.0  # conditional outliers: evidence typical, variable far away
    cpd = pbn.CKDE("y", names[1:])
    cpd.fit(train)
    want = oracle.ckde_logl(train.to_numpy(), cpd.bandwidth, test.to_numpy())
    got = cpd.logl(test)
    assert np.all(np.isfinite(got))
    assert np.allclose(got, want, rtol=RTOL_F64, atol=1e-9)
    assert abs(cpd.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())


@pytest.mark.parametrize("p", [2, 6, 11])
def test_ckde_oracle_parity_random_f32(pbn, oracle, p):
    """fp32 CKDE on the bf16 matrix cores (f16x2 split), incl. NB = 1..3 MFMAs per tile: reference tolerances for
    float data (atol 5e-4 per value, CKDE_test.py:231-232) against the fp64 oracle on the same rounded data."""
    rng = np.random.default_rng(70 + p)
    n, m = 3001, 203
    ev = rng.normal(size=(n + m, p)) @ (np.tril(rng.uniform(-0.3, 0.3, size=(p, p)), -1) + np.eye(p)).T
    y = np.tanh(ev[:, 0]) + 0.2 * ev.sum(axis=1) + rng.normal(scale=0.5, size=n + m)
    data = np.column_stack([y, ev]).astype(np.float32)
    names = ["y"] + [f"e{i}" for i in range(p)]
    train = pd.DataFrame(data[:n], columns=names)
    test = pd.DataFrame(data[n:], columns=names)
    cpd = pbn.CKDE("y", names[1:])
    cpd.fit(train)
    want = oracle.ckde_logl(data[:n].astype(np.float64), cpd.bandwidth, data[n:].astype(np.float64))
    got = cpd.logl(test)
    assert np.allclose(got, want, atol=5e-4, rtol=1e-4)
    assert abs(cpd.slogl(test) - want.sum()) <= RTOL_F32 * abs(want.sum())


# ---- CKDE.cdf (CKDE_test.py:256-314) ---------------------------------------------------------------------------
@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
@pytest.mark.parametrize("tag", ["10k", "10"])
def test_ckde_cdf_golden(pbn, golden, variable, evidence, tag):
    key = variable + "_" + "".join(evidence)
    want = golden[f"ckde_cdf_{key}_{tag}"]
    base = golden["train10k" if tag == "10k" else "train10"]
    cpd = pbn.CKDE(variable, evidence)
    cpd.fit(frame(base))
    got = cpd.cdf(frame(golden["test50"]))
    assert np.allclose(got, want, rtol=1e-8, atol=1e-12)           # reference: np.isclose defaults
    cpd32 = pbn.CKDE(variable, evidence)
    cpd32.fit(frame(base, "float32"))
    assert np.allclose(cpd32.cdf(frame(golden["test50"], "float32")), want, atol=5e-4)  # CKDE_test.py:268-271


def test_ckde_cdf_nulls_and_order(pbn, golden):
    train = frame(golden["train10k"])
    c1, c2 = pbn.CKDE("d", ["a", "b", "c"]), pbn.CKDE("d", ["c", "b", "a"])
    c1.fit(train)
    c2.fit(train)
    tn = frame(golden["test50_null"])
    r1, r2 = c1.cdf(tn), c2.cdf(tn)
    nulls = np.any(np.isnan(golden["test50_null"]), axis=1)
    assert np.array_equal(np.isnan(r1), nulls)
    assert np.allclose(r1[~nulls], r2[~nulls], rtol=1e-9)
    full = c1.cdf(frame(golden["test50"]))
    assert np.allclose(r1[~nulls], full[~nulls], rtol=1e-12)
    kde = pbn.KDE(["a", "b"])
    kde.fit(train)
    assert not hasattr(kde, "cdf")


@pytest.mark.parametrize("p", [0, 1, 3, 4, 5, 9, 13, 16, 17, 20, 33])
def test_ckde_cdf_oracle_parity_random(pbn, oracle, p):
    """Ragged sizes, KS = 1..4 and - beyond 16 evidence variables, where the reference's cdf loops on (CKDE.hpp:509-735) - the
    runtime-sized fp64 kernels (p = 17, 20, 33); evidence outliers (weights underflow in the reference's exp(logl) form only when ALL
    of them do; here the offset keeps the ratio defined) and variable outliers (cdf -> 0 / 1)."""
    rng = np.random.default_rng(90 + p)
    n, m = 1237, 77
    ev = rng.normal(size=(n + m, p)) @ (np.tril(rng.uniform(-0.4, 0.4, size=(p, p)), -1) + np.eye(p)).T
    y = 0.5 * ev.sum(axis=1) + rng.normal(scale=0.7, size=n + m) + (np.sin(ev[:, 0]) if p else 0.0)
    data = np.column_stack([y, ev])
    names = ["y"] + [f"e{i}" for i in range(p)]
    train = pd.DataFrame(data[:n], columns=names)
    test = pd.DataFrame(data[n:], columns=names)
    test.iloc[:3, 0] += 25.0
    test.iloc[3:6, 0] -= 25.0
    cpd = pbn.CKDE("y", names[1:])
    cpd.fit(train)
    want = oracle.ckde_cdf(train.to_numpy(), cpd.bandwidth, test.to_numpy())
    got = cpd.cdf(test)
    assert np.all((got >= 0) & (got <= 1))
    assert np.allclose(got, want, rtol=1e-8, atol=1e-13)
    assert np.all(got[:3] > 1 - 1e-9) and np.all(got[3:6] < 1e-9)
    # monotone in the variable for fixed evidence
    t2 = test.copy()
    t2["y"] = t2["y"] + 0.25
    assert np.all(cpd.cdf(t2) >= got - 1e-14)


def test_ckde_cdf_far_evidence(pbn):
    """Evidence 60 bandwidths away: every weight underflows exp(); the reference returns 0/0 = NaN there, the
    offset form returns the cdf of the nearest kernels.  Check it is finite and equals the log-domain answer."""
    rng = np.random.default_rng(5)
    n = 500
    e = rng.normal(size=n)
    y = e + rng.normal(scale=0.5, size=n)
    train = pd.DataFrame({"y": y, "e": e})
    cpd = pbn.CKDE("y", ["e"])
    cpd.fit(train)
    test = pd.DataFrame({"y": [30.0, 29.0, 31.5], "e": [30.0, 30.0, 30.0]})
    got = cpd.cdf(test)
    H = cpd.bandwidth
    b = H[0, 1] / H[1, 1]
    sd = np.sqrt(H[0, 0] - H[0, 1] ** 2 / H[1, 1])
    from scipy.special import logsumexp
    from scipy.stats import norm

    lw = -0.5 * (30.0 - e) ** 2 / H[1, 1]
    want = [np.exp(logsumexp(lw + norm.logcdf(v, y + b * (30.0 - e), sd)) - logsumexp(lw)) for v in test["y"]]
    assert np.all(np.isfinite(got))
    assert np.allclose(got, want, rtol=1e-8)


def test_new_entry_points_edge_cases(pbn, golden):
    """Empty and tiny inputs through cdf / sample / UCV / independence tests."""
    train = frame(golden["train500"])
    cpd = pbn.CKDE("a", ["b"])
    cpd.fit(train)
    empty = train.iloc[:0]
    assert cpd.cdf(empty).shape == (0,) and cpd.logl(empty).shape == (0,) and cpd.slogl(empty) == 0.0
    assert len(cpd.sample(0, empty[["b"]], 0)) == 0
    one = cpd.cdf(train.iloc[:1])
    assert one.shape == (1,) and 0 <= one[0] <= 1
    tiny = train.iloc[:3]
    c2 = pbn.CKDE("a", [])
    c2.fit(tiny)
    s = c2.sample(7, None, 1).to_numpy()
    assert s.shape == (7,) and np.all(np.isfinite(s))
    assert np.isfinite(pbn.UCV().score(tiny, ["a"], np.array([[0.5]])))
    with pytest.raises(ValueError):
        pbn.UCV().score(train.iloc[:1], ["a"], np.array([[0.5]]))
    lc = pbn.LinearCorrelation(train.iloc[:4])
    assert 0 <= lc.pvalue("a", "b") <= 1
    with pytest.raises(ValueError):                        # 4 rows, 2 conditioning variables: no degrees of freedom left
        lc.pvalue("a", "b", ["c", "d"])


@pytest.mark.parametrize("n", [20_000, 120_000])   # below / above the row count from which fitted handles prune
def test_far_queries_fp32_stay_finite(pbn, n):
    """Queries ~10^6 bandwidths away from fp32 training data with a large offset: the exponents (~1e13) are far beyond
    what fp32 resolves, the result must still be the finite, hugely negative value of the fp64 evaluation (relative 1e-5),
    never -inf (tools/fuzz_pruned.py, seed 1 case 65)."""
    rng = np.random.default_rng(65)
    names = ["y", "e"]
    train = pd.DataFrame(1e4 + rng.normal(size=(n, 2)) * np.array([1.0, 2.0]), columns=names).astype("float32")
    near = 1e4 + rng.normal(size=(4097, 2)) * np.array([1.0, 2.0])
    # (a near variable with far evidence is left out: lj - lm of two ~1e12 values is not resolvable in fp32 anywhere)
    far = np.array([[5.01e5, 5.01e5], [5.0e5, 5.02e5], [-3e5, 1e4], [2e6, -2e6], [6e4, 6e4]])
    test = pd.DataFrame(np.vstack([near, far]), columns=names).astype("float32")
    cpd = pbn.CKDE("y", ["e"])
    cpd.fit(train)
    got = cpd.logl(test)
    ref = pbn.CKDE("y", ["e"])
    ref.fit(train.astype("float64"))
    want = ref.logl(test.astype("float64"))
    assert np.all(np.isfinite(got))
    assert np.allclose(got[-5:], want[-5:], rtol=1e-5, atol=5e-3)
    assert np.allclose(got[:-5], want[:-5], rtol=1e-4, atol=5e-4)
    kde = pbn.KDE(names)
    kde.fit(train)
    ref = pbn.KDE(names)
    ref.fit(train.astype("float64"))
    got, want = kde.logl(test), ref.logl(test.astype("float64"))
    assert np.all(np.isfinite(got)) and np.allclose(got[-5:], want[-5:], rtol=1e-5, atol=5e-3)
