"""CPU tier: pin the oracle (oracle/pbn_oracle.cpp) against the reference tests' own scipy/numpy recipes
(tests/golden/reference_recipes.npz, see gen_golden.py for the cited recipes)."""
import numpy as np
import pytest

from helpers import CKDE_SETS, COLS, VARSETS, frame
from oracle import oracle


def _sel(arr, variables):
    return np.asarray(arr)[:, [COLS.index(v) for v in variables]]


@pytest.mark.parametrize("variables", VARSETS)
def test_bandwidths(golden, variables):
    key = "".join(variables)
    for n in (50, 500):
        data = _sel(golden["train500"], variables)[:n]
        cov, _ = oracle.cov(data)
        assert np.allclose(oracle.bandwidth(0, 0, cov, n), golden[f"kde_bw_nr_{key}_{n}"], rtol=1e-9)
        assert np.allclose(oracle.bandwidth(1, 0, cov, n), golden[f"kde_bw_scott_{key}_{n}"], rtol=1e-9)
    for n in (50, 150, 500):
        data = _sel(golden["train500"], variables)[:n]
        cov, _ = oracle.cov(data)
        assert np.allclose(oracle.bandwidth(0, 1, cov, n), golden[f"pkde_bw_nr_{key}_{n}"], rtol=1e-9)
        assert np.allclose(oracle.bandwidth(1, 1, cov, n), golden[f"pkde_bw_scott_{key}_{n}"], rtol=1e-9)


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_logl_f64(golden, variables):
    key = "".join(variables)
    tr, te = _sel(golden["train500"], variables), _sel(golden["test50"], variables)
    H = golden[f"kde_bw_nr_{key}_500"]
    got = oracle.kde_logl(tr, H, te)
    assert np.allclose(got, golden[f"kde_logl_{key}_f64"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("variables", VARSETS)
def test_kde_logl_f32(golden, variables):
    # tolerance of the reference test for float32: atol 5e-4 per value (KDE_test.py:181-182)
    key = "".join(variables)
    tr = _sel(golden["train500"], variables).astype(np.float32)
    te = _sel(golden["test50"], variables).astype(np.float32)
    cov, _ = oracle.cov(tr)
    H = oracle.bandwidth(0, 0, cov, tr.shape[0])
    got = oracle.kde_logl(tr, H, te)
    assert np.allclose(got, golden[f"kde_logl_{key}_f32"], atol=5e-4)


@pytest.mark.parametrize("variables", VARSETS)
def test_product_kde_logl(golden, variables):
    key = "".join(variables)
    tr, te = _sel(golden["train500"], variables), _sel(golden["test50"], variables)
    h = golden[f"pkde_bw_nr_{key}_500"]
    assert np.allclose(oracle.product_kde_logl(tr, h, te), golden[f"pkde_logl_{key}_f64"], rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
@pytest.mark.parametrize("tag", ["10k", "10"])
def test_ckde_logl(golden, variable, evidence, tag):
    key = variable + "_" + "".join(evidence)
    tr = _sel(golden["train10k" if tag == "10k" else "train10"], [variable] + evidence)
    te = _sel(golden["test50"], [variable] + evidence)
    H = golden[f"ckde_bw_{key}_{tag}"]
    got = oracle.ckde_logl(tr, H, te)
    assert np.allclose(got, golden[f"ckde_logl_{key}_{tag}"], rtol=1e-8, atol=1e-9)


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
@pytest.mark.parametrize("tag", ["10k", "10"])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_ckde_cdf(golden, variable, evidence, tag, dtype):
    """CKDE_test.py:256-314: isclose (fp64) / atol 5e-4 (fp32) against the scipy recipe."""
    key = variable + "_" + "".join(evidence)
    tr = _sel(golden["train10k" if tag == "10k" else "train10"], [variable] + evidence).astype(dtype)
    te = _sel(golden["test50"], [variable] + evidence).astype(dtype)
    H = golden[f"ckde_bw_{key}_{tag}"]
    got = oracle.ckde_cdf(tr, H, te)
    if dtype == np.float64:
        assert np.allclose(got, golden[f"ckde_cdf_{key}_{tag}"], rtol=1e-8, atol=1e-10)
    else:
        assert np.allclose(got, golden[f"ckde_cdf_{key}_{tag}"], atol=5e-4)


def test_shuffle_known_answer(golden):
    assert np.array_equal(oracle.shuffled_indices(12, 0), golden["shuffle12_seed0"])


def test_cv_limits():
    assert list(oracle.cv_limits(103, 10)) == [0, 11, 22, 33, 43, 53, 63, 73, 83, 93, 103]
    with pytest.raises(ValueError):
        oracle.cv_limits(5, 1)
    with pytest.raises(ValueError):
        oracle.cv_limits(5, 6)
    assert oracle.holdout_test_rows(10000, 0.2) == 2000
    assert oracle.holdout_test_rows(13, 0.5) == 7  # std::round half away from zero


# ---- the tuned CPU baseline (oracle/pbn_baseline.cpp: only ever TIMED by bench.py) is held to the checker and the golden recipes ----
@pytest.mark.parametrize("variables", VARSETS)
def test_tuned_cpu_baseline_matches_golden_and_checker(golden, variables):
    from oracle import baseline

    key = "".join(variables)
    tr, te = _sel(golden["train500"], variables), _sel(golden["test50"], variables)
    H, h = golden[f"kde_bw_nr_{key}_500"], golden[f"pkde_bw_nr_{key}_500"]
    assert np.allclose(baseline.kde_logl(tr, H, te), golden[f"kde_logl_{key}_f64"], rtol=1e-9, atol=1e-10)
    assert np.allclose(baseline.product_kde_logl(tr, h, te), golden[f"pkde_logl_{key}_f64"], rtol=1e-9, atol=1e-10)
    rng = np.random.default_rng(len(variables))
    big = rng.normal(size=(5000, len(variables))) @ (np.eye(len(variables)) + 0.3 * np.tril(np.ones((len(variables),) * 2), -1)).T
    q = rng.normal(size=(37, len(variables))) * 2.0
    cov, _ = oracle.cov(big)
    Hb = oracle.bandwidth(0, 0, cov, big.shape[0])
    assert np.allclose(baseline.kde_logl(big, Hb, q), oracle.kde_logl(big, Hb, q), rtol=1e-10, atol=1e-10)
    assert np.allclose(baseline.ckde_logl(big, Hb, q), oracle.ckde_logl(big, Hb, q), rtol=1e-9, atol=1e-9)
    one = baseline.num_threads()
    baseline.set_num_threads(1)
    try:
        assert np.array_equal(baseline.kde_logl(big, Hb, q), baseline.kde_logl(big, Hb, q))
    finally:
        baseline.set_num_threads(one)


def test_bge_from_cached_moments_is_bge_from_rows():
    """bench.py's C4 tie accounting scores through the whole-table moments (bge.hpp:52-68 caches them): bit-identical to the per-call form."""
    rng = np.random.default_rng(5)
    a = rng.normal(size=(3000, 7))
    a[:, 3] += 0.8 * a[:, 1] - 0.4 * a[:, 6]
    cov, means = oracle.cov(a)
    for sel in ([0], [3, 1], [3, 6, 1], [5, 0, 1, 2, 4]):
        assert oracle.bge_cached(cov, means, a.shape[0], sel, 7) == oracle.bge(a[:, sel], 7)
    threads = oracle.num_threads()
    oracle.set_num_threads(1)
    try:
        cov1, means1 = oracle.cov(a)
    finally:
        oracle.set_num_threads(threads)
    assert np.array_equal(cov, cov1) and np.array_equal(means, means1)   # the parallel covariance does not depend on the thread count


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_linear_gaussian_fit_bic_logl(golden, variable, evidence):
    """The LinearGaussian half of the oracle held to the reference tests' numpy recipes directly (mle_test.py:10-56 lstsq beta and
    residual variance, bic_test.py:10-43 numpy_local_score, LinearGaussianCPD_test.py norm.logpdf) - until round 6 it was pinned only
    through the device (device = golden, device = oracle)."""
    key = variable + "_" + "".join(evidence)
    cols = [COLS.index(v) for v in [variable] + list(evidence)]
    data = np.asarray(golden["train10k"])[:, cols]
    beta, var = oracle.lg_fit(data)
    assert np.allclose(beta, golden[f"lg_beta_{key}"], rtol=1e-9, atol=1e-12)
    assert abs(var - golden[f"lg_var_{key}"]) <= 1e-9 * abs(golden[f"lg_var_{key}"])
    assert abs(oracle.bic_lg(data) - golden[f"bic_{key}"]) <= 1e-9 * abs(golden[f"bic_{key}"])
    test = np.asarray(golden["test50"])[:, cols]
    assert np.allclose(oracle.lg_logl(test, beta, var), golden[f"lg_logl_{key}"], rtol=1e-9, atol=1e-11)
