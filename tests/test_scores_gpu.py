"""GPU tier: BIC / BGe / CVLikelihood / HoldoutLikelihood / ValidatedLikelihood local scores and the
MLE<LinearGaussianCPD> fit from device Gram moments, against the golden numpy recipes of the reference tests
(bic_test.py, mle_test.py), the CPU oracle (QR-based, data-level restatement) and - for BGe, which has no
reference test - an independent numpy transcription of bge.hpp:154-234 written here."""
import math

import numpy as np
import pandas as pd
import pytest
from scipy.special import gammaln

from helpers import CKDE_SETS, COLS, RTOL_F64, frame

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


FULL_ARCS = [("a", "b"), ("a", "c"), ("a", "d"), ("b", "c"), ("b", "d"), ("c", "d")]


def numpy_bge(data, total_nodes, iss_mu=1.0, iss_w=None, nu=None):
    """Independent transcription of BGe::bge_no_parents / bge_parents (bge.hpp:154-234)."""
    data = np.asarray(data, dtype=np.float64)
    N, d = data.shape
    p = d - 1
    iss_w = total_nodes + 2 if iss_w is None else iss_w
    mean = data.mean(axis=0)
    nu = mean if nu is None else np.asarray(nu)
    c = data - mean
    sse = c.T @ c
    t = iss_mu * (iss_w - total_nodes - 1) / (iss_mu + 1)
    lp = 0.5 * (math.log(iss_mu) - math.log(N + iss_mu))
    lp += gammaln(0.5 * (N + iss_w - total_nodes + p + 1)) - gammaln(0.5 * (iss_w - total_nodes + p + 1))
    lp -= 0.5 * N * math.log(math.pi)
    cte = N * iss_mu / (N + iss_mu)
    diff = (mean - nu)[:, None]
    R = sse + t * np.eye(d) + cte * (diff @ diff.T)
    if p == 0:
        lp += 0.5 * (iss_w - total_nodes + 1) * math.log(t)
        return lp - 0.5 * (N + iss_w - total_nodes + 1) * math.log(R[0, 0])
    lp += 0.5 * (iss_w - total_nodes + 2 * p + 1) * math.log(t)
    lp -= 0.5 * (N + iss_w - total_nodes + p + 1) * np.linalg.slogdet(R)[1]
    lp += 0.5 * (N + iss_w - total_nodes + p) * np.linalg.slogdet(R[1:, 1:])[1]
    return lp


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_mle_and_bic_golden(pbn, golden, variable, evidence):
    key = variable + "_" + "".join(evidence)
    df = frame(golden["train10k"])
    bic = pbn.BIC(df)
    beta, var = bic.mle_lg(variable, evidence)
    assert np.allclose(beta, golden[f"lg_beta_{key}"], rtol=1e-7)  # mle_test.py: np.isclose defaults are looser
    assert np.isclose(var, golden[f"lg_var_{key}"], rtol=1e-8)
    gbn = pbn.GaussianNetwork(COLS, FULL_ARCS)
    got = bic.local_score(gbn, variable, evidence)
    assert abs(got - golden[f"bic_{key}"]) <= RTOL_F64 * abs(golden[f"bic_{key}"])
    assert bic.local_score(gbn, variable) == bic.local_score(gbn, variable, gbn.parents(variable))
    if len(evidence) == 3:
        perm = bic.local_score(gbn, variable, ["b", "c", "a"])
        assert abs(perm - got) <= 1e-9 * abs(got)


def test_bic_score_is_sum_of_local(pbn, golden):
    df = frame(golden["train10k"])
    bic = pbn.BIC(df)
    gbn = pbn.GaussianNetwork(COLS, [("a", "b"), ("b", "c")])
    assert np.isclose(bic.score(gbn), sum(bic.local_score(gbn, n) for n in COLS), rtol=1e-12)


def test_bge_vs_oracle_and_numpy(pbn, golden, oracle):
    data = golden["train10k"]
    df = frame(data)
    gbn = pbn.GaussianNetwork(COLS)
    bge = pbn.BGe(df)
    for variable, evidence in CKDE_SETS + [("a", ["d", "c"]), ("c", ["d"])]:
        idx = [COLS.index(v) for v in [variable] + evidence]
        got = bge.local_score(gbn, variable, evidence)
        want_np = numpy_bge(data[:, idx], 4)
        want_or = oracle.bge(data[:, idx], 4)
        assert abs(got - want_np) <= 1e-9 * abs(want_np)
        assert abs(got - want_or) <= 1e-9 * abs(want_or)
    nu = np.array([2.5, 7.0, 15.0, 46.0])
    bge2 = pbn.BGe(df, iss_mu=2.5, iss_w=9.0, nu=nu)
    for variable, evidence in CKDE_SETS:
        idx = [COLS.index(v) for v in [variable] + evidence]
        got = bge2.local_score(gbn, variable, evidence)
        want = numpy_bge(data[:, idx], 4, 2.5, 9.0, nu[idx])
        assert abs(got - want) <= 1e-9 * abs(want)
        assert abs(oracle.bge(data[:, idx], 4, 2.5, 9.0, nu[idx]) - want) <= 1e-9 * abs(want)
    # nu is indexed by DataFrame column (bge.hpp:36-49): a dictionary column in front of the continuous ones must not shift it
    df3 = df.copy()
    df3.insert(0, "k", pd.Categorical.from_codes(np.arange(len(df)) % 3, ["k0", "k1", "k2"]))
    bge3 = pbn.BGe(df3, iss_mu=2.5, iss_w=9.0, nu=np.concatenate([[99.0], nu]))
    for variable, evidence in CKDE_SETS:
        assert bge3.local_score(gbn, variable, evidence) == pytest.approx(bge2.local_score(gbn, variable, evidence), rel=1e-13)
    with pytest.raises(ValueError):
        pbn.BGe(df, iss_w=2.0)


def test_fold_membership_is_the_references(pbn, golden, oracle):
    """Fold layout generated in the library (libstdc++ shuffle) == the oracle's restatement of
    CrossValidationProperties, and the known-answer shuffle vector pins both to the reference."""
    df = frame(golden["train500"])
    cv = pbn.CVLikelihood(df, 7, 123)
    perm, limits = cv.fold_layout()
    assert np.array_equal(perm, oracle.shuffled_indices(500, 123))
    assert np.array_equal(limits, oracle.cv_limits(500, 7))
    with pytest.raises(ValueError, match="Cannot split"):
        pbn.CVLikelihood(df, 1, 0)
    with pytest.raises(ValueError, match="Cannot split"):
        pbn.CVLikelihood(df, 501, 0)
    with pytest.raises(ValueError, match="test_ratio"):
        pbn.HoldoutLikelihood(df, 1.5, 0)


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_cv_and_holdout_likelihood_lg(pbn, golden, oracle, variable, evidence):
    data = golden["train10k"][:2000]
    df = frame(data)
    idx = [COLS.index(v) for v in [variable] + evidence]
    gbn = pbn.GaussianNetwork(COLS)
    cv = pbn.CVLikelihood(df, 10, 0)
    got = cv.local_score(gbn, variable, evidence)
    want = oracle.cv_likelihood(data[:, idx], "lg", 10, 0)
    assert abs(got - want) <= RTOL_F64 * abs(want)
    ho = pbn.HoldoutLikelihood(df, 0.2, 5)
    got = ho.local_score(gbn, variable, evidence)
    want = oracle.holdout_likelihood(data[:, idx], "lg", 0.2, 5)
    assert abs(got - want) <= RTOL_F64 * abs(want)


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_cv_and_holdout_likelihood_ckde(pbn, golden, oracle, variable, evidence):
    data = golden["train10k"][:1500]
    df = frame(data)
    idx = [COLS.index(v) for v in [variable] + evidence]
    spbn = pbn.SemiparametricBN(COLS)
    cv = pbn.CVLikelihood(df, 5, 3)
    got = cv.local_score_node_type(spbn, pbn.CKDEType(), variable, evidence)
    want = oracle.cv_likelihood(data[:, idx], "ckde", 5, 3)
    assert abs(got - want) <= RTOL_F64 * abs(want)
    ho = pbn.HoldoutLikelihood(df, 0.3, 9)
    got = ho.local_score_node_type(spbn, pbn.CKDEType(), variable, evidence)
    want = oracle.holdout_likelihood(data[:, idx], "ckde", 0.3, 9)
    assert abs(got - want) <= RTOL_F64 * abs(want)


def test_validated_likelihood(pbn, golden, oracle):
    data = golden["train10k"][:1200]
    df = frame(data)
    spbn = pbn.SemiparametricBN(COLS, [], [("c", pbn.CKDEType())])
    vl = pbn.ValidatedLikelihood(df, 0.2, 5, 11)
    for variable, evidence in [("c", ["a", "b"]), ("b", ["a"]), ("d", [])]:
        idx = [COLS.index(v) for v in [variable] + evidence]
        kind = "ckde" if variable == "c" else "lg"
        got = vl.local_score(spbn, variable, evidence)
        want = oracle.validated_cv_likelihood(data[:, idx], kind, 0.2, 5, 11)
        assert abs(got - want) <= RTOL_F64 * abs(want)
        gotv = vl.vlocal_score(spbn, variable, evidence)
        wantv = oracle.holdout_likelihood(data[:, idx], kind, 0.2, 11)
        assert abs(gotv - wantv) <= RTOL_F64 * abs(wantv)


def test_wide_table_blocked_gram(pbn, oracle):
    """More than 64 columns: the Gram is assembled from 32-column block pairs."""
    rng = np.random.default_rng(5)
    n, cols = 3000, 70
    data = rng.normal(size=(n, cols)) @ (np.eye(cols) + 0.2 * np.tril(rng.normal(size=(cols, cols)), -1)).T
    names = [f"x{i}" for i in range(cols)]
    df = pd.DataFrame(data, columns=names)
    bic = pbn.BIC(df)
    gbn = pbn.GaussianNetwork(names)
    for var, ev in [(3, [65, 40, 1]), (69, [0, 33, 66, 12, 50]), (10, [])]:
        got = bic.local_score(gbn, names[var], [names[e] for e in ev])
        want = oracle.bic_lg(data[:, [var] + ev])
        assert abs(got - want) <= RTOL_F64 * abs(want)


@pytest.mark.parametrize("p", [3, 4, 6])
@pytest.mark.parametrize("kappa", [1e4, 1e6, 1e8])
def test_near_collinear_parents_vs_pivoted_qr(pbn, oracle, p, kappa, monkeypatch):
    """SURVEY.md §7 hard part (c): >= 3 parents, one of them a near copy of another (condition number ~ kappa).  The reference
    solves by column-pivoted Householder QR (mle_LinearGaussianCPD.hpp:152-193, error ~ kappa eps); the engine's normal
    equations on fp64 moments lose kappa^2 eps and hand such candidates to the double-double refit (lg_accurate.hip).
    beta, variance and the BIC local score must match the oracle's QR; with the guard off the coefficients drift."""
    rng = np.random.default_rng(int(p * 1000 + np.log10(kappa)))
    n = 20000
    X = rng.normal(size=(n, p)) @ (np.eye(p) + 0.3 * np.tril(rng.normal(size=(p, p)), -1)).T + 5.0
    X[:, p - 1] = X[:, 0] + rng.normal(size=n) / kappa
    y = X @ rng.uniform(0.5, 1.5, size=p) + rng.normal(scale=0.5, size=n)
    names = ["y"] + [f"x{i}" for i in range(p)]
    df = pd.DataFrame(np.column_stack([y, X]), columns=names)
    want_beta, want_var = oracle.lg_fit(df.to_numpy())
    cpd = pbn.LinearGaussianCPD("y", names[1:])
    cpd.fit(df)
    # the oracle's QR carries ~kappa eps itself on the two nearly collinear coefficients
    tol = max(1e-8, 200 * kappa * 2.2e-16)
    assert np.allclose(cpd.beta, want_beta, rtol=tol, atol=tol * np.abs(want_beta).max()), (cpd.beta, want_beta)
    assert np.isclose(cpd.variance, want_var, rtol=1e-9)
    bic = pbn.BIC(df)
    got = bic.local_score(pbn.GaussianNetwork(names), "y", names[1:])
    want = oracle.bic_lg(df.to_numpy())
    assert abs(got - want) <= 1e-9 * abs(want)
    cv = pbn.CVLikelihood(df, 4, 3)
    got = cv.local_score(pbn.GaussianNetwork(names), "y", names[1:])
    want = oracle.cv_likelihood(df.to_numpy(), "lg", 4, 3)
    assert abs(got - want) <= RTOL_F64 * abs(want)
    if kappa >= 1e6:
        monkeypatch.setenv("PBN_LG_GUARD", "0")
        raw = pbn.LinearGaussianCPD("y", names[1:])
        raw.fit(df)
        err_raw = np.abs(raw.beta - want_beta).max() / np.abs(want_beta).max()
        err_guard = np.abs(cpd.beta - want_beta).max() / np.abs(want_beta).max()
        assert err_guard <= err_raw and err_raw > 1e-9, (err_raw, err_guard)


def test_degenerate_branches(pbn, oracle):
    """Singular parents (mle_LinearGaussianCPD.hpp:37-49,94-120) and BIC = -inf on degenerate variance."""
    rng = np.random.default_rng(0)
    n = 500
    x = rng.normal(size=n)
    df = pd.DataFrame({"y": 2 * x + rng.normal(scale=0.1, size=n), "x": x, "k": np.full(n, 3.0), "x2": 2 * x, "z": 1.5 * x})
    bic = pbn.BIC(df)
    gbn = pbn.GaussianNetwork(list(df.columns))
    for ev in (["k"], ["x", "k"], ["k", "x"], ["x", "x2"]):
        got = bic.local_score(gbn, "y", ev)
        want = oracle.bic_lg(df[["y"] + ev].to_numpy())
        assert abs(got - want) <= RTOL_F64 * abs(want), ev
    assert bic.local_score(gbn, "z", ["x"]) == -np.inf  # exact linear function: variance < machine_tol
    assert bic.local_score(gbn, "k", []) == -np.inf


@pytest.mark.parametrize("variable,evidence", CKDE_SETS)
def test_linear_gaussian_cpd_fit_logl(pbn, golden, variable, evidence):
    """LinearGaussianCPD_test.py:26-120 / mle_test.py: lstsq beta, residual variance, norm.logpdf."""
    key = variable + "_" + "".join(evidence)
    cpd = pbn.LinearGaussianCPD(variable, evidence)
    cpd.fit(frame(golden["train10k"]))
    assert np.allclose(cpd.beta, golden[f"lg_beta_{key}"], rtol=1e-7)
    assert np.isclose(cpd.variance, golden[f"lg_var_{key}"], rtol=1e-8)
    test = frame(golden["test50"])
    want = golden[f"lg_logl_{key}"]
    assert np.allclose(cpd.logl(test), want, rtol=1e-7, atol=1e-9)
    assert abs(cpd.slogl(test) - want.sum()) <= RTOL_F64 * abs(want.sum())
    tn = frame(golden["test50_null"])
    ll = cpd.logl(tn)
    nulls = np.any(np.isnan(tn[[variable] + evidence].to_numpy()), axis=1)
    assert np.array_equal(np.isnan(ll), nulls)
    assert np.isclose(cpd.slogl(tn), np.nansum(ll))
    p = pbn.MLE(pbn.LinearGaussianCPDType()).estimate(frame(golden["train10k"]), variable, evidence)
    assert np.allclose(p.beta, cpd.beta) and p.variance == cpd.variance
    with pytest.raises(ValueError, match="MLE not available"):
        pbn.MLE(pbn.CKDEType())
    cpd32 = pbn.LinearGaussianCPD(variable, evidence)
    cpd32.fit(frame(golden["train10k"], "float32"))
    assert np.allclose(cpd32.beta, golden[f"lg_beta_{key}"], rtol=2e-3, atol=2e-3)


def test_scores_with_nulls(pbn, golden, oracle):
    """bic_test.py:45-85 (test_bic_local_score_null): BIC over the rows valid in [variable]+parents; BGe likewise;
    CV/holdout scores drop rows with a null in any column before splitting (crossvalidation_adaptator.hpp:24-37)."""
    data = golden["train10k"][:3000].copy()
    np.random.seed(0)
    for j in range(4):
        data[np.random.randint(0, data.shape[0], size=100), j] = np.nan
    df = frame(data)
    gbn = pbn.GaussianNetwork(COLS, FULL_ARCS)
    bic, bge = pbn.BIC(df), pbn.BGe(df)
    for variable, evidence in CKDE_SETS:
        idx = [COLS.index(v) for v in [variable] + evidence]
        sub = data[:, idx]
        sub = sub[~np.isnan(sub).any(axis=1)]
        want = oracle.bic_lg(sub)
        got = bic.local_score(gbn, variable, evidence)
        assert abs(got - want) <= RTOL_F64 * abs(want)
        wantb = numpy_bge(sub, 4)
        assert abs(bge.local_score(gbn, variable, evidence) - wantb) <= 1e-9 * abs(wantb)
    clean = data[~np.isnan(data).any(axis=1)]
    cv = pbn.CVLikelihood(df, 5, 2)
    for variable, evidence, kind, nt in [("c", ["a", "b"], "lg", pbn.LinearGaussianCPDType()), ("b", ["a"], "ckde", pbn.CKDEType())]:
        idx = [COLS.index(v) for v in [variable] + evidence]
        want = oracle.cv_likelihood(clean[:, idx], kind, 5, 2)
        got = cv.local_score_node_type(pbn.SemiparametricBN(COLS), nt, variable, evidence)
        assert abs(got - want) <= RTOL_F64 * abs(want)


def test_fp32_scores(pbn, golden, oracle):
    """fp32 tables (config C5 dtype): statistics are accumulated in double from the float data; sweeps run on
    v_mfma_f32 + v_exp_f32.  Tolerance: 1e-3 relative (north star) against the fp64 oracle on the same rounded data."""
    data32 = golden["train10k"][:1500].astype(np.float32)
    df = frame(data32, "float32")
    data = data32.astype(np.float64)
    spbn = pbn.SemiparametricBN(COLS)
    cv = pbn.CVLikelihood(df, 4, 1)
    for variable, evidence in [("b", ["a"]), ("c", ["a", "b"])]:
        idx = [COLS.index(v) for v in [variable] + evidence]
        for kind, nt in (("lg", pbn.LinearGaussianCPDType()), ("ckde", pbn.CKDEType())):
            got = cv.local_score_node_type(spbn, nt, variable, evidence)
            want = oracle.cv_likelihood(data[:, idx], kind, 4, 1)
            assert abs(got - want) <= 1e-3 * abs(want), (variable, kind, got, want)
    bic = pbn.BIC(df)
    for variable, evidence in CKDE_SETS:
        idx = [COLS.index(v) for v in [variable] + evidence]
        want = oracle.bic_lg(data[:, idx])
        assert abs(bic.local_score(pbn.GaussianNetwork(COLS), variable, evidence) - want) <= 1e-3 * abs(want)


def test_split_accessors_and_kde_views(pbn, golden, oracle):
    """CVLikelihood.cv, Holdout training_data/test_data, ValidatedLikelihood.cv_lik/holdout_lik
    (pybindings_scores.cpp:516-660) and CKDE.kde_joint/kde_marg (pybindings_factors.cpp:583-640)."""
    data = golden["train10k"][:400]
    df = frame(data)
    cv = pbn.CVLikelihood(df, 4, 6)
    folds = list(cv.cv.indices())
    want = oracle.cv_folds(400, 4, 6)
    assert len(folds) == 4
    for (tr, te), (wtr, wte) in zip(folds, want):
        assert np.array_equal(tr, wtr) and np.array_equal(te, wte)
    tr_df, te_df = cv.cv.fold(1)
    assert tr_df.num_rows + te_df.num_rows == 400
    assert np.allclose(te_df.column(0).to_numpy(), data[want[1][1], 0])
    ho = pbn.HoldoutLikelihood(df, 0.25, 3)
    wtr, wte = oracle.holdout_split(400, 0.25, 3)
    assert np.allclose(ho.training_data().column(1).to_numpy(), data[wtr, 1])
    assert np.allclose(ho.test_data().column(1).to_numpy(), data[wte, 1])
    vl = pbn.ValidatedLikelihood(df, 0.2, 3, 8)
    net = pbn.SemiparametricBN(COLS)
    assert vl.cv_lik.local_score(net, "b", ["a"]) == vl.local_score(net, "b", ["a"])
    assert vl.holdout_lik.local_score(net, "b", ["a"]) == vl.vlocal_score(net, "b", ["a"])
    assert vl.training_data().num_rows + vl.validation_data().num_rows == 400
    cpd = pbn.CKDE("c", ["a", "b"])
    cpd.fit(df)
    test = frame(golden["test50"])
    lj, lm = cpd.kde_joint().logl(test), cpd.kde_marg().logl(test)
    assert np.allclose(lj - lm, cpd.logl(test), rtol=1e-9, atol=1e-10)


@pytest.mark.parametrize("world", [2, 3, 5, 8])
def test_row_sharded_moments(pbn, world):
    """SURVEY.md §8e (BGe / BIC / LG-CV): every region is cut into a fixed number of super-blocks; rank r takes the Gram of its
    share of those segments, the ranks' buffers are added (every segment is non-zero on one rank only) and installed, and the
    regions' totals are rebuilt in segment order: the scores are BIT-IDENTICAL to the unsharded handle's for every world size
    (so a tie of a BGe / BIC hill-climb is broken the same way on 1, 2, 4 or 8 GPUs).  A handle whose totals have not been
    installed is refused."""
    import ctypes as C

    from pybnesian_amd import _lib

    rng = np.random.default_rng(11)
    n, rows = 6, 10007
    data = rng.normal(size=(rows, n)) @ (np.eye(n) + np.triu(rng.uniform(-0.5, 0.5, (n, n)), 1)) + 3.0
    df = pd.DataFrame(data, columns=[f"v{i}" for i in range(n)])
    lib = _lib.load()
    for make, kind, split, args in (
        (lambda: pbn.BGe(df), _lib.PBN_SCORE_BGE, _lib.PBN_SPLIT_NONE, (0, 0, 0.0)),
        (lambda: pbn.BIC(df), _lib.PBN_SCORE_BIC, _lib.PBN_SPLIT_NONE, (0, 0, 0.0)),
        (lambda: pbn.CVLikelihood(df, k=7, seed=5), _lib.PBN_SCORE_CVLIK, _lib.PBN_SPLIT_CV, (7, 5, 0.0)),
        (lambda: pbn.ValidatedLikelihood(df, test_ratio=0.3, k=4, seed=2), _lib.PBN_SCORE_HOLDOUT, _lib.PBN_SPLIT_VALIDATED, (4, 2, 0.3)),
    ):
        score = make()
        model = pbn.GaussianNetwork(list(df.columns))
        var, ntype, off, par = [0, 1, 5, 3], [0, 0, 0, 0], [0, 0, 1, 3, 8], [0, 0, 1, 0, 1, 2, 4, 5]
        want = score._batch_raw(model, var, ntype, off, par, kind)
        handles, bufs = [], []
        for r in range(world):
            h = C.c_void_p()
            _lib.check(lib.pbn_scoredata_create_sharded(score._ctx.handle, score._table.handle, split, args[0],
                                                        C.c_uint32(args[1]), args[2], r, world, C.byref(h)))
            ln = C.c_int64(0)
            _lib.check(lib.pbn_scoredata_moments(h, None, C.byref(ln), 0))
            b = np.zeros(ln.value)
            _lib.check(lib.pbn_scoredata_moments(h, _lib.dptr(b), C.byref(ln), 0))
            handles.append(h)
            bufs.append(b)
        out = np.zeros(len(var))
        params = score._batch_params(model)
        call = lambda h: lib.pbn_score_batch(h, kind, len(var), _lib.int_array(var), _lib.int_array(ntype), _lib.int_array(off),
                                             _lib.int_array(par), _lib.dptr(params) if params.size else None, int(params.size),
                                             _lib.dptr(out))
        with pytest.raises(ValueError, match="row-sharded"):
            _lib.check(call(handles[0]))
        total = bufs[0].copy()
        for b in bufs[1:]:
            total += b
        for h in handles:
            ln = C.c_int64(total.size)
            _lib.check(lib.pbn_scoredata_moments(h, _lib.dptr(total), C.byref(ln), 1))
            _lib.check(call(h))
            assert np.array_equal(out, want), (world, kind, out - want)
            lib.pbn_scoredata_destroy(h)


def test_construction_args_bandwidth_selector(pbn):
    """CVLikelihood(df, k, seed, Arguments({CKDEType(): (ScottsBandwidth(),)})): every CKDE fitted while scoring uses
    that selector (cv_likelihood.hpp:19-27); checked against explicit per-fold CKDE fits."""
    rng = np.random.default_rng(21)
    n = 3000
    a = rng.normal(size=n)
    b = np.sin(a) + rng.normal(scale=0.3, size=n)
    c = 0.5 * a - b + rng.normal(scale=0.5, size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c})
    model = pbn.SemiparametricBN(["a", "b", "c"], [], [("c", pbn.CKDEType())])
    for sel in (pbn.ScottsBandwidth(), pbn.NormalReferenceRule()):
        for args in (pbn.Arguments({pbn.CKDEType(): (sel,)}), pbn.Arguments({pbn.CKDEType(): pbn.Kwargs(bandwidth_selector=sel)})):
            score = pbn.CVLikelihood(df, k=4, seed=9, construction_args=args)
            want = 0.0
            for tr, te in score.cv.indices():
                cpd = pbn.CKDE("c", ["a", "b"], sel)
                cpd.fit(df.iloc[tr])
                want += cpd.slogl(df.iloc[te])
            got = score.local_score(model, "c", ["a", "b"])
            assert abs(got - want) <= 1e-9 * abs(want)
    plain = pbn.CVLikelihood(df, k=4, seed=9).local_score(model, "c", ["a", "b"])
    scott = pbn.CVLikelihood(df, k=4, seed=9, construction_args=pbn.Arguments({pbn.CKDEType(): (pbn.ScottsBandwidth(),)}))
    assert abs(scott.local_score(model, "c", ["a", "b"]) - plain) > 1e-3
    with pytest.raises(ValueError, match="construction arguments"):
        pbn.CVLikelihood(df, k=4, seed=9, construction_args=pbn.Arguments({"c": (1,)}))
    hold = pbn.HoldoutLikelihood(df, 0.25, 3, pbn.Arguments({pbn.CKDEType(): (pbn.ScottsBandwidth(),)}))
    cpd = pbn.CKDE("c", ["a"], pbn.ScottsBandwidth())
    cpd.fit(hold.training_data())
    want = cpd.slogl(hold.test_data())
    assert abs(hold.local_score(model, "c", ["a"]) - want) <= 1e-9 * abs(want)


def test_ckde_set_function_cache_paths(pbn):
    """CKDE likelihood scores through the engine's set-function cache: whichever of the fused / joint-only /
    marginal-only / fully-cached paths a candidate takes (it depends on what was scored before), the value equals the
    explicit per-fold CKDE fit + slogl; all ordered pairs cost one joint sweep per unordered pair."""
    rng = np.random.default_rng(31)
    n, k = 2500, 3
    a = rng.normal(size=n)
    b = np.tanh(a) + rng.normal(scale=0.4, size=n)
    c = a * b + rng.normal(scale=0.5, size=n)
    d = rng.normal(size=n) - 0.5 * c
    df = pd.DataFrame({"a": a, "b": b, "c": c, "d": d})
    names = list(df.columns)
    model = pbn.SemiparametricBN(names, [], [(v, pbn.CKDEType()) for v in names])

    def explicit(score, var, ev):
        tot = 0.0
        for tr, te in score.cv.indices():
            cpd = pbn.CKDE(var, ev)
            cpd.fit(df.iloc[tr])
            tot += cpd.slogl(df.iloc[te])
        return tot

    cands = [(v, [e]) for v in names for e in names if e != v] + [("a", []), ("c", ["a", "b"]), ("d", ["c", "a", "b"]), ("b", ["a", "c"])]
    orders = [list(range(len(cands))), list(reversed(range(len(cands)))), list(rng.permutation(len(cands)))]
    results = []
    for order in orders:
        score = pbn.CVLikelihood(df, k=k, seed=4)
        got = {}
        for i in order:                                   # one candidate per call: every cache state is exercised
            v, ev = cands[i]
            got[i] = score.local_score(model, v, ev)
        results.append(got)
        if order is orders[0]:
            want = {i: explicit(score, *cands[i]) for i in range(len(cands))}
        for i in range(len(cands)):
            assert abs(got[i] - want[i]) <= 1e-9 * abs(want[i]), (cands[i], got[i], want[i])
    for got in results[1:]:
        assert all(abs(got[i] - results[0][i]) <= 1e-11 * abs(results[0][i]) for i in got)
    # batched: the 12 ordered pairs need 6 joint sets + 4 marginal sets per fold, however they are evaluated
    score = pbn.CVLikelihood(df, k=k, seed=4)
    pairs = [(v, pbn.CKDEType(), [e]) for v in names for e in names if e != v]
    vals = score._batch(model, pairs, score._kind)
    entries, sweeps = score.kde_cache_stats()
    assert entries == (6 + 4) * k and sweeps <= (6 + 1) * k + 4 * k
    assert np.allclose(vals, [results[0][i] for i in range(12)], rtol=1e-11)
    again = score._batch(model, pairs, score._kind)      # everything cached now: no new sweeps, identical values
    assert score.kde_cache_stats() == (entries, sweeps) and np.array_equal(vals, again)


@pytest.mark.parametrize("dtype,pin,rel", [("float64", {"PBN_PRUNE_MARGIN": "52"}, 1e-11), ("float64", {}, 3e-7),
                                           ("float32", {"PBN_PRUNE_MARGIN_F32": "40"}, 2e-6), ("float32", {}, 3e-5)])
def test_pruned_sweeps_match_unpruned(pbn, dtype, pin, rel, monkeypatch):
    """Low-dimensional CKDE candidates of the score engine run the Morton-sorted, tile-pruned sweep once the training folds
    are large enough.  With the margin pinned at 52 (fp32: 40) what is dropped is below 2.2e-10 (9e-7) of a sum: the machinery
    reproduces the unpruned scores to 1e-11 (2e-6).  The default margins of the sum-only sweeps (43 / 36 at 10^6 rows, prune_margin)
    are chosen so that the dropped-mass BOUND equals the arithmetic's own error - 1.1e-7 (1.5e-5) of a sum - and are held to it."""
    import os

    for k_, v_ in pin.items():
        monkeypatch.setenv(k_, v_)

    rng = np.random.default_rng(11)
    n = 90000
    a = rng.normal(size=n)
    b = np.tanh(a) + 0.4 * rng.normal(size=n)
    c = 0.5 * a - 0.7 * b + 0.5 * rng.standard_t(5, size=n)       # heavy tails: far-away queries
    d = rng.normal(size=n) * (1.0 + 0.5 * (a > 1))
    e = 0.3 * c + rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c, "d": d, "e": e}).astype(dtype)
    model = pbn.SemiparametricBN(list(df.columns))
    cands = [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"]), ("e", ["a", "b", "c", "d"])]
    values = {}
    for flag in ("0", "1"):
        os.environ["PBN_SWEEP_PRUNE"] = flag
        try:
            score = pbn.CVLikelihood(df, k=3, seed=5)
            values[flag] = [score.local_score_node_type(model, pbn.CKDEType(), v, ev) for v, ev in cands]
            vl = pbn.ValidatedLikelihood(df, test_ratio=0.25, k=2, seed=1)
            values[flag] += [vl.vlocal_score_node_type(model, pbn.CKDEType(), "c", ["a", "b"])]
        finally:
            os.environ.pop("PBN_SWEEP_PRUNE", None)
    for off, on in zip(values["0"], values["1"]):
        assert np.isfinite(off) and on == pytest.approx(off, rel=rel)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("d", [1, 7, 16, 17, 32, 40, 48, 49, 64])
def test_device_table_sse_shapes_and_offsets(pbn, d, dtype):
    """DataFrame::means / ::sse (dataset.hpp:208-234, 340-512) from the device Gram over row ranges that start at odd rows, end
    inside a 32-row chunk, or hold fewer rows than one chunk - every column-tile count (1-4 tiles of 16) of the Gram kernels
    (LDS-DMA for double tables, register staging for float ones), with a large mean so that the pilot shift matters."""
    rng = np.random.default_rng(100 + d)
    n = 70001
    data = (rng.normal(size=(n, d)) @ (np.eye(d) + 0.3 * np.tril(rng.normal(size=(d, d)), -1)).T + 1000.0).astype(dtype)
    names = [f"x{i}" for i in range(d)]
    table, _ = pbn.DeviceTable.from_dataframe(pbn.default_context(), pd.DataFrame(data, columns=names), names)
    cols = [names[i] for i in rng.permutation(d)]
    idx = [names.index(c) for c in cols]
    for row0, rows in [(0, n), (1, n - 1), (3, 65536), (4097, 31), (12345, 33), (7, 1), (100, 20000 + d)]:
        x = data[row0:row0 + rows][:, idx].astype(np.float64)
        want_mean = x.mean(axis=0)
        c = x - want_mean
        want = c.T @ c
        means, sse = table.sse(cols, row0, rows)
        tol = 1e-10 if dtype == "float64" else 1e-9
        assert np.allclose(means, want_mean, rtol=tol, atol=0), (row0, rows)
        scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1.0
        assert np.max(np.abs(sse - want) / scale) <= 1e-9, (row0, rows, np.max(np.abs(sse - want) / scale))


def test_gram_random_shapes_and_ranges(pbn):
    """Short form of tools/gram_fuzz.py: random (rows, columns, first row, dtype, column subset) against numpy - the LDS-DMA Gram
    kernels count their DMA completions by hand, a miscount would show as a wrong tile for some shapes only."""
    rng = np.random.default_rng(77)
    ctx = pbn.default_context()
    for case in range(24):
        d = int(rng.integers(1, 65))
        n = int(rng.choice([int(rng.integers(1, 300)), int(rng.integers(300, 20000)), int(rng.integers(20000, 150000))]))
        dtype = "float64" if case % 2 == 0 else "float32"
        data = (rng.normal(size=(n, d)) * rng.uniform(0.5, 3.0, size=d) + rng.uniform(-50, 50, size=d)).astype(dtype)
        names = [f"x{i}" for i in range(d)]
        table, _ = pbn.DeviceTable.from_dataframe(ctx, pd.DataFrame(data, columns=names), names)
        for _ in range(3):
            row0 = int(rng.integers(0, max(1, n // 3)))
            rows = int(rng.integers(1, n - row0 + 1))
            k = int(rng.integers(1, d + 1))
            idx = [int(i) for i in rng.choice(d, size=k, replace=False)]
            x = data[row0:row0 + rows][:, idx].astype(np.float64)
            mean = x.mean(axis=0)
            c = x - mean
            want = c.T @ c
            means, sse = table.sse([names[i] for i in idx], row0, rows)
            scale = np.sqrt(np.outer(np.diag(want), np.diag(want))) + 1.0
            assert np.max(np.abs(sse - want) / scale) < 1e-9, (case, n, d, dtype, row0, rows)
            assert np.max(np.abs(means - mean) / (np.abs(mean) + 1.0)) < 1e-10, (case, n, d, dtype, row0, rows)


def test_cv_likelihood_ckde_with_18_parents(pbn, oracle):
    """A CKDE candidate over 19 variables (more than the 16 whitened dimensions one fp64 MFMA chain of four covers): the score
    engine's folds go through the 17-32-dimension sweeps (KS = 5, conditional form)."""
    rng = np.random.default_rng(19)
    n, d = 1500, 19
    mix = np.tril(rng.uniform(-0.3, 0.3, size=(d, d)), -1) + np.eye(d)
    data = rng.normal(size=(n, d)) @ mix.T
    names = [f"v{i}" for i in range(d)]
    df = pd.DataFrame(data, columns=names)
    score = pbn.CVLikelihood(df, 3, 5)
    got = score.local_score_node_type(pbn.SemiparametricBN(names), pbn.CKDEType(), "v0", names[1:])
    want = oracle.cv_likelihood(data, "ckde", 3, 5)
    assert abs(got - want) <= RTOL_F64 * abs(want)


def test_cv_likelihood_ckde_with_40_parents(pbn, oracle):
    """A CKDE candidate over 41 variables: beyond the templated sweeps (32 whitened dimensions) the score engine's terms - plain
    joint and marginal sweeps - take the generic runtime-sized path (kde_kernels.hpp "wide"); a KDENetwork search without
    max_indegree on a wide table can reach such candidates (the reference's kernels loop over any d: KDE.hpp:592-640)."""
    rng = np.random.default_rng(41)
    n, d = 1200, 41
    mix = np.tril(rng.uniform(-0.2, 0.2, size=(d, d)), -1) + np.eye(d)
    data = rng.normal(size=(n, d)) @ mix.T
    names = [f"v{i}" for i in range(d)]
    df = pd.DataFrame(data, columns=names)
    score = pbn.CVLikelihood(df, 3, 7)
    got = score.local_score_node_type(pbn.SemiparametricBN(names), pbn.CKDEType(), "v0", names[1:])
    want = oracle.cv_likelihood(data, "ckde", 3, 7)
    assert abs(got - want) <= RTOL_F64 * abs(want)
    hold = pbn.HoldoutLikelihood(df, 0.25, 7)
    got = hold.local_score_node_type(pbn.SemiparametricBN(names), pbn.CKDEType(), "v3", names[4:40])
    want = oracle.holdout_likelihood(data[:, [3] + list(range(4, 40))], "ckde", 0.25, 7)
    assert abs(got - want) <= RTOL_F64 * abs(want)


@pytest.mark.parametrize("kind", ["cv", "holdout"])
def test_ckde_terms_are_the_halves_of_the_local_score(pbn, kind):
    """pbn_score_terms / _put / _missing (a job with one process per GPU deals the TERMS of a delta-cache batch):
    local(v | P) = A({v} u P, d) - A(P, d) bit for bit; totals installed in a fresh handle reproduce the scores without a sweep."""
    from pybnesian_amd import _lib

    rng = np.random.default_rng(8)
    n = 6000
    a = rng.normal(size=n)
    b = 0.6 * a + rng.normal(scale=0.7, size=n)
    c = np.sin(a) + 0.3 * b + rng.normal(scale=0.5, size=n)
    d = rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c, "d": d})
    make = (lambda: pbn.CVLikelihood(df, k=4, seed=1)) if kind == "cv" else (lambda: pbn.HoldoutLikelihood(df, test_ratio=0.25, seed=1))
    code = _lib.PBN_SCORE_CVLIK if kind == "cv" else _lib.PBN_SCORE_HOLDOUT
    net = pbn.SemiparametricBN(list("abcd"), [], [(v, pbn.CKDEType()) for v in "abcd"])
    cands = [("c", ["a", "b"]), ("b", ["a"]), ("a", ["b"]), ("d", []), ("c", ["b", "a"])]
    col = {v: i for i, v in enumerate("abcd")}
    score = make()
    want = [score.local_score(net, v, p) for v, p in cands]
    terms = []
    for v, p in cands:
        m = len(p) + 1
        terms.append((m,) + tuple(col[x] for x in [v] + p))
        if p:
            terms.append((m,) + tuple(col[x] for x in p))
    fresh = make()
    assert fresh._terms("missing", code, terms) == [1] * len(terms)
    vals = fresh._terms("eval", code, terms)
    it = iter(vals)
    for (v, p), w in zip(cands, want):
        j = next(it)
        mg = next(it) if p else 0.0
        assert j - mg == w, (v, p)
    # A({a, b}, 2) serves a -> b and b -> a; the column order of a term does not matter
    assert fresh._terms("eval", code, [(2, col["a"], col["b"]), (2, col["b"], col["a"])]).tolist() == [vals[2], vals[2]]
    other = make()
    other._terms("put", code, terms, vals)
    assert other._terms("missing", code, terms + [(2, col["d"], col["a"])]) == [0] * len(terms) + [1]
    before = other.kde_cache_stats()[1]
    assert [other.local_score(net, v, p) for v, p in cands] == want
    assert other.kde_cache_stats()[1] == before        # assembled from the installed totals: not one sweep
    with pytest.raises(ValueError, match="term"):
        fresh._terms("eval", code, [(4, col["a"], col["b"])])
    # pbn_score_term_regions: a term's total is its regions (CV folds; one hold-out region) added in region order - evaluated one region
    # at a time, in any order, by a handle that has seen nothing else
    regions = fresh._term_regions(code)
    assert regions == (4 if kind == "cv" else 1)
    single = make()
    items = [(t, f) for f in reversed(range(regions)) for t in terms]
    got = single._terms("eval_regions", code, [t for t, _ in items], regions=[f for _, f in items])
    per = {}
    for (t, f), v in zip(items, got):
        per[(t, f)] = v
    for t, total in zip(terms, vals):
        acc = 0.0
        for f in range(regions):
            acc += per[(t, f)]
        assert acc == total, t
    with pytest.raises(ValueError, match="region"):
        single._terms("eval_regions", code, [terms[0]], regions=[regions])


def test_marginal_terms_do_not_depend_on_an_unrelated_column(pbn):
    """A marginal term A(P, m) is the KDE of P under the rule for m dimensions - no child in it (advisor, round 3: it used to be
    evaluated as a pseudo-candidate whose child was the smallest column outside P, so a constant / collinear column 0 failed terms
    no real candidate of a restricted search pairs with it).  With a constant first column every term over the other columns
    still evaluates, equals the halves of the real candidates' local scores bit for bit, and a term that DOES contain the
    degenerate column raises SingularCovarianceData - not a generic device error."""
    from pybnesian_amd import _lib

    rng = np.random.default_rng(12)
    n = 5000
    a = rng.normal(size=n)
    b = 0.6 * a + rng.normal(scale=0.7, size=n)
    c = np.tanh(a) + 0.3 * b + rng.normal(scale=0.5, size=n)
    df = pd.DataFrame({"k": np.full(n, 3.25), "a": a, "b": b, "c": c})
    score = pbn.CVLikelihood(df, k=4, seed=3)
    code = _lib.PBN_SCORE_CVLIK
    net = pbn.SemiparametricBN(list(df.columns), [], [(v, pbn.CKDEType()) for v in df.columns])
    col = {v: i for i, v in enumerate(df.columns)}
    cands = [("c", ["a", "b"]), ("b", ["a"]), ("a", ["c"])]
    want = [score.local_score(net, v, p) for v, p in cands]
    fresh = pbn.CVLikelihood(df, k=4, seed=3)
    for (v, p), w in zip(cands, want):
        m = len(p) + 1
        j, mg = fresh._terms("eval", code, [(m,) + tuple(col[x] for x in [v] + p), (m,) + tuple(col[x] for x in p)])
        assert j - mg == w, (v, p)
    # every column but the constant one in one marginal term: there is no column outside it to borrow as a child any more
    assert np.isfinite(fresh._terms("eval", code, [(4, col["a"], col["b"], col["c"])])).all()
    with pytest.raises(pbn.SingularCovarianceData):
        fresh._terms("eval", code, [(3, col["k"], col["a"])])
    with pytest.raises(pbn.SingularCovarianceData):
        fresh._terms("eval", code, [(2, col["k"])])


def test_ckde_terms_of_a_validated_score_keep_their_kind(pbn):
    """ValidatedLikelihood asks ONE handle for both kinds (CV over the training part: local_score; the hold-out part: vlocal_score).
    The installed totals are keyed by kind: the same variable set has one total per kind."""
    from pybnesian_amd import _lib

    rng = np.random.default_rng(9)
    n = 5000
    a = rng.normal(size=n)
    b = 0.5 * a + rng.normal(scale=0.8, size=n)
    c = np.cos(a) + 0.4 * b + rng.normal(scale=0.5, size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c})
    make = lambda: pbn.ValidatedLikelihood(df, test_ratio=0.3, k=3, seed=4)
    net = pbn.SemiparametricBN(list("abc"), [], [(v, pbn.CKDEType()) for v in "abc"])
    ref = make()
    want = [(ref.local_score(net, "c", ["a", "b"]), ref.vlocal_score(net, "c", ["a", "b"])), (ref.local_score(net, "b", ["a"]), ref.vlocal_score(net, "b", ["a"]))]
    terms = [(3, 2, 0, 1), (3, 0, 1), (2, 1, 0), (2, 0)]
    src, dst = make(), make()
    for kind in (_lib.PBN_SCORE_CVLIK, _lib.PBN_SCORE_HOLDOUT):
        dst._terms("put", kind, terms, src._terms("eval", kind, terms))
    assert dst._terms("missing", _lib.PBN_SCORE_CVLIK, terms) == [0] * 4 and dst._terms("missing", _lib.PBN_SCORE_HOLDOUT, terms) == [0] * 4
    before = dst.kde_cache_stats()[1]
    got = [(dst.local_score(net, "c", ["a", "b"]), dst.vlocal_score(net, "c", ["a", "b"])), (dst.local_score(net, "b", ["a"]), dst.vlocal_score(net, "b", ["a"]))]
    assert got == want and want[0][0] != want[0][1]
    assert dst.kde_cache_stats()[1] == before
