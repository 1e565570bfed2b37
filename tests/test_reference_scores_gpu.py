"""GPU tier: the reference's CVLikelihood / HoldoutLikelihood tests (/root/reference/tests/learning/scores/
cvlikelihood_test.py, holdoutlikelihood_test.py) re-typed: the package's scores against the scipy / numpy recipe
evaluated on the folds of the package's own CrossValidation / HoldOut objects, with and without nulls."""
import numpy as np
import pytest
from scipy.stats import gaussian_kde, norm

from helpers import frame

pytestmark = pytest.mark.gpu
SIZE, SEED = 1000, 0
FULL = [("a", "b"), ("a", "c"), ("a", "d"), ("b", "c"), ("b", "d"), ("c", "d")]


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def df(golden):
    return frame(golden["train10k"]).iloc[:SIZE].reset_index(drop=True)


@pytest.fixture(scope="module")
def df_null(df):
    np.random.seed(0)
    out = df.copy()
    for c in "abcd":
        out.loc[out.index[np.random.randint(0, SIZE, size=100)], c] = np.nan
    return out


def recipe(pbn, kind, splits, variable, evidence):
    """log-likelihood of the test parts under a factor fitted on the train parts (cvlikelihood_test.py:12-52)."""
    total = 0.0
    for train, test in splits:
        tr = train.to_pandas().loc[:, [variable] + evidence].dropna()
        te = test.to_pandas().loc[:, [variable] + evidence].dropna()
        if kind == "lg":
            n, d = tr.shape[0], len(evidence)
            A = np.column_stack((np.ones(n), tr[evidence].to_numpy()))
            beta, res, _, _ = np.linalg.lstsq(A, tr[variable].to_numpy(), rcond=None)
            var = res / (n - d - 1)
            means = beta[0] + (te[evidence].to_numpy() @ beta[1:] if evidence else 0.0)
            total += norm.logpdf(te[variable].to_numpy(), means, np.sqrt(var)).sum()
        else:
            kj = gaussian_kde(tr.to_numpy().T, bw_method=lambda s: np.power(4 / (s.d + 2), 1 / (s.d + 4)) * s.scotts_factor())
            ll = kj.logpdf(te.to_numpy().T)
            if evidence:
                km = gaussian_kde(tr[evidence].to_numpy().T, bw_method=kj.covariance_factor())
                ll = ll - km.logpdf(te[evidence].to_numpy().T)
            total += ll.sum()
    return float(total)


def kind_of(pbn, t):
    return "ckde" if t == pbn.CKDEType() else "lg"


def test_cvl_create(pbn, df):   # cvlikelihood_test.py:54-73
    assert len(list(pbn.CVLikelihood(df).cv)) == 10 and len(list(pbn.CVLikelihood(df, 5).cv)) == 5
    s, s2 = pbn.CVLikelihood(df, 10, 0), pbn.CVLikelihood(df, 10, 0)
    for (a, b), (a2, b2) in zip(s.cv, s2.cv):
        assert a.equals(a2) and b.equals(b2)
    with pytest.raises(ValueError, match="Cannot split"):
        pbn.CVLikelihood(df, SIZE + 1)


@pytest.mark.parametrize("with_nulls", [False, True])
def test_cvl_local_scores(pbn, df, df_null, with_nulls):   # cvlikelihood_test.py:75-207
    data = df_null if with_nulls else df
    cvl = pbn.CVLikelihood(data, 10, SEED)
    folds = lambda: pbn.CrossValidation(data, 10, SEED)
    gbn = pbn.GaussianNetwork(FULL)
    spbn = pbn.SemiparametricBN(FULL, [("a", pbn.CKDEType()), ("c", pbn.CKDEType())])
    cases = [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]
    for v, ev in cases:
        assert np.isclose(cvl.local_score(gbn, v, ev), recipe(pbn, "lg", folds(), v, ev))
        assert np.isclose(cvl.local_score(spbn, v, ev), recipe(pbn, kind_of(pbn, spbn.node_type(v)), folds(), v, ev))
        assert cvl.local_score(gbn, v) == cvl.local_score(gbn, v, gbn.parents(v))
        assert cvl.local_score(spbn, v) == cvl.local_score(spbn, v, spbn.parents(v))
        other = pbn.LinearGaussianCPDType() if spbn.node_type(v) == pbn.CKDEType() else pbn.CKDEType()
        assert np.isclose(cvl.local_score_node_type(spbn, other, v, ev), recipe(pbn, kind_of(pbn, other), folds(), v, ev))
    assert np.isclose(cvl.local_score(gbn, "d", ["a", "b", "c"]), cvl.local_score(gbn, "d", ["b", "c", "a"]))
    assert np.isclose(cvl.local_score_node_type(spbn, pbn.CKDEType(), "d", ["a", "b", "c"]), recipe(pbn, "ckde", folds(), "d", ["b", "c", "a"]))
    for m in (gbn, spbn):   # cvlikelihood_test.py:209-230
        assert np.isclose(cvl.score(m), sum(cvl.local_score(m, v) for v in "abcd"))


@pytest.mark.parametrize("with_nulls", [False, True])
def test_holdout_local_scores(pbn, df, df_null, with_nulls):   # holdoutlikelihood_test.py:45-200
    data = df_null if with_nulls else df
    hl = pbn.HoldoutLikelihood(data, 0.2, SEED)
    ho = pbn.HoldOut(data, 0.2, SEED)
    assert hl.training_data().equals(ho.training_data()) and hl.test_data().equals(ho.test_data())
    split = lambda: [(ho.training_data(), ho.test_data())]
    gbn = pbn.GaussianNetwork(FULL)
    spbn = pbn.SemiparametricBN(FULL, [("a", pbn.CKDEType()), ("c", pbn.CKDEType())])
    for v, ev in [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]:
        assert np.isclose(hl.local_score(gbn, v, ev), recipe(pbn, "lg", split(), v, ev))
        assert np.isclose(hl.local_score(spbn, v, ev), recipe(pbn, kind_of(pbn, spbn.node_type(v)), split(), v, ev))
        assert hl.local_score(spbn, v) == hl.local_score(spbn, v, spbn.parents(v))
    assert np.isclose(hl.score(spbn), sum(hl.local_score(spbn, v) for v in "abcd"))
    for bad in (10, 0):
        with pytest.raises(ValueError, match="test_ratio must be a number"):
            pbn.HoldoutLikelihood(df, bad)
