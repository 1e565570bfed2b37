"""GPU tier: the argument / accessor / fit assertions of the reference's continuous-factor tests re-typed
(/root/reference/tests/factors/continuous/KDE_test.py:14-165, ProductKDE_test.py:15-196, CKDE_test.py:21-160).  The
log-likelihood values themselves are pinned in test_kde_gpu.py / test_ckde_gpu.py against the golden vectors; here the
expected numbers are scipy's gaussian_kde covariance and closed-form bandwidth rules, as in the reference tests."""
import numpy as np
import pyarrow as pa
import pytest
from scipy.stats import gaussian_kde

from helpers import CKDE_SETS, VARSETS, frame

pytestmark = pytest.mark.gpu
MISMATCH = "Data type of training and test datasets is different."


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def df(golden):
    return frame(golden["train500"])


@pytest.fixture(scope="module")
def df10k(golden):
    return frame(golden["train10k"])


def _nr_factor(s):
    return np.power(4 / (s.d + 2), 1 / (s.d + 4)) * s.scotts_factor()


def _with_nulls(df, seed=0, count=100):
    rng = np.random.RandomState(seed)
    out = df.copy()
    for c in "abcd":
        out.loc[out.index[rng.randint(0, len(df), size=count)], c] = np.nan
    return out


def _nr_diag(df, variables):   # ProductKDE_test.py:38-46
    cov = df[variables].cov().to_numpy()
    delta = np.linalg.inv(np.diag(np.diag(cov))).dot(cov)
    delta_inv = np.linalg.inv(delta)
    n, d = df.shape[0], len(variables)
    k = 4 * d * np.sqrt(np.linalg.det(delta)) / (2 * (delta_inv.dot(delta_inv)).trace() + delta_inv.trace() ** 2)
    return np.power(k / n, 2 / (d + 4)) * np.diag(cov)


def _scott_diag(df, variables):   # ProductKDE_test.py:48-53
    return np.power(df.shape[0], -2 / (len(variables) + 4)) * df[variables].var().to_numpy()


@pytest.mark.parametrize("cls", ["KDE", "ProductKDE"])
def test_check_type(pbn, df, cls):   # KDE_test.py:14-30, ProductKDE_test.py:15-31
    cpd = getattr(pbn, cls)(["a"])
    df_float = df.astype("float32")
    for train, test in ((df, df_float), (df_float, df)):
        cpd.fit(train)
        for fn in (cpd.logl, cpd.slogl):
            with pytest.raises(ValueError, match=MISMATCH):
                fn(test)


@pytest.mark.parametrize("cls", ["KDE", "ProductKDE"])
def test_variables(pbn, cls):   # KDE_test.py:32-35
    for variables in VARSETS:
        assert getattr(pbn, cls)(variables).variables() == variables


def test_kde_bandwidth(pbn, df):   # KDE_test.py:37-70
    df_float = df.astype("float32")
    for variables in VARSETS:
        for instances in (50, 150, 500):
            npdata = df.loc[:, variables].to_numpy()
            nr = gaussian_kde(npdata[:instances].T, bw_method=_nr_factor)
            scott = gaussian_kde(npdata[:instances].T)
            cpd = pbn.KDE(variables)
            cpd.fit(df.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, nr.covariance))
            cpd.fit(df_float.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, nr.covariance, atol=0.0005))
            cpd = pbn.KDE(variables, pbn.ScottsBandwidth())
            cpd.fit(df.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, scott.covariance))
            cpd.fit(df_float.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, scott.covariance, atol=0.0005))
    cpd = pbn.KDE(["a"])
    for data in (df, df_float):
        cpd.fit(data)
        cpd.bandwidth = [[1]]
        assert cpd.bandwidth == np.asarray([[1]])


def test_productkde_bandwidth(pbn, df):   # ProductKDE_test.py:55-88
    df_float = df.astype("float32")
    for variables in (["c", "a", "b"], ["d", "a", "b", "c"]):
        for instances in (50, 150, 500):
            cpd = pbn.ProductKDE(variables)
            cpd.fit(df.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, _nr_diag(df[:instances], variables)))
            cpd.fit(df_float.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, _nr_diag(df[:instances], variables), atol=0.0005))
            cpd = pbn.ProductKDE(variables, pbn.ScottsBandwidth())
            cpd.fit(df.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, _scott_diag(df[:instances], variables)))
            cpd.fit(df_float.iloc[:instances])
            assert np.all(np.isclose(cpd.bandwidth, _scott_diag(df[:instances], variables), atol=0.0005))
    cpd = pbn.ProductKDE(["a"])
    for data in (df, df_float):
        cpd.fit(data)
        cpd.bandwidth = [1]
        assert cpd.bandwidth == np.asarray([1])


def test_new_bandwidth_selector(pbn, df):   # KDE_test.py:72-91, ProductKDE_test.py:90-103
    class UnitaryBandwidth(pbn.BandwidthSelector):
        def __init__(self):
            pbn.BandwidthSelector.__init__(self)

        def bandwidth(self, df, variables):
            return np.eye(len(variables))

        def diag_bandwidth(self, df, variables):
            return np.ones((len(variables),))

    df_float = df.astype("float32")
    for variables in (["a"], ["a", "b", "c", "d"]):
        kde = pbn.KDE(variables, UnitaryBandwidth())
        pkde = pbn.ProductKDE(variables, UnitaryBandwidth())
        for data in (df, df_float):
            kde.fit(data)
            assert np.all(kde.bandwidth == np.eye(len(variables)))
            pkde.fit(data)
            assert np.all(pkde.bandwidth == np.ones(len(variables)))


@pytest.mark.parametrize("make", [lambda p: p.KDE(["a"]), lambda p: p.ProductKDE(["a"]), lambda p: p.CKDE("a", [])])
def test_data_type(pbn, df, make):   # KDE_test.py:93-104, ProductKDE_test.py:105-115, CKDE_test.py:31-41
    k = make(pbn)
    with pytest.raises(ValueError, match="factor not fitted"):
        k.data_type()
    k.fit(df)
    assert k.data_type() == pa.float64()
    k.fit(df.astype("float32"))
    assert k.data_type() == pa.float32()


def test_kde_fit(pbn, df):   # KDE_test.py:106-123, ProductKDE_test.py:117-134
    for variables in VARSETS:
        for data in (df, df.astype("float32")):
            for instances in (50, 150, 500):
                npdata = data.loc[:, variables].to_numpy()
                sk = gaussian_kde(npdata[:instances].T, bw_method=_nr_factor)
                cpd = pbn.KDE(variables)
                assert not cpd.fitted()
                cpd.fit(data.iloc[:instances])
                assert cpd.fitted()
                assert sk.n == cpd.num_instances() and sk.d == cpd.num_variables()
                pk = pbn.ProductKDE(variables)
                assert not pk.fitted()
                pk.fit(data.iloc[:instances])
                assert pk.fitted() and pk.num_instances() == instances and pk.num_variables() == len(variables)


def test_kde_fit_null(pbn, df):   # KDE_test.py:125-165, ProductKDE_test.py:136-196
    df_null = _with_nulls(df)
    for variables in VARSETS:
        for data in (df_null, df_null.astype("float32")):
            for instances in (50, 150, 500):
                npdata = data.loc[:, variables].to_numpy()[:instances]
                keep = npdata[~np.any(np.isnan(npdata), axis=1)]
                sk = gaussian_kde(keep.T, bw_method=_nr_factor)
                cpd = pbn.KDE(variables)
                cpd.fit(data.iloc[:instances])
                assert cpd.fitted() and cpd.num_instances() == sk.n and cpd.num_variables() == sk.d
                tol = {} if data is df_null else {"atol": 0.0005}
                assert np.all(np.isclose(cpd.bandwidth, sk.covariance, **tol))
                pk = pbn.ProductKDE(variables)
                pk.fit(data.iloc[:instances])
                assert pk.fitted() and pk.num_instances() == sk.n


def test_ckde_variable_evidence(pbn):   # CKDE_test.py:21-29
    for variable, evidence in CKDE_SETS:
        cpd = pbn.CKDE(variable, evidence)
        assert cpd.variable() == variable and cpd.evidence() == evidence


def test_ckde_members_are_references(pbn, df10k):   # CKDE_test.py:43-71
    for variable, evidence in CKDE_SETS:
        for data in (df10k, df10k.astype("float32")):
            cpd = pbn.CKDE(variable, evidence)
            cpd.fit(data)
            kde_joint = cpd.kde_joint
            kde_joint().bandwidth = np.eye(len(evidence) + 1)
            assert np.all(cpd.kde_joint().bandwidth == np.eye(len(evidence) + 1))
            if evidence:
                kde_marg = cpd.kde_marg
                assert kde_marg().fitted()
                kde_marg().bandwidth = np.eye(len(evidence))
                assert np.all(cpd.kde_marg().bandwidth == np.eye(len(evidence)))


def test_ckde_member_bandwidth_drives_logl(pbn, df, golden):
    """The reference's members are what logl evaluates (CKDE.hpp:387-433): after their bandwidths are replaced the
    factor must equal joint.logl - marg.logl of stand-alone KDEs carrying the same matrices."""
    test = frame(golden["test50"])
    cpd = pbn.CKDE("c", ["a", "b"])
    cpd.fit(df)
    base = cpd.logl(test)
    Hj = np.diag([0.3, 0.5, 0.7])
    Hm = np.array([[0.4, 0.1], [0.1, 0.6]])
    cpd.kde_joint().bandwidth = Hj
    cpd.kde_marg().bandwidth = Hm
    kj, km = pbn.KDE(["c", "a", "b"]), pbn.KDE(["a", "b"])
    kj.fit(df), km.fit(df)
    kj.bandwidth, km.bandwidth = Hj, Hm
    want = kj.logl(test) - km.logl(test)
    got = cpd.logl(test)
    assert not np.allclose(got, base)
    np.testing.assert_allclose(got, want, rtol=1e-12)
    np.testing.assert_allclose(cpd.slogl(test), want.sum(), rtol=1e-12)
    import pickle

    again = pickle.loads(pickle.dumps(cpd))
    np.testing.assert_allclose(again.logl(test), want, rtol=1e-12)
    cpd.fit(df)   # a new fit discards the replaced bandwidths
    np.testing.assert_allclose(cpd.logl(test), base, rtol=1e-12)


def test_ckde_fit(pbn, df10k):   # CKDE_test.py:73-97
    for variable, evidence in CKDE_SETS:
        variables = [variable] + evidence
        for data in (df10k, df10k.astype("float32")):
            for instances in (50, 1000, 10000):
                npdata = data.loc[:, variables].to_numpy()
                sk = gaussian_kde(npdata[:instances].T, bw_method=_nr_factor)
                cpd = pbn.CKDE(variable, evidence)
                assert not cpd.fitted()
                cpd.fit(data.iloc[:instances])
                assert cpd.fitted()
                tol = {}
                assert np.all(np.isclose(cpd.kde_joint().bandwidth, sk.covariance, **tol))
                if evidence:
                    assert np.all(np.isclose(cpd.kde_marg().bandwidth, sk.covariance[1:, 1:], **tol))
                assert cpd.num_instances() == instances


def test_ckde_fit_null(pbn, df10k):   # CKDE_test.py:99-160
    df_null = _with_nulls(df10k)
    for variable, evidence in CKDE_SETS:
        variables = [variable] + evidence
        for data in (df_null, df_null.astype("float32")):
            for instances in (50, 1000, 10000):
                npdata = data.loc[:, variables].to_numpy()[:instances]
                keep = npdata[~np.any(np.isnan(npdata), axis=1)]
                sk = gaussian_kde(keep.T, bw_method=_nr_factor)
                cpd = pbn.CKDE(variable, evidence)
                cpd.fit(data.iloc[:instances])
                assert cpd.fitted()
                tol = {}
                assert np.all(np.isclose(cpd.kde_joint().bandwidth, sk.covariance, **tol))
                if evidence:
                    assert np.all(np.isclose(cpd.kde_marg().bandwidth, sk.covariance[1:, 1:], **tol))
                assert cpd.num_instances() == sk.n
