"""Generate tests/golden/*.npz — run ONCE in the build container (needs /root/reference).

The reference module itself cannot be imported here, so these fixtures are produced the way
the reference's own tests pin this path: the deterministic table generator of
/root/reference/tests/helpers/util_test.py (imported, not copied) feeds the scipy / numpy recipes that the
reference tests compare `pybnesian` against.  Each recipe below cites the test it re-types.  The committed
.npz files hold only data: inputs and expected outputs.

    python tests/golden/gen_golden.py
"""
import os
import sys

import numpy as np
from scipy.stats import gaussian_kde, norm

sys.path.insert(0, "/root/reference/tests/helpers")
import util_test  # noqa: E402  (reference data generator: tests/helpers/util_test.py:5-20)

OUT = os.path.dirname(os.path.abspath(__file__))
VARSETS = [["a"], ["b", "a"], ["c", "a", "b"], ["d", "a", "b", "c"]]
CKDE_SETS = [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]


def nr_factor(s):  # KDE_test.py:43-44
    return np.power(4 / (s.d + 2), 1 / (s.d + 4)) * s.scotts_factor()


def factor_product_kernel(train_data):  # ProductKDE_test.py:180-196
    cov_data = np.atleast_2d(np.cov(train_data, rowvar=False, bias=False))
    delta = np.diag(np.reciprocal(np.diag(cov_data))).dot(cov_data)
    delta_inv = np.linalg.inv(delta)
    N, d = train_data.shape
    k = 4 * d * np.sqrt(np.linalg.det(delta)) / (2 * np.trace(np.dot(delta_inv, delta_inv)) + np.trace(delta_inv) ** 2)
    return (k / N) ** (1.0 / (d + 4.0))


def py_nr_bandwidth(df, variables):  # ProductKDE_test.py:38-46
    cov = df[variables].cov().to_numpy()
    delta = np.linalg.inv(np.diag(np.diag(cov))).dot(cov)
    delta_inv = np.linalg.inv(delta)
    N, d = df.shape[0], len(variables)
    k = 4 * d * np.sqrt(np.linalg.det(delta)) / (2 * (delta_inv.dot(delta_inv)).trace() + delta_inv.trace() ** 2)
    return np.power(k / N, 2 / (d + 4)) * np.diag(cov)


def py_scott_bandwidth(df, variables):  # ProductKDE_test.py:48-53
    var = df[variables].var().to_numpy()
    return np.power(df.shape[0], -2 / (len(variables) + 4)) * var


def product_logpdf(npdata, test_npdata):  # ProductKDE_test.py:199-220
    factor = factor_product_kernel(npdata)
    kde = gaussian_kde(npdata.T, bw_method=lambda g: factor * np.eye(npdata.shape[1], dtype=npdata.dtype))
    kde.cho_cov = np.linalg.cholesky(kde.covariance)
    kde.log_det = 2 * np.log(np.diag(kde.cho_cov * np.sqrt(2 * np.pi))).sum()
    return kde.logpdf(test_npdata.T)


def numpy_fit_mle_lg(data, variable, evidence):  # mle_test.py:10-27
    node = data.loc[:, [variable] + evidence].dropna()
    y = node.loc[:, variable].to_numpy()
    X = node.loc[:, evidence].to_numpy()
    N, d = X.shape
    A = np.column_stack((np.ones(N), X))
    beta, res, _, _ = np.linalg.lstsq(A, y, rcond=None)
    return beta, float(res[0] / (N - d - 1)) if res.size else float(np.sum((y - A @ beta) ** 2) / (N - d - 1))


def numpy_bic(data, variable, evidence):  # bic_test.py:10-30
    node = data.loc[:, [variable] + evidence].dropna()
    y = node.loc[:, variable]
    X = node.loc[:, evidence]
    N, d = X.shape
    A = np.column_stack((np.ones(N), X.to_numpy()))
    beta, res, _, _ = np.linalg.lstsq(A, y.to_numpy(), rcond=None)
    var = res / (N - d - 1)
    means = beta[0] + np.sum(beta[1:] * X, axis=1)
    return float((norm.logpdf(y, means, np.sqrt(var))).sum() - np.log(N) * 0.5 * (d + 2))


def null_test_df(test_df):  # KDE_test.py:236-248 (same seeds and draw order)
    n = test_df.shape[0]
    np.random.seed(0)
    nulls = {c: np.random.randint(0, n, size=10) for c in ["a", "b", "c", "d"]}
    out = test_df.copy()
    for c in ["a", "b", "c", "d"]:
        out.loc[out.index[nulls[c]], c] = np.nan
    return out


def ckde_cdf_recipe(joint, train, test):  # CKDE_test.py:181-222 (column 0 = variable, 1.. = evidence)
    """P(X <= x | e) of a Gaussian-kernel CKDE: mixture of normal cdfs at the per-instance conditional means,
    weighted by the evidence kernel."""
    from scipy.stats import multivariate_normal as mvn
    H = np.atleast_2d(joint.covariance)
    if train.shape[1] == 1:
        return norm.cdf(test[:, [0]], train[:, 0][None, :], np.sqrt(H[0, 0])).mean(axis=1)
    Hee_inv = np.linalg.inv(H[1:, 1:])
    cond_sd = np.sqrt(H[0, 0] - H[0, 1:] @ Hee_inv @ H[1:, 0])
    out = np.empty(test.shape[0])
    for i in range(test.shape[0]):
        w = np.exp(mvn.logpdf(train[:, 1:], mean=test[i, 1:], cov=H[1:, 1:]))
        cmean = train[:, 0] + (test[i, 1:] - train[:, 1:]) @ (Hee_inv @ H[1:, 0])
        out[i] = w @ norm.cdf(test[i, 0], cmean, cond_sd) / w.sum()
    return out


def main():
    g = {}
    train500 = util_test.generate_normal_data(500, seed=0)    # KDE_test.py:10-11
    train10k = util_test.generate_normal_data(10000, seed=0)  # CKDE_test.py:13-16, bic_test.py:6-8
    test50 = util_test.generate_normal_data(50, seed=1)       # KDE_test.py:185
    test50_null = null_test_df(test50)
    cols = ["a", "b", "c", "d"]
    g["train500"] = train500[cols].to_numpy()
    g["train10k"] = train10k[cols].to_numpy()
    g["test50"] = test50[cols].to_numpy()
    g["test50_null"] = test50_null[cols].to_numpy()

    for variables in VARSETS:
        key = "".join(variables)
        # KDE bandwidth (KDE_test.py:37-59), incl. Scott
        for n in (50, 500):
            nd = train500.loc[:, variables].to_numpy()[:n]
            g[f"kde_bw_nr_{key}_{n}"] = gaussian_kde(nd.T, bw_method=nr_factor).covariance
            g[f"kde_bw_scott_{key}_{n}"] = gaussian_kde(nd.T).covariance
        # KDE logl / slogl on 500 train, 50 test (KDE_test.py:167-184, 267-285), fp64 and fp32 inputs
        for tag, tr, te in (("f64", train500, test50), ("f32", train500.astype("float32"), test50.astype("float32"))):
            nd = tr.loc[:, variables].to_numpy()
            kde = gaussian_kde(nd.T, bw_method=nr_factor)
            g[f"kde_logl_{key}_{tag}"] = kde.logpdf(te.loc[:, variables].to_numpy().T)
        # null rows (KDE_test.py:205-233)
        tn = test50_null.loc[:, variables].to_numpy()
        res = np.full(tn.shape[0], np.nan)
        ok = ~np.any(np.isnan(tn), axis=1)
        res[ok] = gaussian_kde(train500.loc[:, variables].to_numpy().T, bw_method=nr_factor).logpdf(tn[ok].T)
        g[f"kde_logl_null_{key}_f64"] = res
        # ProductKDE (ProductKDE_test.py:38-71, 199-231)
        for n in (50, 150, 500):
            g[f"pkde_bw_nr_{key}_{n}"] = py_nr_bandwidth(train500[:n], variables)
            g[f"pkde_bw_scott_{key}_{n}"] = py_scott_bandwidth(train500[:n], variables)
        g[f"pkde_logl_{key}_f64"] = product_logpdf(train500.loc[:, variables].to_numpy(), test50.loc[:, variables].to_numpy())

    # CKDE (CKDE_test.py:146-179, 221-254, 316-349): 10000 and 10 training rows
    for variable, evidence in CKDE_SETS:
        key = variable + "_" + "".join(evidence)
        variables = [variable] + evidence
        for tag, tr in (("10k", train10k), ("10", train10k.iloc[:10])):
            # util_test.generate_normal_data(10, 0) == first 10 rows only for column a; regenerate to be exact
            if tag == "10":
                tr = util_test.generate_normal_data(10, seed=0)
                g["train10"] = tr[cols].to_numpy()
            joint = gaussian_kde(tr.loc[:, variables].to_numpy().T, bw_method=nr_factor)
            lj = joint.logpdf(test50.loc[:, variables].to_numpy().T)
            if evidence:
                marg = gaussian_kde(tr.loc[:, evidence].to_numpy().T, bw_method=joint.covariance_factor())
                lj = lj - marg.logpdf(test50.loc[:, evidence].to_numpy().T)
            g[f"ckde_logl_{key}_{tag}"] = lj
            g[f"ckde_bw_{key}_{tag}"] = joint.covariance
            # CKDE.cdf (CKDE_test.py:256-314)
            g[f"ckde_cdf_{key}_{tag}"] = ckde_cdf_recipe(joint, tr.loc[:, variables].to_numpy(), test50.loc[:, variables].to_numpy())

    # LinearGaussianCPD MLE + BIC (mle_test.py:10-56, bic_test.py:10-43) on the 10k table
    for variable, evidence in CKDE_SETS:
        key = variable + "_" + "".join(evidence)
        beta, var = numpy_fit_mle_lg(train10k, variable, evidence)
        g[f"lg_beta_{key}"] = beta
        g[f"lg_var_{key}"] = np.float64(var)
        g[f"bic_{key}"] = np.float64(numpy_bic(train10k, variable, evidence))
        # LinearGaussianCPD logl on the test table (LinearGaussianCPD_test.py:76-120)
        te = test50
        means = beta[0] + (te.loc[:, evidence].to_numpy() @ beta[1:] if evidence else 0.0)
        g[f"lg_logl_{key}"] = norm.logpdf(te[variable].to_numpy(), means, np.sqrt(var))

    # libstdc++ std::shuffle known answer (crossvalidation_adaptator.hpp:25-40)
    g["shuffle12_seed0"] = np.array([0, 2, 1, 5, 9, 11, 4, 7, 6, 10, 3, 8], dtype=np.int32)

    np.savez_compressed(os.path.join(OUT, "reference_recipes.npz"), **g)
    print("wrote", len(g), "arrays")
    # SURVEY.md Appendix B anchors
    print("anchor slogl KDE(a)      ", g["kde_logl_a_f64"].sum(), "expected -34.73903536927675")
    print("anchor slogl KDE(dabc)   ", g["kde_logl_dabc_f64"].sum(), "expected -225.92553570505632")
    print("anchor slogl CKDE d|abc  ", g["ckde_logl_d_abc_10k"].sum(), "expected -37.16862836312814")
    print("anchor BIC d|abc         ", g["bic_d_abc"], "expected -7365.898169173852")


if __name__ == "__main__":
    main()
