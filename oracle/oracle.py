"""ORACLE — TEST INFRASTRUCTURE ONLY (see pbn_oracle.cpp).  ctypes wrapper around oracle/_build/libpbn_oracle.so.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libpbn_oracle.so")
_lib = None


def build():
    import subprocess

    subprocess.check_call(["make", "-C", _HERE], stdout=subprocess.DEVNULL)


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_PATH):
            build()
        _lib = C.CDLL(_PATH)
        _lib.oracle_holdout_test_rows.restype = C.c_int64
        for f in ("oracle_lg_fit_f64", "oracle_lg_fit_f32", "oracle_bic_lg", "oracle_bge_f64", "oracle_bge_cached"):
            getattr(_lib, f).restype = C.c_double
    return _lib


def num_threads():
    return lib().oracle_num_threads()


def set_num_threads(n):
    lib().oracle_set_num_threads(int(n))


def _colmajor(a, dtype):
    a = np.asarray(a, dtype=dtype)
    if a.ndim == 1:
        a = a[:, None]
    return np.asfortranarray(a)


def _dp(a):
    return a.ctypes.data_as(C.c_void_p)


def cov(data):
    """DataFrame::cov in the dtype of `data` (n x d)."""
    data = np.asarray(data)
    f32 = data.dtype == np.float32
    a = _colmajor(data, np.float32 if f32 else np.float64)
    n, d = a.shape
    cols = (C.c_void_p * d)(*[a[:, j].ctypes.data for j in range(d)])
    out = np.zeros((d, d), order="F")
    means = np.zeros(d)
    fn = lib().oracle_cov_f32 if f32 else lib().oracle_cov_f64
    fn(cols, C.c_int64(n), d, _dp(out), _dp(means))
    return out, means


def bandwidth(selector, kind, covm, n):
    covm = np.asfortranarray(covm, dtype=np.float64)
    d = covm.shape[0]
    out = np.zeros((d, d), order="F") if kind == 0 else np.zeros(d)
    rc = lib().oracle_bandwidth(selector, kind, _dp(covm), d, C.c_int64(n), _dp(out))
    if rc:
        raise ValueError("singular")
    return out


def _logl(fn32, fn64, train, bw, test):
    train = np.asarray(train)
    f32 = train.dtype == np.float32
    dt = np.float32 if f32 else np.float64
    tr, te = _colmajor(train, dt), _colmajor(np.asarray(test), dt)
    bw = np.asfortranarray(bw, dtype=np.float64)
    out = np.zeros(te.shape[0])
    fn = fn32 if f32 else fn64
    rc = fn(_dp(tr), C.c_int64(tr.shape[0]), tr.shape[1], _dp(bw), _dp(te), C.c_int64(te.shape[0]), _dp(out))
    if rc:
        raise ValueError("singular bandwidth")
    return out


def kde_logl(train, H, test):
    return _logl(lib().oracle_kde_logl_f32, lib().oracle_kde_logl_f64, train, H, test)


def product_kde_logl(train, h, test):
    return _logl(lib().oracle_product_kde_logl_f32, lib().oracle_product_kde_logl_f64, train, h, test)


# TIMING ONLY (bench.py's CPU baselines of the hill-climb legs): when set, the CKDE log-likelihoods inside the likelihood scores below
# come from the tuned port (oracle/baseline.py: whitened, blocked, vectorised; fp64 data) instead of the reference-arithmetic checker.
# Tests never set it.
_TIMED_CKDE = None


def use_tuned_ckde(on):
    global _TIMED_CKDE
    if on:
        from . import baseline

        _TIMED_CKDE = baseline.ckde_logl
    else:
        _TIMED_CKDE = None


def ckde_logl(train, H, test):
    """Column 0 = variable, columns 1.. = evidence."""
    return _logl(lib().oracle_ckde_logl_f32, lib().oracle_ckde_logl_f64, train, H, test)


def ckde_cdf(train, H, test):
    """CKDE::cdf restated step by step in the table's dtype (column 0 = variable, columns 1.. = evidence).

    factors/continuous/CKDE.hpp:561-594 (_cdf_univariate) and :596-735 (_cdf_multivariate): the N x m weight
    matrix W = exp(logl_mat) of the evidence KDE, the conditional means mu[t, q] = x_t + b.(e_q - e_t) with
    b = H12 H22^-1 (kde/opencl_kernels/KDE.cl.src:376-430), 0.5 erfc((mu - x_q) / (sigma_c sqrt 2)) (:447-456),
    the element-wise product, column sums and the final division."""
    from scipy.special import erfc
    dt = np.float32 if train.dtype == np.float32 else np.float64
    train = np.asarray(train, dtype=dt)
    test = np.asarray(test, dtype=dt)
    H = np.atleast_2d(np.asarray(H, dtype=np.float64))
    N, d = train.shape
    sqrt1_2 = dt(np.sqrt(0.5))
    if d == 1:
        inv_std = dt(1.0 / np.sqrt(H[0, 0]))
        inv_n = dt(1.0 / N)
        mat = inv_n * (dt(0.5) * erfc(sqrt1_2 * inv_std * -(test[None, :, 0] - train[:, 0, None])))
        return mat.astype(dt).sum(axis=0, dtype=dt).astype(np.float64)
    p = d - 1
    L = np.linalg.cholesky(H[1:, 1:])
    Linv = np.linalg.solve(L, np.eye(p))
    R = Linv @ H[1:, 0]
    cond_var = H[0, 0] - R @ R
    transform = (R @ Linv).astype(dt)
    inv_std = dt(1.0 / np.sqrt(cond_var))
    # logl_mat of the evidence KDE without the 1/N term (CKDE.hpp:653 new_lognorm_marg)
    lognorm = dt(-np.log(np.diag(L)).sum() - 0.5 * p * np.log(2 * np.pi))
    Ld = L.astype(dt)
    diff = (test[None, :, 1:] - train[:, None, 1:]).reshape(-1, p).T       # p x (N m)
    import scipy.linalg as sla
    z = sla.solve_triangular(Ld, diff, lower=True, check_finite=False).astype(dt)
    W = np.exp((dt(-0.5) * (z * z).sum(axis=0, dtype=dt) + lognorm).astype(dt)).reshape(N, -1)
    mu = train[:, 0, None] + ((test[None, :, 1:] - train[:, None, 1:]) * transform).sum(axis=2, dtype=dt)
    cdf = (dt(0.5) * erfc(sqrt1_2 * inv_std * (mu - test[None, :, 0]))).astype(dt)
    num = (cdf * W).sum(axis=0, dtype=dt)
    return (num / W.sum(axis=0, dtype=dt)).astype(np.float64)


def ckde_sample(train, H, ev, n, seed, stream_n=0):
    """(values, instance indices) of CKDE::sample; train column 0 = variable; ev: n x p evidence rows (or None)."""
    f32 = train.dtype == np.float32
    dt = np.float32 if f32 else np.float64
    ct = C.c_float if f32 else C.c_double
    tr = _colmajor(train, dt)
    N, d = train.shape
    e = _colmajor(ev, dt) if ev is not None and d > 1 else np.zeros(1, dtype=dt)
    Hm = np.asfortranarray(np.atleast_2d(H), dtype=np.float64)
    out = np.zeros(n, dtype=dt)
    idx = np.zeros(n, dtype=np.int32)
    fn = lib().oracle_ckde_sample_f32 if f32 else lib().oracle_ckde_sample_f64
    fn.restype = C.c_int
    rc = fn(tr.ctypes.data_as(C.POINTER(ct)), C.c_int64(N), C.c_int(d), _dp(Hm), e.ctypes.data_as(C.POINTER(ct)), C.c_int64(n),
            C.c_int64(max(n, stream_n)), C.c_uint32(seed), out.ctypes.data_as(C.POINTER(ct)), idx.ctypes.data_as(C.POINTER(C.c_int32)))
    if rc:
        raise ValueError("singular bandwidth")
    return out, idx


def lg_sample(n, beta, variance, seed, ev=None):
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    p = beta.size - 1
    e = np.asfortranarray(ev, dtype=np.float64) if p else np.zeros(1)
    out = np.zeros(n)
    lib().oracle_lg_sample(C.c_int64(n), _dp(beta), C.c_int(p), C.c_double(variance), C.c_uint32(seed), _dp(e), _dp(out))
    return out


def discrete_sample(n, logprob, card, seed, parent_offset=None):
    lp = np.ascontiguousarray(logprob, dtype=np.float64)
    out = np.zeros(n, dtype=np.int32)
    po = None if parent_offset is None else np.ascontiguousarray(parent_offset, dtype=np.int32)
    lib().oracle_discrete_sample(C.c_int64(n), _dp(lp), C.c_int(card), C.c_int64(lp.size),
                                 po.ctypes.data_as(C.POINTER(C.c_int32)) if po is not None else None, C.c_uint32(seed),
                                 out.ctypes.data_as(C.POINTER(C.c_int32)))
    return out


def ucv_score(data, bandwidth):
    """N * UCV(H) (kde/UCV.cpp:226-360 with the pair terms of kde/opencl_kernels/KDE.cl.src:470-574): all i < j pairs,
    in the table's dtype for the kernel terms.  bandwidth: d x d matrix or vector of d variances."""
    dt = np.float32 if data.dtype == np.float32 else np.float64
    X = np.asarray(data, dtype=dt)
    N, d = X.shape
    bw = np.asarray(bandwidth, dtype=np.float64)
    if bw.ndim == 1:
        L = np.diag(np.sqrt(bw))
    else:
        L = np.linalg.cholesky(bw)
    lognorm_H = dt(-np.log(np.diag(L)).sum() - 0.5 * d * np.log(2 * np.pi))
    lognorm_2H = dt(lognorm_H - 0.5 * d * np.log(2.0))
    Ld = L.astype(dt)
    import scipy.linalg as sla

    s2h = sh = 0.0
    for i in range(1, N):
        diff = (X[i][None, :] - X[:i]).T                                   # pairs (i, j < i)
        z = sla.solve_triangular(Ld, diff, lower=True, check_finite=False).astype(dt)
        dist = (z * z).sum(axis=0, dtype=dt)
        s2h += float(np.exp((dt(-0.25) * dist + lognorm_2H).astype(dt)).sum(dtype=np.float64))
        sh += float(np.exp((dt(-0.5) * dist + lognorm_H).astype(dt)).sum(dtype=np.float64))
    return float(np.exp(np.float64(lognorm_2H)) + 2 * s2h / N - 4 * sh / (N - 1))


def shuffled_indices(n, seed):
    idx = np.arange(n, dtype=np.int32)
    lib().oracle_shuffle(_dp(idx), C.c_int64(n), C.c_uint32(seed))
    return idx


def shuffle_inplace(idx, seed):
    idx = np.ascontiguousarray(idx, dtype=np.int32)
    lib().oracle_shuffle(_dp(idx), C.c_int64(idx.size), C.c_uint32(seed))
    return idx


def cv_limits(n, k):
    lim = np.zeros(k + 1, dtype=np.int32)
    if lib().oracle_cv_limits(C.c_int64(n), k, _dp(lim)):
        raise ValueError("Cannot split")
    return lim


def holdout_test_rows(n, ratio):
    return int(lib().oracle_holdout_test_rows(C.c_int64(n), C.c_double(ratio)))


def _colptrs(data):
    """(keepalive, void*[d], n, d, is_f32) for an n x d array (column 0 first)."""
    data = np.asarray(data)
    f32 = data.dtype == np.float32
    a = _colmajor(data, np.float32 if f32 else np.float64)
    n, d = a.shape
    ptrs = (C.c_void_p * d)(*[a[:, j].ctypes.data for j in range(d)])
    return a, ptrs, n, d, f32


def lg_fit(data):
    """MLE<LinearGaussianCPD>::estimate; data = [y, x1..xp] columns.  Returns (beta, variance)."""
    a, ptrs, n, d, f32 = _colptrs(data)
    beta = np.zeros(d)
    fn = lib().oracle_lg_fit_f32 if f32 else lib().oracle_lg_fit_f64
    var = fn(ptrs, C.c_int64(n), d - 1, _dp(beta))
    return beta, var


def lg_logl(data, beta, variance):
    a, ptrs, n, d, f32 = _colptrs(data)
    out = np.zeros(n)
    beta = np.ascontiguousarray(beta, dtype=np.float64)
    fn = lib().oracle_lg_logl_f32 if f32 else lib().oracle_lg_logl_f64
    fn(ptrs, C.c_int64(n), d - 1, _dp(beta), C.c_double(variance), _dp(out))
    return out


def bic_lg(data):
    beta, var = lg_fit(data)
    n, d = np.asarray(data).shape if np.asarray(data).ndim == 2 else (len(data), 1)
    return lib().oracle_bic_lg(C.c_int64(n), d - 1, C.c_double(var))


def bge(data, total_nodes, iss_mu=1.0, iss_w=None, nu=None):
    """BGe local score; data = [variable, parents...] fp64 columns; nu defaults to the sample means."""
    a, ptrs, n, d, f32 = _colptrs(np.asarray(data, dtype=np.float64))
    if iss_w is None:
        iss_w = total_nodes + 2
    nu = a.mean(axis=0) if nu is None else np.asarray(nu, dtype=np.float64)
    _, means = cov(a)
    if nu is None:
        nu = means
    nu = np.ascontiguousarray(nu, dtype=np.float64)
    return lib().oracle_bge_f64(ptrs, C.c_int64(n), d - 1, int(total_nodes), C.c_double(iss_mu), C.c_double(iss_w), _dp(nu))


def bge_cached(cov_all, means_all, n_rows, sel, total_nodes, iss_mu=1.0, iss_w=None, nu_all=None):
    """BGe local score of sel = [variable, parents...] from the whole-table moments `cov(table)` the reference's constructor
    caches (bge.hpp:52-68); equal to bge(table[:, sel], total_nodes) bit for bit, without another pass over the rows."""
    cov_all = np.asfortranarray(cov_all, dtype=np.float64)
    means_all = np.ascontiguousarray(means_all, dtype=np.float64)
    if iss_w is None:
        iss_w = total_nodes + 2
    nu_all = means_all if nu_all is None else np.ascontiguousarray(nu_all, dtype=np.float64)
    sel = np.ascontiguousarray(sel, dtype=np.int32)
    return lib().oracle_bge_cached(_dp(cov_all), _dp(means_all), cov_all.shape[0], C.c_int64(n_rows), _dp(sel), len(sel) - 1,
                                   int(total_nodes), C.c_double(iss_mu), C.c_double(iss_w), _dp(nu_all))


def cv_folds(n, k, seed):
    """[(train_idx, test_idx)] exactly as CrossValidation::generate_cv_pair (crossvalidation_adaptator.cpp:5-31)."""
    idx = shuffled_indices(n, seed)
    lim = cv_limits(n, k)
    return [(np.concatenate([idx[: lim[f]], idx[lim[f + 1]:]]), idx[lim[f]: lim[f + 1]]) for f in range(k)]


def holdout_split(n, ratio, seed):
    idx = shuffled_indices(n, seed)
    test_rows = holdout_test_rows(n, ratio)
    return idx[: n - test_rows], idx[n - test_rows:]


def nr_bandwidth(data):
    c, _ = cov(data)
    return bandwidth(0, 0, c, np.asarray(data).shape[0])


def cv_likelihood(data, node_type, k, seed):
    """CVLikelihood::local_score (cv_likelihood.cpp:11-25) for columns [variable, parents...]."""
    data = np.asarray(data)
    total = 0.0
    for tr, te in cv_folds(data.shape[0], k, seed):
        total += _fit_slogl(data[tr], data[te], node_type)
    return total


def holdout_likelihood(data, node_type, ratio, seed):
    data = np.asarray(data)
    tr, te = holdout_split(data.shape[0], ratio, seed)
    return _fit_slogl(data[tr], data[te], node_type)


def validated_cv_likelihood(data, node_type, ratio, k, seed):
    """ValidatedLikelihood::local_score: CV(k, seed) over the hold-out training part (validated_likelihood.hpp:14-60)."""
    data = np.asarray(data)
    tr, _ = holdout_split(data.shape[0], ratio, seed)
    return cv_likelihood(data[tr], node_type, k, seed)


def _fit_slogl(train, test, node_type):
    if node_type == "lg":
        beta, var = lg_fit(train)
        return float(lg_logl(test, beta, var).sum())
    H = nr_bandwidth(train)
    if _TIMED_CKDE is not None and np.asarray(train).dtype == np.float64:
        return float(_TIMED_CKDE(train, H, test).sum())
    return float(ckde_logl(train, H, test).sum())


# ---- hybrid (discrete parents / discrete variables) ----------------------------------------------------------------
# Restates factors/discrete/discrete_indices.cpp:93-204, DiscreteAdaptator.hpp:201-348, bic.cpp:29-96,
# mle_DiscreteFactor.cpp:5-41 and DiscreteFactor.cpp:133-171 with per-slice calls into the C routines above.
MACHINE_TOL = 1.4901161193847656e-08


def config_index(codes, cards, n=None):
    """codes: list of int arrays (evidence order); stride_0 = 1, stride_i = stride_{i-1} * card_{i-1}."""
    idx = np.zeros(len(codes[0]) if codes else (n or 0), dtype=np.int64)
    stride, n = 1, 1
    for c, k in zip(codes, cards):
        idx += np.asarray(c, dtype=np.int64) * stride
        stride *= k
        n *= k
    return idx, n


def bic_clg(cont, dcodes, dcards):
    """BIC::bic_clg (bic.cpp:29-64); cont = [variable, continuous parents...] (N x d)."""
    cont = np.asarray(cont)
    cfg, ncfg = config_index(dcodes, dcards, cont.shape[0])
    p = cont.shape[1] - 1
    loglik = 0.0
    for c in range(ncfg):
        rows = np.nonzero(cfg == c)[0]
        if rows.size == 0:
            continue
        beta, var = lg_fit(cont[rows])
        if var < MACHINE_TOL or np.isinf(var):
            return -np.inf
        n = rows.size
        loglik += 0.5 * (1 + p - n) - 0.5 * n * np.log(2 * np.pi) - n * 0.5 * np.log(var)
    return loglik - np.log(cont.shape[0]) * 0.5 * ncfg * (p + 2)


def _joint_counts(vcodes, card0, pcodes, pcards, rows=None):
    idx, n = config_index([vcodes] + list(pcodes), [card0] + list(pcards))
    if rows is not None:
        idx = idx[rows]
    return np.bincount(idx, minlength=n).astype(np.int64)


def bic_discrete(vcodes, card0, pcodes, pcards):
    """BIC::bic_discrete (bic.cpp:66-96)."""
    jc = _joint_counts(vcodes, card0, pcodes, pcards).reshape(-1, card0)
    ll = 0.0
    for row in jc:
        s = row.sum()
        if s > 0:
            nz = row[row > 0].astype(np.float64)
            ll += float(np.sum(nz * np.log(nz * (1.0 / s))))
    return ll - np.log(float(jc.sum())) * 0.5 * (card0 - 1) * jc.shape[0]


def discrete_fit_slogl(vcodes, card0, pcodes, pcards, train, test):
    """MLE<DiscreteFactor> on train rows (uniform for unseen configurations) + DiscreteFactor::slogl on test rows."""
    tr = _joint_counts(vcodes, card0, pcodes, pcards, train).reshape(-1, card0)
    te = _joint_counts(vcodes, card0, pcodes, pcards, test).reshape(-1, card0)
    res = 0.0
    with np.errstate(divide="ignore"):
        for k in range(tr.shape[0]):
            s = tr[k].sum()
            lp = np.full(card0, np.log(1.0 / card0)) if s == 0 else np.log(tr[k].astype(np.float64)) - np.log(float(s))
            for i in range(card0):
                if te[k, i]:
                    res += te[k, i] * lp[i]
    return float(res)


def adaptator_fit_slogl(cont, dcodes, dcards, train, test, node_type, arithmetic=None):
    """DiscreteAdaptator<LinearGaussianCPD | CKDE>::fit on train rows + slogl on test rows.
    The arithmetic follows the dtype of `cont`, as the reference's does the Arrow type of the columns: float32 data -> covariance in
    float (dataset.hpp:429-481 through NormalReferenceRule.hpp:124-133: `df.cov<ArrowType>`, then `k * cov.cast<double>()`), distances,
    exponentials, sums and logl in float (KDE.hpp:466-470, KDE.cl.src with @dt@ = float; pbn_oracle.cpp kde_full<float>, cov_T<float>).
    arithmetic = "float32" / "float64" casts `cont` first - "float64" on float data is the TRUTH for those values, "float32" what the
    reference computes."""
    cont = np.asarray(cont)
    if arithmetic is not None:
        cont = cont.astype({"float32": np.float32, "float64": np.float64}[arithmetic])
    cfg, ncfg = config_index(dcodes, dcards, cont.shape[0])
    total = 0.0
    for c in range(ncfg):
        tr = train[cfg[train] == c]
        te = test[cfg[test] == c]
        if tr.size == 0:
            continue
        if node_type == "lg":
            beta, var = lg_fit(cont[tr])
            if var < MACHINE_TOL or np.isinf(var):
                continue  # LinearGaussianFitter -> no factor
            if te.size:
                total += float(lg_logl(cont[te], beta, var).sum())
        else:
            d = cont.shape[1]
            if tr.size <= d:
                continue  # SingularCovarianceData swallowed by CKDEFitter
            c_, _ = cov(cont[tr])
            ev = np.linalg.eigvalsh(c_)
            if ev.min() < ev.max() * d * np.finfo(np.float64).eps:
                continue  # is_psd fails -> SingularCovarianceData
            H = bandwidth(0, 0, c_, tr.size)
            if te.size:
                total += float((_TIMED_CKDE if (_TIMED_CKDE is not None and cont.dtype == np.float64) else ckde_logl)(cont[tr], H, cont[te]).sum())
    return total


def hybrid_cv(fn, n, k, seed):
    return sum(fn(tr, te) for tr, te in cv_folds(n, k, seed))


def hybrid_holdout(fn, n, ratio, seed):
    tr, te = holdout_split(n, ratio, seed)
    return fn(tr, te)


def kmi(data, k, seed=0, shuffle_neighbors=5, samples=0):
    """KMutualInformation on the columns of `data` = [x, y, z...] (N x dims): (mi, pvalue); pvalue is None when samples == 0."""
    a = np.asfortranarray(np.asarray(data, dtype=np.float64))
    n, d = a.shape
    cols = (C.c_void_p * d)(*[a[:, j].ctypes.data for j in range(d)])
    mi, pv = C.c_double(0.0), C.c_double(0.0)
    lib().oracle_kmi(cols, C.c_int(d), C.c_int64(n), C.c_int(k), C.c_uint32(seed), C.c_int(shuffle_neighbors), C.c_int(max(samples, 1)),
                     C.byref(mi), C.byref(pv) if samples else None)
    return mi.value, (pv.value if samples else None)
