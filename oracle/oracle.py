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


def ckde_logl(train, H, test):
    """Column 0 = variable, columns 1.. = evidence."""
    return _logl(lib().oracle_ckde_logl_f32, lib().oracle_ckde_logl_f64, train, H, test)


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
