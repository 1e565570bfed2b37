"""CPU BASELINE — TEST / BENCH INFRASTRUCTURE ONLY (see pbn_baseline.cpp).  ctypes wrapper around
oracle/_build/libpbn_baseline.so, the tuned (whitened, blocked, vectorised, OpenMP) CPU form of the KDE log-likelihoods that
bench.py TIMES beside the device path.  The checker is oracle.py / pbn_oracle.cpp, never this file.

The library is compiled with -march=native, so it belongs to the CPU it was built on: `lib()` rebuilds it whenever the stamp
beside it names another CPU (a copy built in the development container must not be loaded on the GPU box's host).
"""
import ctypes as C
import hashlib
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_build", "libpbn_baseline.so")
_STAMP = os.path.join(_HERE, "_build", "libpbn_baseline.cpu")
_lib = None


def _cpu_id():
    model, flags = "", ""
    try:
        with open("/proc/cpuinfo") as f:
            for line in f:
                if line.startswith("model name") and not model:
                    model = line.split(":", 1)[1].strip()
                elif line.startswith("flags") and not flags:
                    flags = line.split(":", 1)[1].strip()
                if model and flags:
                    break
    except OSError:
        pass
    with open(os.path.join(_HERE, "pbn_baseline.cpp"), "rb") as f:
        src = hashlib.sha1(f.read()).hexdigest()
    return f"{model}|{hashlib.sha1(flags.encode()).hexdigest()}|{src}"


def cpu_model():
    return _cpu_id().split("|", 1)[0]


def lib():
    global _lib
    if _lib is None:
        want = _cpu_id()
        have = None
        if os.path.exists(_PATH) and os.path.exists(_STAMP):
            with open(_STAMP) as f:
                have = f.read()
        if have != want:
            if os.path.exists(_PATH):
                os.remove(_PATH)
            subprocess.check_call(["make", "-C", _HERE, "baseline"], stdout=subprocess.DEVNULL)
            with open(_STAMP, "w") as f:
                f.write(want)
        _lib = C.CDLL(_PATH)
    return _lib


def num_threads():
    return lib().baseline_num_threads()


def set_num_threads(n):
    lib().baseline_set_num_threads(int(n))


def _call(fn, train, bw, test):
    tr = np.asfortranarray(np.asarray(train, dtype=np.float64))
    te = np.asfortranarray(np.asarray(test, dtype=np.float64))
    if tr.ndim == 1:
        tr, te = tr[:, None], te[:, None]
    bw = np.asfortranarray(bw, dtype=np.float64)
    out = np.zeros(te.shape[0])
    rc = fn(tr.ctypes.data_as(C.c_void_p), C.c_int64(tr.shape[0]), tr.shape[1], bw.ctypes.data_as(C.c_void_p),
            te.ctypes.data_as(C.c_void_p), C.c_int64(te.shape[0]), out.ctypes.data_as(C.c_void_p))
    if rc:
        raise ValueError("singular bandwidth" if rc == 2 else "too many dimensions")
    return out


def kde_logl(train, H, test):
    return _call(lib().baseline_kde_logl_f64, train, H, test)


def product_kde_logl(train, h, test):
    return _call(lib().baseline_product_kde_logl_f64, train, h, test)


def ckde_logl(train, H, test):
    """Column 0 = variable, columns 1.. = evidence."""
    return _call(lib().baseline_ckde_logl_f64, train, H, test)
