"""GPU tier: a plain-C program (tests/c/capi_demo.c) drives the C ABI directly — no Python, no torch — and
must agree with the Python mirror on the same data."""
import os
import subprocess

import numpy as np
import pandas as pd
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
pytestmark = pytest.mark.gpu


def _xorshift_stream(n):
    x = 88172645463325252
    mask = (1 << 64) - 1
    out = np.empty(n)
    for i in range(n):
        x ^= (x << 13) & mask
        x ^= x >> 7
        x ^= (x << 17) & mask
        out[i] = (x >> 11) / 9007199254740992.0
    return out


def _table(u, n):
    u = u.reshape(n, 6)
    a = u[:, 0] + u[:, 1] + u[:, 2] - 1.5
    b = u[:, 3] + u[:, 4] - 1.0
    c = u[:, 5] - 0.5
    return pd.DataFrame({"a": a, "b": 0.6 * a + b, "c": a - 0.4 * b + c})


def test_c_program_matches_python(tmp_path):
    import pybnesian_amd as pbn

    exe = str(tmp_path / "capi_demo")
    subprocess.check_call(["gcc", "-O2", "-I" + os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "c", "capi_demo.c"),
                           "-L" + os.path.join(ROOT, "pybnesian_amd"), "-lpbn_hip", "-Wl,-rpath," + os.path.join(ROOT, "pybnesian_amd"),
                           "-lm", "-o", exe])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stderr
    vals = dict(line.split() for line in out.stdout.strip().splitlines())
    n, m = 5000, 700
    u = _xorshift_stream(6 * (n + m))
    train, test = _table(u[: 6 * n], n), _table(u[6 * n:], m)
    kde = pbn.KDE(["a", "b", "c"])
    kde.fit(train)
    assert abs(float(vals["kde_slogl"]) - kde.slogl(test)) <= 1e-9 * abs(kde.slogl(test))
    ckde = pbn.CKDE("a", ["b", "c"])
    ckde.fit(train)
    assert abs(float(vals["ckde_slogl"]) - ckde.slogl(test)) <= 1e-9 * abs(ckde.slogl(test))
    bic = pbn.BIC(train).local_score(pbn.GaussianNetwork(["a", "b", "c"]), "c", ["a", "b"])
    assert abs(float(vals["bic_c_ab"]) - bic) <= 1e-9 * abs(bic)
    assert int(vals["bad_rc"]) == 1  # PBN_ERR_INVALID
