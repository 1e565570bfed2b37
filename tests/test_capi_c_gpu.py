"""GPU tier: plain-C programs (tests/c/*.c) drive the C ABI directly - no Python, no torch - and a C caller must get the REFERENCE's
numbers: the outputs are compared with the oracle (oracle/pbn_oracle.cpp, the reference's arithmetic) on the same xorshift tables, and
only in second place with the Python mirror of the same library."""
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


def test_c_program_matches_the_oracle(tmp_path):
    import pybnesian_amd as pbn
    from oracle import oracle

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
    tr, te = train.to_numpy(), test.to_numpy()
    # the reference's arithmetic on the same rows (north star: slogl within 1e-6 relative in fp64; held to 1e-8 here - slogl takes the
    # sum-only sweep, 2^f on the fp32 transcendental unit: measured 1.2e-9 on this table)
    H = oracle.bandwidth(0, 0, oracle.cov(tr)[0], n)          # NormalReferenceRule, full matrix (NormalReferenceRule.hpp:72-134)
    want_kde = oracle.kde_logl(tr, H, te).sum()
    want_ckde = oracle.ckde_logl(tr, H, te).sum()          # column 0 given columns 1, 2 (CKDE.hpp:256-287)
    want_bic = oracle.bic_lg(tr[:, [2, 0, 1]])             # c | a, b (bic.cpp:29-64)
    assert abs(float(vals["kde_slogl"]) - want_kde) <= 1e-8 * abs(want_kde)
    assert abs(float(vals["ckde_slogl"]) - want_ckde) <= 1e-8 * abs(want_ckde)
    assert abs(float(vals["bic_c_ab"]) - want_bic) <= 1e-9 * abs(want_bic)
    assert int(vals["bad_rc"]) == 1  # PBN_ERR_INVALID
    # and the Python mirror gives what the C caller got.  Not bit for bit: the two hosts hand over tables whose whitened rows differ in the
    # last ulp, and the sum-only sweeps' 2^x (exp2_magic) quantises every exponent to 2^-32 and its fraction to 2^-24 - an ulp on the input
    # moves a few of the 3.5e6 terms across a rounding boundary (1.6e-10 / 8e-8 of ONE term each): 1e-11 of the sum here, held to 1e-10
    kde = pbn.KDE(["a", "b", "c"])
    kde.fit(train)
    assert abs(float(vals["kde_slogl"]) - kde.slogl(test)) <= 1e-10 * abs(kde.slogl(test))
    ckde = pbn.CKDE("a", ["b", "c"])
    ckde.fit(train)
    assert abs(float(vals["ckde_slogl"]) - ckde.slogl(test)) <= 1e-10 * abs(ckde.slogl(test))


def test_c_host_with_rccl_shards_the_delta_cache(tmp_path):
    """tests/c/shard_rccl_demo.c: a C host that owns an RCCL communicator (one rank on this GPU) and hands the library ONE function, an
    all-gather over ncclAllGather on device memory.  pbn_score_batch on the handle with the communicator bound plans the batch, evaluates
    this rank's share, makes one collective and assembles - the scores must be the one-process scores bit for bit (the program checks), and
    the CV likelihoods the oracle's (checked here)."""
    from oracle import oracle

    exe = str(tmp_path / "shard_rccl_demo")
    subprocess.check_call(["gcc", "-O2", "-D__HIP_PLATFORM_AMD__", "-I" + os.path.join(ROOT, "include"), "-I/opt/rocm/include",
                           os.path.join(ROOT, "tests", "c", "shard_rccl_demo.c"), "-L" + os.path.join(ROOT, "pybnesian_amd"), "-lpbn_hip",
                           "-L/opt/rocm/lib", "-lrccl", "-lamdhip64", "-Wl,-rpath," + os.path.join(ROOT, "pybnesian_amd"), "-Wl,-rpath,/opt/rocm/lib",
                           "-lm", "-o", exe])
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0")
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    vals = dict(line.split() for line in out.stdout.strip().splitlines() if len(line.split()) == 2)
    assert int(vals["rccl_ranks"]) == 1 and int(vals["bit_identical"]) == 1
    assert int(vals["collectives_first_batch"]) == 1            # ONE all-gather per batch
    assert int(vals["collectives_total"]) == 2                  # moments + the first batch; the second batch finds every term installed and deals nothing: no collective
    assert int(vals["sweeps_sharded"]) == int(vals["sweeps_one"])   # one rank: dealt everything, nothing swept twice
    n = 3000
    u = _xorshift_stream(8 * n).reshape(n, 8)
    a = u[:, 0] + u[:, 1] + u[:, 2] - 1.5
    b = u[:, 3] + u[:, 4] - 1.0
    c = u[:, 5] - 0.5
    e = u[:, 6] + u[:, 7] - 1.0
    x = np.column_stack([a, 0.6 * a + b, a - 0.4 * b + c, 0.3 * c + e])
    cands = [(1, [0], "ckde"), (0, [1], "ckde"), (2, [0, 1], "ckde"), (3, [], "ckde"), (3, [2], "ckde"), (2, [0], "lg")]
    for i, (v, ps, kind) in enumerate(cands):
        want = oracle.cv_likelihood(x[:, [v] + ps], kind, 3, 7)
        assert abs(float(vals[f"score_{i}"]) - want) <= 1e-6 * abs(want), (i, vals[f"score_{i}"], want)
