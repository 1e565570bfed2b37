"""GPU tier, fp32 tables: adversarial inputs for the f16x2 fragments (round 6; csrc/kde_kernels.hip "fp32 path on the 16-bit matrix cores").
Every whitened coordinate is cut into two f16 pieces z^ = a1 + a2; the products a1 b1, a1 b2, a2 b1 are exact, a2 b2 rides along only where
the 32-slot blocks have room (spd = 4: d = 1...7, 10...15) and is DROPPED otherwise (spd = 3: d = 8, 9, 16...20).  On ordinary data the
dropped products are zero-mean and tiny; here they are made one-sided and as large as the geometry allows: a training cluster and its
queries sit at the BOTTOM of an f16 binade in every coordinate (|z_k| just above 16, ulp 2^-6) with the same residual 0.4 ulp in every
coordinate of every row, so that every pair that matters drops sum_k a2 b2 = d (0.4 x 2^-6)^2 with one sign.
  (i)   d = 8 (three products): the error of every logl is one-sided and stays inside the stated bound 2^-22 |z|^2 ln 2 - and inside the
        reference tests' fp32 tolerance (atol 5e-4, KDE_test.py:181-182); beside it the f32 accumulation of the matrix cores (partial sums of
        ~2 000: spacing 2.4e-4), which is one-sided on this table too;
  (ii)  d = 7 (four products) on the same construction: nothing is dropped - what is left is that accumulation, <= 3 x 2^-24 |z|^2;
  (iii) the fragments rule: the same d = 8 table pushed out until 2^-22 max|z|^2 exceeds 2 x PBN_F32_WIDEN_AT is evaluated on fp64 fragments
        (error ~1e-7), one that stays just inside is not.
The truth is the f64 oracle on the same float values (oracle/pbn_oracle.cpp following kde/opencl_kernels/KDE.cl.src:123-233 in double)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634
ULP16 = 2.0 ** -6          # f16 spacing in [16, 32)


def _case(d, base=16.0, resid=0.4, n_side=600, n_q=400, seed=3):
    """Training rows: a cluster at +v and its mirror at -v (column means exactly 0), v_k = base + resid ulp; rows of a cluster differ by whole
    ulps (residuals untouched).  Queries beside the + cluster, also whole ulps away.  Bandwidth h = log2(e) I: whitened z = x."""
    rng = np.random.default_rng(seed)
    v = base + resid * ULP16
    off_t = rng.integers(0, 97, size=(n_side, d)) * ULP16            # up to 1.5 whitened units, never below the binade's bottom (ulp halves there)
    plus = (v + off_t).astype(np.float32).astype(np.float64)
    train = np.empty((2 * n_side, d))
    train[0::2], train[1::2] = plus, -plus      # interleaved: the library centres on the mean of the first 1 024 rows - exactly 0 here
    train = train.astype(np.float32)
    test = (v + rng.integers(0, 97, size=(n_q, d)) * ULP16).astype(np.float32)
    return train, test


def _logl(train, test, d):
    import pybnesian_amd as pbn
    from oracle import oracle

    names = [f"v{i}" for i in range(d)]
    k = pbn.ProductKDE(names)
    k.fit(pd.DataFrame(train, columns=names))
    k.bandwidth = np.full(d, LOG2E)
    got = k.logl(pd.DataFrame(test, columns=names))
    want = oracle.product_kde_logl(train.astype(np.float64), k.bandwidth, test.astype(np.float64))
    return got, want


def test_dropped_products_one_sided_stay_inside_their_bound():
    d = 8
    train, test = _case(d)
    got, want = _logl(train, test, d)
    err = got - want
    z2 = float((train.astype(np.float64) ** 2).sum(axis=1).max())
    bound = 2.0 ** -22 * z2 / LOG2E                     # 2^-22 |z|^2 exponent units -> natural log
    expect = d * (0.4 * ULP16) ** 2 / LOG2E             # what this construction drops for every pair that matters
    print(f"\n[f32 budget] d = 8, three products: logl error mean {err.mean():.3e} (construction drops {expect:.3e}), worst {np.abs(err).max():.3e} = "
          f"{np.abs(err).max() / bound:.2f} of the bound {bound:.3e}")
    assert np.all(np.isfinite(got))
    # the device's exponents lack +sum a2 b2 (one-sided by construction) AND carry the f32 accumulation of the matrix cores, which at partial sums of
    # ~2 000 (f32 spacing 2.4e-4) is one-sided too here: up to ~2 x 2^-24 |z|^2
    acc = 2.0 * 2.0 ** -24 * z2 / LOG2E
    assert expect * 0.9 <= -err.mean() <= expect + acc, "the one-sided error is the dropped products plus the accumulation's"
    assert np.abs(err).max() <= bound + acc
    assert np.abs(err).max() <= 5e-4                    # the reference tests' own fp32 tolerance per logl


def test_four_products_drop_nothing():
    d = 7
    train, test = _case(d)
    got, want = _logl(train, test, d)
    err = np.abs(got - want).max()
    print(f"\n[f32 budget] d = 7, four products: worst logl error {err:.3e}")
    z2 = float((train.astype(np.float64) ** 2).sum(axis=1).max())
    assert err <= 3.0 * 2.0 ** -24 * z2 / LOG2E       # the f32 accumulation of the Gram form (what PBN_F32_WIDEN_AT prices for these layouts), one-sided here


def test_fragments_rule_follows_the_layout(monkeypatch):
    """d = 8: 2^-22 max|z|^2 <= 1e-3 stays on f16x2 fragments (error of the dropped products visible), beyond it fp64 fragments (error gone)."""
    d = 8
    inside = _case(d, base=16.0)                         # |z|^2 = 2 050: 2^-22 |z|^2 = 4.9e-4
    outside = _case(d, base=32.0, resid=0.4 * 2)         # |z|^2 = 8 200: 2^-22 |z|^2 = 2.0e-3 > 1e-3 (ulp 2^-5 there: same residual in units of it)
    e_in = np.abs(np.subtract(*_logl(*inside, d))).max()
    e_out = np.abs(np.subtract(*_logl(*outside, d))).max()
    print(f"\n[f32 budget] fragments rule: inside {e_in:.3e} (f16x2 fragments), outside {e_out:.3e} (fp64 fragments)")
    assert e_in > 1e-4 and e_out < 5e-6
    monkeypatch.setenv("PBN_F32_WIDEN_AT", "inf")        # what the f16x2 fragments would have given out there: 4x the inside error
    e_forced = np.abs(np.subtract(*_logl(*outside, d))).max()
    assert e_forced > 1.5 * e_in
