"""CPU tier: the arithmetic of exp2_magic (pybnesian_amd/csrc/kde_kernels.hip), word for word in numpy.  The kernel form cannot run here; what can be
checked without a GPU is that the bit manipulation IS 2^x: the constants the source defines (parsed from it), the fraction cut with its
round-to-nearest, the exponent add on the high word, the clamp's two ends, and what accumulators outside the binade - |x| >= 2^19, inf, NaN - turn
into.  v_exp_f32 is taken as numpy's float32 exp2 (the hardware instruction is good to 1 ulp; the GPU tier measures it)."""
import os
import re

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = open(os.path.join(ROOT, "pybnesian_amd", "csrc", "kde_kernels.hip")).read()


def _constants():
    c = re.search(r"#define PBN_MAGIC_C \((0x1\.8p20) - (1\.0) \+ (0x1p-24)\)", SRC)
    h = re.search(r"#define PBN_MAGIC_H0 (0x[0-9a-fA-F]+)", SRC)
    assert c and h, "the constants of exp2_magic moved: update this test with them"
    return float.fromhex(c.group(1)) - float(c.group(2)) + float.fromhex(c.group(3)), int(h.group(1), 16)


def _words(y):
    b = np.asarray(y, dtype=np.float64).view(np.uint64)
    return (b & np.uint64(0xFFFFFFFF)).astype(np.uint32), (b >> np.uint64(32)).astype(np.uint32)


def exp2_magic(x, clamp=True):
    """2^x from the accumulator y = x + PBN_MAGIC_C, as the device function does it."""
    mc, h0 = _constants()
    with np.errstate(invalid="ignore", over="ignore"):
        y = np.asarray(x, dtype=np.float64) + mc
        lo, hi = _words(y)
        u = ((np.uint32(0x7F) << np.uint32(23)) | (lo >> np.uint32(9))).view(np.float32)          # v_alignbit_b32(0x7f, lo, 9)
        ed = np.exp2(u).astype(np.float32).astype(np.float64)                                      # v_exp_f32, v_cvt_f64_f32
        t = hi.view(np.int32).astype(np.int64)
        if clamp:
            t = np.clip(t, h0 - 1024, h0 + 1023)                                                   # v_med3_i32
        elo, ehi = _words(ed)
        hi2 = ((ehi.astype(np.int64) + ((t & 0xFFF) << 20)) & 0xFFFFFFFF).astype(np.uint64)        # v_lshl_add_u32 (mod 2^32)
        return ((hi2 << np.uint64(32)) | elo.astype(np.uint64)).view(np.float64)


def test_constants_are_what_the_form_needs():
    mc, h0 = _constants()
    _, hi = _words(np.array([1.5 * 2.0 ** 20]))
    assert int(hi[0]) == h0 and (h0 & 0xFFF) == 0          # the shift by 20 must drop the constant's own bits
    assert mc == 1.5 * 2.0 ** 20 - 1.0 + 2.0 ** -24        # exactly representable: 45 significant bits


def test_it_is_two_to_the_x_inside_the_range():
    rng = np.random.default_rng(0)
    x = np.concatenate([rng.uniform(-1021.0, 1022.0, 200_000), rng.uniform(-3.0, 3.0, 50_000), np.arange(-1021, 1023, dtype=np.float64),
                        np.arange(-1021, 1022) + 1.0 - 2.0 ** -30, np.arange(-1021, 1022) + 2.0 ** -30])
    got, bare = exp2_magic(x), exp2_magic(x, clamp=False)
    want = np.exp2(x)
    rel = np.abs(got - want) / want
    assert rel.max() <= 1.65e-7, rel.max()                 # grid 2^-32 + fraction to 2^-24 + one float ulp of a value in [2, 4]
    assert np.array_equal(got, bare)                       # inside +-1022 the clamp is a no-op: the guards drop it without changing a bit
    ints = np.arange(-1000, 1000, dtype=np.float64)
    assert np.array_equal(exp2_magic(ints), np.exp2(ints)) # continuous across the integers: f = 0 gives exactly 2^n
    # round-to-nearest on the fraction: no bias from the cut (the exponent the form evaluates, against x; numpy's float32 exp2 has a bias of
    # its own, -4e-9, so the RESULT's mean says nothing about the form)
    mc, h0 = _constants()
    lo, hi = _words(x + mc)
    xq = (hi.astype(np.int64) - h0) + 1.0 + (lo >> np.uint32(9)).astype(np.float64) * 2.0 ** -23
    assert np.abs(xq - x).max() <= 2.0 ** -24 + 2.0 ** -32 and abs(np.mean(xq - x)) <= 1e-9


def test_the_clamp_makes_the_form_total():
    with np.errstate(over="ignore", invalid="ignore"):
        big = exp2_magic(np.array([1025.0, 1500.0, 2.0 ** 19, 2.0 ** 30, 1e300, np.inf, np.nan]))
        assert np.all(~np.isfinite(big))                   # NaN or inf: what the sweeps' overflow tests catch (sum < 2^1000 fails)
        small = exp2_magic(np.array([-1024.5, -1100.0, -1e4, -(2.0 ** 19), -(2.0 ** 30), -1e300, -np.inf]))
        assert np.all(np.abs(small) <= 2.0 ** -1019)       # a term at least 2^-1147 below a sum that carries 2^128
        edge = exp2_magic(np.array([-1023.5, 1023.5]))
        assert 0.0 < edge[0] <= 2.0 ** -1022 and edge[1] > 2.0 ** 1000
        # without the clamp these wrap: that is what the guards (tile radii / batch boxes) must exclude before the bare form runs
        wrapped = exp2_magic(np.array([-3000.0, 1500.0]), clamp=False)
        assert np.isfinite(wrapped).all() and (abs(wrapped[0]) > 1e-300 or wrapped[1] < 1e300)
