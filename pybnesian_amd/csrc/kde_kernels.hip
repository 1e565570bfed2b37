// KDE / ProductKDE / CKDE log-likelihood sweep for gfx950 (MI355X).
//
// Replaces the reference's tile-of-64 OpenCL pipeline
//   substract -> solve -> square -> logl_values_mat_* -> max_mat_cols -> logsumexp_coeffs ->
//   sum_mat_cols -> finish_lse_offset -> sum1d
// (/root/reference/pybnesian/kde/KDE.hpp:592-640, kde/opencl_kernels/KDE.cl.src:115-233,
//  opencl/opencl_config.hpp:517-536) with three kernels:
//
//   pack_rows     z = sqrt(log2 e) * L^-1 (x - mu)  (whiten + centre + scale to base-2 units), written
//                 in MFMA 16x16x4 operand-fragment order together with -1/2 |z|^2.  After this
//                 s2(t,q) = log2(e) * (-1/2 |L^-1 (x_t - y_q)|^2) = z_t . z_q - 1/2|z_t|^2 - 1/2|z_q|^2,
//                 so the per-pair triangular solve of KDE.cl.src:123-135 disappears.
//   kde_sweep     each wave owns QG groups of 16 query rows (B fragments live in registers) and
//                 streams 16-row training tiles (A fragments, coalesced 512 B loads).  The
//                 dot products run on the matrix pipe (v_mfma_f64_16x16x4_f64 / v_mfma_f32_16x16x4_f32)
//                 with the accumulator pre-loaded with  -1/2|z_t|^2 - 1/2|z_q|^2 - m_q ; the VALU only
//                 evaluates 2^x and the running sum.  Online logsumexp: m_q is a per-query offset
//                 that is raised (rarely, wave-uniform slow path) when a term would overflow.
//                 No N x 64 tile is ever materialised.
//                 CKDE (COND=true): evidence-first Cholesky makes the marginal's whitened coordinates a
//                 prefix of the joint's, so ONE sweep yields both logsumexps: the joint accumulator is
//                 the marginal accumulator plus one extra augmented MFMA k-step.
//   kde_finish    merge the per-split (m, sum) partials in fixed order, add the log-normalisation,
//                 write logl and/or a deterministic tree-reduced slogl.
#include "common.hpp"
#include <atomic>
#include "kde_kernels.hpp"
#include "kde_group.hpp"

#include <cmath>
#include <cstdlib>
#include <type_traits>

namespace pbn {

// Device pointers of the sweeps are typed as GLOBAL-address-space pointers.  A pointer that reaches a kernel through a record in
// memory (the grouped launches' per-unit table) is otherwise a generic pointer: its loads become flat_load, whose completion order
// against LDS traffic is unknown, so every wait is a full `s_waitcnt vmcnt(0) lgkmcnt(0)` - the prefetch of the next tile is waited
// for before the current one is used.  (The kernel-argument pointers of the stand-alone launches are inferred global anyway.)
#define PBN_GLOBAL __attribute__((address_space(1)))

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

// 2^r on [-1/2, 1/2]: interpolants at Chebyshev nodes (exp2_poly), and minimax polynomials on [0, 1) for the v_fract form
// of the sweep's main loop (exp2_f64_fract).  The DP units are the binding resource of the fp64 sweep (DESIGN.md
// "roofline"); every polynomial degree costs one of its ~42 DP instructions per (tile, group, 4 values).  The
// log-likelihood sweeps use degree 6 on [0, 1) pinned to p(0) = 1, p(1) = 2 (Remez on the relative error among such
// polynomials, tools/exp2_coeffs.py 6 pinned): max relative error 2.22e-9 per term, i.e. <= 2.3e-9 ABSOLUTE on a logl
// whatever the number of terms (the error of a sum of positive terms is bounded by the per-term bound) - 400x inside the
// 1e-6 parity bar.  The pinning keeps 2^x continuous across the integers, where v_fract wraps: the largest term of a KDE
// sum sits at x = bias + 0 exactly, and the free minimax of even degree (1.86e-9) has errors of opposite sign at the two
// ends - a jump of 3.7e-9 on that term under perturbations of the last ulp.  -DPBN_EXP2_DEGREE=7 restores 4.0e-11
// (free minimax: same sign at both ends), =8 1.07e-12 (v_rndne form).
// The weight kernels (CKDE::cdf / sample, UCV) always use degree 8: the UCV objective is a difference of two pair sums
// and amplifies per-term errors.
#ifndef PBN_EXP2_DEGREE
#define PBN_EXP2_DEGREE 6
#endif

template <int DEG>
__device__ __forceinline__ double exp2_poly(double r);
template <>
__device__ __forceinline__ double exp2_poly<8>(double r) {
    double p = 0x1.63d136366db24p-20;
    p = __builtin_fma(p, r, 0x1.00dc4a532fb8ep-16);
    p = __builtin_fma(p, r, 0x1.4308ac85aa947p-13);
    p = __builtin_fma(p, r, 0x1.5d8745a728441p-10);
    p = __builtin_fma(p, r, 0x1.3b2ab7181b755p-7);
    p = __builtin_fma(p, r, 0x1.c6b08dd6fd234p-5);
    p = __builtin_fma(p, r, 0x1.ebfbdff823cedp-3);
    p = __builtin_fma(p, r, 0x1.62e42fef84cf0p-1);
    return __builtin_fma(p, r, 0x1.0000000000000p+0);
}
template <>
__device__ __forceinline__ double exp2_poly<7>(double r) {
    double p = 0x1.00c0e56000f6ep-16;
    p = __builtin_fma(p, r, 0x1.446c79f27429dp-13);
    p = __builtin_fma(p, r, 0x1.5d8775970d4b9p-10);
    p = __builtin_fma(p, r, 0x1.3b29d8bb04b01p-7);
    p = __builtin_fma(p, r, 0x1.c6b08da70e83cp-5);
    p = __builtin_fma(p, r, 0x1.ebfbe0aa03e9fp-3);
    p = __builtin_fma(p, r, 0x1.62e42fef9cc4fp-1);
    return __builtin_fma(p, r, 0x1.ffffffffa7138p-1);
}

// the leading coefficient of the degree-7 polynomial as a register operand: v_fma_f64 reads one scalar/literal only, so
// the first step C7 r + C6 needs one of the two in a VGPR; a caller that pins it once (pin_top) saves the v_mov the
// compiler otherwise re-materialises per 4 values
__device__ __forceinline__ double pin_top() {
    double c;
    asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "s"(0x1.00c0e56000f6ep-16));
    return c;
}
__device__ __forceinline__ double exp2_f64_top(double x, double top) {
    double nf = __builtin_rint(x);
    double r = x - nf;
    double p = __builtin_fma(top, r, 0x1.446c79f27429dp-13);
    p = __builtin_fma(p, r, 0x1.5d8775970d4b9p-10);
    p = __builtin_fma(p, r, 0x1.3b29d8bb04b01p-7);
    p = __builtin_fma(p, r, 0x1.c6b08da70e83cp-5);
    p = __builtin_fma(p, r, 0x1.ebfbe0aa03e9fp-3);
    p = __builtin_fma(p, r, 0x1.62e42fef9cc4fp-1);
    p = __builtin_fma(p, r, 0x1.ffffffffa7138p-1);
    int n;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(nf));
    return __builtin_ldexp(p, n);
}

// 2^x for the sweep's main loop, 10 instructions instead of 11: the caller keeps its exponents biased by
// PBN_EXP2_BIAS (folded into the per-query constant, so free), which makes every value that matters non-negative; then
// v_fract_f64 is the whole range reduction (f = x - floor(x), exact) and the truncating v_cvt_i32_f64 of x itself is
// floor(x) - no v_rndne / subtract pair.  Degree-7 minimax (relative, Remez: tools/exp2_coeffs.py 7 0 1) on [0, 1):
// 4.02e-11.  A negative x (a term below 2^-128 of its query's sum, which holds a term >= 2^0) comes out at most 2x too
// large: invisible (N * 2^-128 relative); x <= -2^31 saturates to INT_MIN and gives 0 like the general form.
#define PBN_EXP2_BIAS 128.0
#if PBN_EXP2_DEGREE == 6
#define PBN_FRACT_TOP 0x1.c765a82c535bdp-13
#else
#define PBN_FRACT_TOP 0x1.68b07e4ac7b5bp-16
#endif
__device__ __forceinline__ double pin_top_fract() {
    double c;
    asm volatile("v_mov_b64 %0, %1" : "=v"(c) : "s"(PBN_FRACT_TOP));
    return c;
}
// FAST: 2^f of the FRACTION on the fp32 transcendental unit instead of the fp64 polynomial - fract, cvt_i32, cvt_f32_f64, v_exp_f32,
// cvt_f64_f32, ldexp: 6 instructions (v_exp_f32 holds the issue port for two slots) instead of 9.  f in [0, 1) converts to float with
// <= 6e-8 absolute error, v_exp_f32 is good to 1 ulp of a value in [1, 2]: <= 1.4e-7 relative per term (measured on C2: 5e-8 absolute
// on a logl at worst, 1e-10 relative on the slogl) instead of 2.2e-9 - and a sum of positive terms moves by at most the per-term
// bound.  Like the pinned polynomial it is continuous across the integers (f = 0 gives exactly 1; an f that rounds to 1.0f gives
// exactly 2) and, with integer offsets, a function of the (row, query) pair only: sums taken in different partitions still agree to
// rounding.  Used by the sweeps whose result is a SUM over the test rows (slogl, the score engine's terms: the north star's bar is
// 1e-6 relative on slogl); per-row logl outputs keep the polynomial (SweepArgs::fast).  C2 51.5 -> 46.3 ms, cv64 3.42 -> 3.06 s,
// bounded C3 15.7 -> 14.0 s (profiles/r4/expf32_probe.txt).  -DPBN_EXP2_F32=0 compiles it out.
// Round 6: the plain sum-only sweeps (every shape but the fused CKDE ones) take exp2_magic below instead - the same v_exp_f32, fed from the
// accumulator's own words; this form stays for the fused conditional sweeps and for -DPBN_EXP2_MAGIC=0.
#ifndef PBN_EXP2_F32
#define PBN_EXP2_F32 1
#endif
template <bool FAST = false>
__device__ __forceinline__ double exp2_f64_fract(double x, double top) {
    const double f = __builtin_amdgcn_fract(x);      // v_fract_f64
    int n;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(x));  // truncation = floor for x >= 0; saturating
    if constexpr (FAST && PBN_EXP2_F32) {
        (void)top;
        return __builtin_ldexp((double)__builtin_amdgcn_exp2f((float)f), n);
    }
#if PBN_EXP2_DEGREE == 6
    double p = __builtin_fma(top, f, 0x1.46214fe0d40c9p-10);
    p = __builtin_fma(p, f, 0x1.3d217bf137896p-7);
    p = __builtin_fma(p, f, 0x1.c686b389d4c31p-5);
    p = __builtin_fma(p, f, 0x1.ebfd7378d3f76p-3);
    p = __builtin_fma(p, f, 0x1.62e42af6f5a89p-1);
    p = __builtin_fma(p, f, 1.0);
#else
    double p = __builtin_fma(top, f, 0x1.2cfd657b74f58p-13);
    p = __builtin_fma(p, f, 0x1.5fddc72ac74dep-10);
    p = __builtin_fma(p, f, 0x1.3b0838502e0f5p-7);
    p = __builtin_fma(p, f, 0x1.c6b2b0142cedbp-5);
    p = __builtin_fma(p, f, 0x1.ebfbcf8c8da34p-3);
    p = __builtin_fma(p, f, 0x1.62e4301f16f2dp-1);
    p = __builtin_fma(p, f, 0x1.ffffffffa7934p-1);
#endif
    return __builtin_ldexp(p, n);
}

// MAGIC (round 6): the 2^x of sum-only fp64 sweeps without a single DP instruction of range reduction.  The per-query constant that
// starts the MFMA accumulator carries PBN_MAGIC_C = 1.5 * 2^20 - 1 + 2^-24 on top of the biased exponent, so the MFMA chain itself leaves
// y = 1.5 * 2^20 + (x - 1 + 2^-24): a double of FIXED exponent whose mantissa is x in fixed point - the low word is the fraction (32
// bits), the high word is 0x41380000 + floor(x - 1 + 2^-24).  Then
//   u  = v_alignbit_b32(0x7f, y.lo, 9)      the float 1 + f, f = the fraction's top 23 bits (the 2^-24 in the constant makes the cut
//                                            a round-to-nearest of x: +-2^-24, no bias);
//   e  = v_exp_f32(u) in [2, 4]              = 2^(1 + f), 1 ulp;
//   ed = v_cvt_f64_f32(e);  ed.hi += n << 20 (v_lshl_add_u32; n = y.hi clamped by v_med3_i32 to 0x41380000 - 1024 ... + 1023)
// = 2^x in 6 instructions / 7 issue slots with the sum's FMA, against 7 / 8 of the v_fract form (fract, cvt_i32, cvt_f32, exp, cvt_f64,
// ldexp, fma).  The clamp makes the form total: the exponent field of ed (1024 or 1025) + n stays inside [0, 2047] - n = -1024 gives a
// subnormal or 2^-1022 (a term 2^-1150 below its sum), n = 1023 gives NaN or inf, which the sums' overflow tests catch exactly like the
// inf of the v_fract form; an accumulator outside [2^20, 2^21) - |x| beyond 2^19, NaN, inf - has a high word beyond the clamp's ends and
// comes out as ~0 (x -> -inf) or NaN (everything else).  Accuracy per term: x on a 2^-32 grid (the MFMA chain rounds there: <= 1e-9),
// f to 2^-24 (4.1e-8 relative), v_exp_f32 1 ulp of a value in [2, 4] (<= 1.2e-7): <= 1.65e-7, against 1.4e-7 of the v_fract form.
// Like that form it is a function of the (row, query) pair alone: the offsets are integers, and an integer added to y moves the high
// word only (the grid and every rounding of the chain stay where they are while y stays in its binade).
#ifndef PBN_EXP2_MAGIC
#define PBN_EXP2_MAGIC 1
#endif
#ifndef PBN_MAGIC_CLAMP
#define PBN_MAGIC_CLAMP 1   // 0: probe builds only (the unclamped 5-instruction form: wraps on exponents beyond +-1023)
#endif
#ifndef PBN_MAGIC_PRUNED
#define PBN_MAGIC_PRUNED 1   // the pruned / grouped sum-only sweeps too (their far tiles pay one v_add_f64 per value to take the constant off)
#endif
#ifndef PBN_MAGIC_GUARD
#define PBN_MAGIC_GUARD 1   // unpruned sweeps: chunks whose exponents are proven inside +-1022 skip the clamp (kde_sweep_body: GUARD)
#endif
#define PBN_MAGIC_C (0x1.8p20 - 1.0 + 0x1p-24)
#define PBN_MAGIC_H0 0x41380000
template <bool CLAMP = true>
__device__ __forceinline__ double exp2_magic(double y) {
    const unsigned lo = (unsigned)__double2loint(y);
    int t = __double2hiint(y);
    const float u = __uint_as_float(__builtin_amdgcn_alignbit(0x7fu, lo, 9));
    const double ed = (double)__builtin_amdgcn_exp2f(u);
    if constexpr (CLAMP) {
        t = t < PBN_MAGIC_H0 - 1024 ? PBN_MAGIC_H0 - 1024 : t;
        t = t > PBN_MAGIC_H0 + 1023 ? PBN_MAGIC_H0 + 1023 : t;   // (v_med3_i32)
    }
    unsigned h2;
    if constexpr (CLAMP) h2 = (unsigned)__double2hiint(ed) + ((unsigned)t << 20);
    else asm("v_lshl_add_u32 %0, %1, 20, %2" : "=v"(h2) : "v"(t), "v"(__double2hiint(ed)));   // (left to the compiler this becomes three 64-bit operations)
    return __hiloint2double((int)h2, __double2loint(ed));
}

template <int DEG>
__device__ __forceinline__ double exp2_f64(double x) {
    // x <= ~1000 (larger values are caught by the overflow check of the caller), any negative value.
    double nf = __builtin_rint(x);  // v_rndne_f64
    double r = x - nf;              // exact
    const double p = exp2_poly<DEG>(r);
    int n;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(n) : "v"(nf));  // saturating: -1e30 -> INT_MIN -> ldexp gives 0
    return __builtin_ldexp(p, n);                      // v_ldexp_f64
}

template <typename T>
struct Tr;
template <>
struct Tr<double> {
    using vec4 = d4;
    static __device__ __forceinline__ vec4 mfma(double a, double b, vec4 c) {
        return __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    }
    static constexpr int GEN_DEG = PBN_EXP2_DEGREE < 7 ? 7 : PBN_EXP2_DEGREE;   // rare paths: v_rndne form, degree >= 7
    static __device__ __forceinline__ double ex2(double x) { return exp2_f64<GEN_DEG>(x); }
    static __device__ __forceinline__ double top() { return PBN_EXP2_DEGREE <= 7 ? pin_top_fract() : 0.0; }
    // main-loop form: x carries bias() (see exp2_f64_fract)
    template <bool FAST = false>
    static __device__ __forceinline__ double ex2p(double x, double top) {
        return PBN_EXP2_DEGREE <= 7 ? exp2_f64_fract<FAST>(x, top) : exp2_f64<GEN_DEG>(x);
    }
    static __device__ __forceinline__ double bias() { return PBN_EXP2_BIAS; }
    // MAGIC sweeps (exp2_magic): the constant on top of the biased exponents, and 2^x from such an accumulator
    static __device__ __forceinline__ double magic() { return PBN_MAGIC_C; }
    template <bool CLAMP = true>
    static __device__ __forceinline__ double ex2m(double y) { return exp2_magic<CLAMP>(y); }
    static __device__ __forceinline__ double ex2_hi(double x) { return exp2_f64<8>(x); }
    static __device__ __forceinline__ double big() { return 0x1p900; }
    // C/D row held by (lane group lg, register i): cdna_hip_programming.md §3 "f64 MFMA"
    static __host__ __device__ __forceinline__ int crow(int lg, int i) { return lg + 4 * i; }
};
template <>
struct Tr<float> {
    using vec4 = f4;
    static __device__ __forceinline__ vec4 mfma(float a, float b, vec4 c) {
        return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
    }
    static __device__ __forceinline__ float ex2(float x) { return __builtin_amdgcn_exp2f(x); }  // v_exp_f32
    static __device__ __forceinline__ float ex2_hi(float x) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float top() { return 0.0f; }
    template <bool FAST = false>
    static __device__ __forceinline__ float ex2p(float x, float) { return __builtin_amdgcn_exp2f(x); }
    static __device__ __forceinline__ float bias() { return 0.0f; }
    static __device__ __forceinline__ float magic() { return 0.0f; }
    template <bool CLAMP = true>
    static __device__ __forceinline__ float ex2m(float y) { return __builtin_amdgcn_exp2f(y); }
    static __device__ __forceinline__ float big() { return 0x1p100f; }
    static __host__ __device__ __forceinline__ int crow(int lg, int i) { return 4 * lg + i; }
};

#define PBN_PAD_NORM (-1e30)
// measurement aid, not part of the C ABI header: (wave, split) units of the fp64 sweep that had to redo their split checked
__device__ unsigned long long g_sweep_redo = 0, g_sweep_units = 0;
__device__ unsigned long long g_sweep_visit = 0, g_sweep_tiles = 0;
__device__ unsigned long long g_mom_pairs = 0, g_mom_batches = 0, g_mom_visits = 0, g_mom_left = 0;
__device__ unsigned long long g_mom_taken[2] = {0, 0};   // always on: (tile, group) pairs the moment pass took, by dimension - one atomic per wave (pbn_debug_moment_totals)   // moment pass (PBN_SWEEP_COUNT_REDO): pairs taken / (batch, group) passes made
// (pruned sweeps: tiles visited / tiles offered, per wave)
// waves per SIMD the pruned fp64 sweeps are compiled for: 3 (<= 168 VGPRs) - the blind-batch shapes fit anyway, the checked
// d = 4 / 5 and norm-multiplying shapes (183-207 unconstrained) gain 3-9 % on the 1e6 x 1e5 handles; 4 (128, spills) loses on C3
#ifndef PBN_F64_PRUNE_WAVES
#define PBN_F64_PRUNE_WAVES 3
#endif
#ifndef PBN_FAR_F32
#define PBN_FAR_F32 1   // pruned plain fp64 sum-only sweeps: far tiles through the fp32 unit (kde_sweep_body: FARP; SweepArgs::far_span)
#endif
#ifndef PBN_SWEEP_UNCHECKED
#define PBN_SWEEP_UNCHECKED 1   // fp64 plain unpruned sweeps: blind first pass, checked redo (kde_sweep_kernel)
#endif

// ------------------------------------------------------------------------------------------------
// pack_rows: one thread per (padded) row.
//   main components c < dm   -> pack[(tile*KS + c/4)*64 + (c%4)*16 + idx]   (A and B fragment order
//                               coincide: element [idx = lane&15][k = lane>>4])
//   norm  -1/2 sum_{c<dm} z^2 -> npack: training side in C-row order [tile][lg][i], query side [tile][idx]
//   extra component (CKDE)   -> xpack[tile*64 + k*16 + idx]:
//        training: k0 z_e, k1 -1/2 z_e^2, k2 1, k3 0      query: k0 z_e, k1 1, k2 -1/2 z_e^2, k3 0
// ------------------------------------------------------------------------------------------------
// T = fragment type, TS = element type of the table (float under double fragments: PackArgs::src_f32)
template <typename T, typename TS = T>
__global__ __launch_bounds__(256) void pack_rows_kernel(PackArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t npad = a.ntiles * 16;
    if (r >= npad) return;
    const int64_t tile = r >> 4;
    const int idx = (int)(r & 15);
    const int d = a.d, dm = a.dm, KS = a.KS;
    T* pack = (T*)a.pack;
    T* npack = (T*)a.npack;
    T* xpack = (T*)a.xpack;
    const bool valid = r < a.n;

    double xc[PBN_MAX_D];
    if (valid) {
        const int64_t rr = a.perm ? (int64_t)a.perm[a.perm_stride > 1 ? r * a.perm_stride : r] : r;
        const int64_t lr = rr < a.n0 ? a.row0 + rr : a.row1 + (rr - a.n0);
        const int64_t src = a.rows ? (int64_t)a.rows[lr] : lr;
        for (int j = 0; j < d; ++j) {
            const TS* col = (const TS*)a.base + (int64_t)a.cols[j] * a.ld;
            xc[j] = (double)col[src] - a.mu[j];
        }
    }
    double nrm = 0.0;
    for (int i = 0; i < KS * 4; ++i) {
        double z = 0.0;
        if (valid && i < dm) {
            const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)i * d;
            for (int j = 0; j <= i; ++j) z = __builtin_fma(w[j], xc[j], z);
        }
        const T zt = (T)z;
        // the norm is taken from the ROUNDED coordinate so that s2(t,t) == 0 up to one rounding
        nrm = __builtin_fma((double)zt, (double)zt, nrm);
        pack[(tile * KS + (i >> 2)) * 64 + (i & 3) * 16 + idx] = zt;
    }
    double nv = -0.5 * nrm;
    if (!valid) nv = a.is_query ? 0.0 : PBN_PAD_NORM;
    if (a.fold_norm && dm < KS * 4) pack[(tile * KS + (dm >> 2)) * 64 + (dm & 3) * 16 + idx] = a.is_query ? (T)1 : (T)nv;
    if (a.is_query) {
        npack[tile * 16 + idx] = (T)nv;
    } else {
        // idx -> (lg, i) with crow(lg, i) == idx
        int lg, i;
        if (sizeof(T) == 8) { lg = idx & 3; i = idx >> 2; } else { lg = idx >> 2; i = idx & 3; }
        npack[tile * 16 + lg * 4 + i] = (T)nv;
        // weights of the WMUL sweep behind the norms: 2^norm; NaN where it would lose bits (the sweep then takes its
        // classic path for that tile), 0 for padding
        if (a.write_w) npack[a.ntiles * 16 + tile * 16 + lg * 4 + i] = !valid ? (T)0 : (nv < -1000.0 ? (T)NAN : (T)exp2(nv));
        if constexpr (sizeof(T) == 8) {
            if (a.write_r) {   // the tile's radius: sqrt(max -norm) over its 16 rows (consecutive lanes), +inf with a padding row
                double rr = (valid && nv == nv) ? -nv : INFINITY;   // (a NaN row closes its chunk: the clamped form keeps the NaN)
#pragma unroll
                for (int o = 1; o < 16; o <<= 1) { const double v = __shfl_xor(rr, o); rr = v > rr ? v : rr; }
                if (idx == 0) ((double*)npack)[a.ntiles * 32 + tile] = __builtin_sqrt(rr);
            }
        }
    }
    if (a.upack) {  // CKDE::cdf: standardised "x - b.e" of the row, in the norm's layout
        double u = 0.0;
        if (valid)
            for (int j = 0; j < d; ++j) u = __builtin_fma(a.wu[j], xc[j], u);
        T* up = (T*)a.upack;
        if (a.is_query) {
            up[tile * 16 + idx] = (T)u;
        } else {
            int lg, i;
            if (sizeof(T) == 8) { lg = idx & 3; i = idx >> 2; } else { lg = idx >> 2; i = idx & 3; }
            up[tile * 16 + lg * 4 + i] = (T)u;
        }
    }
    if (xpack) {
        double z = 0.0;
        if (valid) {
            const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)dm * d;
            for (int j = 0; j <= dm; ++j) z = __builtin_fma(w[j], xc[j], z);
        }
        const T zt = (T)z;
        const T hn = (T)(-0.5 * (double)zt * (double)zt);
        T* xp = xpack + tile * 64 + idx;
        xp[0] = zt;
        if (a.is_query) { xp[16] = (T)1; xp[32] = hn; } else { xp[16] = hn; xp[32] = (T)1; }
        xp[48] = (T)0;
    }
}

// ------------------------------------------------------------------------------------------------
// Tile pruning support (SweepArgs::prune): whitened coordinates + Morton keys of the logical rows, bounding boxes of the
// sorted 16-row tiles, and per query tile a lower bound of its queries' largest exponents.
// ------------------------------------------------------------------------------------------------
// (key cells: prune_key_bits / prune_key_cell in kde_kernels.hpp)

#define PBN_PRUNE_WINDOW 32     // training rows scanned on either side of a query's Morton position
// terms below 2^-52 of their query's largest known term are dropped: at most N * 2^-52 of a sum (2.2e-10 at 10^6 rows), a
// tenth of the error bound of the 2^x polynomial the kept terms go through.  (Round 1 and the first half of round 2 used
// 2^-64: C3's first iteration 28.2 s instead of 26.4 s, a pruned d = 2 sweep at 10^6 x 10^5 rows 15.0 ms instead of 13.6;
// 2^-44 would give 24.9 s / 12.5 ms at a worst case of 6e-8.)  PBN_PRUNE_MARGIN overrides at run time.  Since round 3 the value is
// the margin at 10^6 training rows and follows log2(n / 10^6) (prune_margin below): the BOUND is what is held constant.
#ifndef PBN_PRUNE_MARGIN
#define PBN_PRUNE_MARGIN 52.0
#endif
// fp32 (f16x2) sweeps: 2^-40.  What is dropped is at most N * 2^-40 of a sum (9e-7 at 10^6 rows, against the fp32 bar of
// 1e-3 and fp32's own 6e-8 per term); the support shrinks from 9.4 to 7.4 bandwidths per axis (a third of the tiles at 2-3
// dimensions).
#ifndef PBN_PRUNE_MARGIN_F32
#define PBN_PRUNE_MARGIN_F32 36.0   // round 4 (40 until then): N * 2^-36 = 1.5e-5 of a sum at 10^6 rows - the size of the fp32 Gram form's own error
#endif
#ifndef PBN_PRUNE_MARGIN_SUM
#define PBN_PRUNE_MARGIN_SUM 43.0   // fp64 sweeps whose result is a sum: 1.1e-7 of a sum at 10^6 rows, beside the 1.4e-7 of their 2^f (prune_margin)
#endif

// largest |z|^2 of the whitened rows (all d coordinates): one atomic max per block on the bits of a non-negative double
template <typename TS>
__global__ __launch_bounds__(256) void max_norm2_kernel(PackArgs a, unsigned long long* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double nrm = 0.0;
    if (r < a.n) {
        const int d = a.d;
        const int64_t lr = r < a.n0 ? a.row0 + r : a.row1 + (r - a.n0);
        const int64_t src = a.rows ? (int64_t)a.rows[lr] : lr;
        double xc[PBN_MAX_D];
        for (int j = 0; j < d; ++j) xc[j] = (double)((const TS*)a.base + (int64_t)a.cols[j] * a.ld)[src] - a.mu[j];
        for (int i = 0; i < d; ++i) {
            double z = 0.0;
            const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)i * d;
            for (int j = 0; j <= i; ++j) z = __builtin_fma(w[j], xc[j], z);
            nrm = __builtin_fma(z, z, nrm);
        }
        if (!(nrm == nrm)) nrm = INFINITY;   // a NaN row: as far out as it gets
    }
    for (int off = 32; off >= 1; off >>= 1) {
        const double o = __shfl_xor(nrm, off);
        nrm = o > nrm ? o : nrm;
    }
    __shared__ double wmax[4];
    if ((threadIdx.x & 63) == 0) wmax[threadIdx.x >> 6] = nrm;
    __syncthreads();
    if (threadIdx.x == 0) {
        double m = wmax[0];
        for (int w = 1; w < 4; ++w) m = wmax[w] > m ? wmax[w] : m;
        const unsigned long long bits = (unsigned long long)__double_as_longlong(m);
        if (bits > *(volatile unsigned long long*)out) atomicMax(out, bits);   // (see group_pack_train_kernel)
    }
}

// T = fragment type (the rounding the keys see), TS = element type of the table
template <typename T, typename TS = T>
__global__ __launch_bounds__(256) void prune_keys_kernel(PackArgs a, int zd, int kd, double* __restrict__ zrow, uint32_t* __restrict__ keys,
                                                         int32_t* __restrict__ iota, double inv_cell, int hilbert_nd) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.n) return;
    const int d = a.d;
    const int64_t lr = r < a.n0 ? a.row0 + r : a.row1 + (r - a.n0);
    const int64_t src = a.rows ? (int64_t)a.rows[lr] : lr;
    double xc[PBN_MAX_D];
    for (int j = 0; j < d; ++j) xc[j] = (double)((const TS*)a.base + (int64_t)a.cols[j] * a.ld)[src] - a.mu[j];
    uint32_t key = 0;
    uint32_t cells[4] = {0, 0, 0, 0};
    const int bits = prune_key_bits(kd);
    const bool curve = kd == 2 || (hilbert_nd && (kd == 3 || kd == 4));
    for (int i = 0; i < zd; ++i) {
        double z = 0.0;
        const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)i * d;
        for (int j = 0; j <= i; ++j) z = __builtin_fma(w[j], xc[j], z);
        z = (double)(T)z;   // the rounding the pack applies
        zrow[r * zd + i] = z;
        if (i < kd) {
            const double half = (double)(1 << (bits - 1)), top = (double)((1 << bits) - 1);
            double c = __builtin_floor(z * inv_cell) + half;
            c = c < 0.0 ? 0.0 : (c > top ? top : c);
            const uint32_t cell = (uint32_t)c;
            if (curve) cells[i] = cell;
            else for (int b = 0; b < bits; ++b) key |= ((cell >> b) & 1u) << (b * kd + i);   // Morton interleave
        }
    }
    if (kd == 2) {   // two key dimensions: position along the Hilbert curve (see kde_group.hip group_keys_kernel); 16 bits per axis
        uint32_t x = cells[0], y = cells[1];
        const uint32_t n1 = (1u << bits) - 1u;
        for (uint32_t sq = 1u << (bits - 1); sq > 0; sq >>= 1) {
            const uint32_t rx = (x & sq) ? 1u : 0u, ry = (y & sq) ? 1u : 0u;
            key += sq * sq * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) { x = n1 - x; y = n1 - y; }
                const uint32_t tmp = x; x = y; y = tmp;
            }
        }
    }
    if (curve && kd > 2) key = hilbert_key(cells, kd, bits);   // three / four key dimensions: the n-dimensional form of the same curve
    keys[r] = key;
    iota[r] = (int32_t)r;
}

__global__ __launch_bounds__(256) void tile_box_kernel(const double* __restrict__ zrow, const int32_t* __restrict__ perm, int64_t n, int zd, int pd,
                                                       double* __restrict__ box, double* __restrict__ zsorted) {
    // one thread per sorted row, 16 lanes per tile (one thread per TILE walked its 16 gathered rows in sequence: 57 us for the
    // 90 000 rows of a cv64 fold, next to a 250 us sweep)
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = r < n;
    double lo[PBN_PRUNE_PD], hi[PBN_PRUNE_PD];
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    if (valid) {
        const double* z = zrow + (int64_t)perm[r] * zd;
        for (int k = 0; k < zd; ++k) {
            const double v = z[k];
            zsorted[r * zd + k] = v;
#pragma unroll
            for (int j = 0; j < PBN_PRUNE_PD; ++j)
                if (j == k && j < pd && v == v) { lo[j] = v; hi[j] = v; }   // a NaN leaves the box alone, as the comparisons of the serial form did
        }
    }
    for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k) {
            const double l = __shfl_xor(lo[k], off), h = __shfl_xor(hi[k], off);
            lo[k] = l < lo[k] ? l : lo[k];
            hi[k] = h > hi[k] ? h : hi[k];
        }
    }
    if (valid && (threadIdx.x & 15) == 0) {
        const int64_t tile = r >> 4;
        for (int k = 0; k < pd; ++k) { box[tile * 2 * pd + k] = lo[k]; box[tile * 2 * pd + pd + k] = hi[k]; }
    }
}

// one thread per (sorted) query: largest exponent against the training rows around its Morton position - a valid lower
// bound of its largest term whatever those rows are - then per 16-query tile the smallest of those bounds and the box
__global__ __launch_bounds__(256) void query_prepass_kernel(const double* __restrict__ zq_row, const int32_t* __restrict__ qperm, int64_t nq,
                                                            const uint32_t* __restrict__ qkeys, const double* __restrict__ zt,
                                                            const uint32_t* __restrict__ tkeys, int64_t n, int zd, int pd,
                                                            double* __restrict__ qbox, double* __restrict__ qthr, double* __restrict__ qlb,
                                                            const double* __restrict__ subpart, int P, int which, double log2_nsub, int sum_bound,
                                                            const double* __restrict__ tile_box, int tile_window) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const bool valid = q < nq;
    double z[PBN_MAX_D];
    double best = -INFINITY;
    double sumb = -INFINITY;   // lower bound of log2 of the query's WHOLE sum: the part of it that has been looked at
    int64_t tpos_ = 0;   // the query's position in the (Morton-sorted) training order
    if (valid) {
        const double* zp = zq_row + (int64_t)qperm[q] * zd;
        for (int k = 0; k < zd; ++k) z[k] = zp[k];
        const uint32_t key = qkeys[q];
        int64_t lo = 0, hi = n;
        while (lo < hi) { const int64_t mid = (lo + hi) >> 1; if (tkeys[mid] < key) lo = mid + 1; else hi = mid; }
        tpos_ = lo;
        const int64_t b = lo - PBN_PRUNE_WINDOW > 0 ? lo - PBN_PRUNE_WINDOW : 0, e = lo + PBN_PRUNE_WINDOW < n ? lo + PBN_PRUNE_WINDOW : n;
        double acc = 0.0;
        for (int64_t t = b; t < e; ++t) {
            double d2 = 0.0;
            for (int k = 0; k < zd; ++k) { const double dd = zt[t * zd + k] - z[k]; d2 = __builtin_fma(dd, dd, d2); }
            const double ex = -0.5 * d2;
            if (ex > best) { acc = acc * exp2(best - ex) + 1.0; best = ex; }
            else acc += exp2(ex - best);
        }
        if (acc > 0.0) sumb = best + log2(acc);
        // Neighbours in Morton order are neighbours in the keyed (<= 3) dimensions only: with more dimensions than that the
        // scan above finds rows that are close in 3 coordinates and anywhere in the others - a loose bound (d = 4: 6 % of the
        // tiles pruned where 70 % could be).  The sweep over a stratified subsample of the training rows bounds the largest
        // exponent whatever the dimension: max_t s2 >= log2(sum over the subsample of 2^s2) - log2(size of the subsample).
        if (subpart) {
            const double* sp = subpart + q * P + which;
            const double ls = sp[0] + log2(sp[1]);   // log2 of the sum over the subsample: a part of the whole sum
            const double lb = ls - log2_nsub;        // ... and its mean term: a lower bound of the LARGEST term
            best = lb > best ? lb : best;
            sumb = ls > sumb ? ls : sumb;
        }
    }
    // The pruning threshold stands on the bound of the query's SUM (the scanned neighbours' terms added up, or the subsample's sum -
    // log2(nsub) = up to 12 units above its mean term): what a skipped tile could add is then below 2^-margin of the sum itself, not
    // merely of its largest term - the same "at most N 2^-margin of a sum" as before, with a radius that is 5-10 % smaller per axis.
    // (PBN_GROUP_SUM_BOUND=0 restores the largest-term threshold, here and in the grouped evaluation.)
    // reduce over the 16 lanes of a query tile
    double thr = valid ? ((sum_bound && sumb > best) ? sumb : best) : INFINITY;
    double lob[PBN_PRUNE_PD], hib[PBN_PRUNE_PD];
    for (int k = 0; k < PBN_PRUNE_PD; ++k) { lob[k] = (valid && k < pd) ? z[k] : INFINITY; hib[k] = (valid && k < pd) ? z[k] : -INFINITY; }
    for (int off = 1; off < 16; off <<= 1) {
        const double o = __shfl_xor(thr, off);
        thr = o < thr ? o : thr;
        for (int k = 0; k < PBN_PRUNE_PD; ++k) {
            const double l = __shfl_xor(lob[k], off), h = __shfl_xor(hib[k], off);
            lob[k] = l < lob[k] ? l : lob[k];
            hib[k] = h > hib[k] ? h : hib[k];
        }
    }
    // Round 4: the boxes of the training tiles around the queries' position bound their sums from below, too (see group_prepass_kernel):
    // only where the boxes cover every dimension (pd == zd)
    if (tile_box && tile_window > 0 && sum_bound && pd == zd) {
        const int l16 = threadIdx.x & 15;
        const int64_t tp0 = __shfl(valid ? tpos_ : (int64_t)0, 0, 16);
        const int64_t full = n >> 4, tt = tp0 >> 4;
        const int64_t t_lo = tt - tile_window > 0 ? tt - tile_window : 0, t_hi = tt + tile_window < full ? tt + tile_window : full;
        double bmax = -INFINITY, bacc = 0.0;
        if (lob[0] <= hib[0])
            for (int64_t t = t_lo + l16; t < t_hi; t += 16) {
                const double* bx = tile_box + t * 2 * pd;
                double d2 = 0.0;
                for (int k = 0; k < pd; ++k) {
                    const double a1 = bx[pd + k] - lob[k], a2 = hib[k] - bx[k];
                    const double a = a1 > a2 ? a1 : a2;
                    d2 = __builtin_fma(a, a, d2);
                }
                const double ex = -0.5 * d2;
                if (!(ex == ex)) continue;
                if (ex > bmax) { bacc = bacc * exp2(bmax - ex) + 1.0; bmax = ex; }
                else bacc += exp2(ex - bmax);
            }
        for (int off = 1; off < 16; off <<= 1) {
            const double om = __shfl_xor(bmax, off), oa = __shfl_xor(bacc, off);
            if (om > bmax) { bacc = bacc * exp2(bmax - om) + oa; bmax = om; }
            else if (om > -INFINITY) bacc += oa * exp2(om - bmax);
        }
        if (bacc > 0.0) {
            const double tb = bmax + log2(bacc) + 4.0;
            if (tb > thr && thr < INFINITY) thr = tb;
            if (valid && bmax > best) best = bmax;
        }
    }
    if (qlb && q < (nq + 15) / 16 * 16) qlb[q] = valid ? best : -INFINITY;   // per query: the sweep's starting offset
    if (valid && (threadIdx.x & 15) == 0) {
        const int64_t tile = q >> 4;
        qthr[tile] = thr;
        for (int k = 0; k < pd; ++k) { qbox[tile * 2 * pd + k] = lob[k]; qbox[tile * 2 * pd + pd + k] = hib[k]; }
    }
}

// ------------------------------------------------------------------------------------------------
// kde_sweep
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T max4(typename Tr<T>::vec4 v) {
    T a = v[0] > v[1] ? v[0] : v[1];
    T b = v[2] > v[3] ? v[2] : v[3];
    return a > b ? a : b;
}
template <typename T>
__device__ __forceinline__ T colmax(T v) {  // max over the 4 lanes (lane>>4 = 0..3) that share a query column
    T o = __shfl_xor(v, 16);
    v = v > o ? v : o;
    o = __shfl_xor(v, 32);
    return v > o ? v : o;
}

// XCD-aware block order (cdna_hip_programming.md T1, bijective form): workgroups are handed round-robin to the 8 XCDs, so
// `linear id % 8` labels the blocks that share an L2.  The remap gives every XCD a CONTIGUOUS range of the logical
// (split-major) grid: all blocks resident on an XCD sweep the same training split, which then lives in that XCD's 4 MB
// L2 instead of being re-fetched through the fabric by every query block.  Pure placement - results do not depend on it.
__device__ __forceinline__ void xcd_block(int& qx, int& split) {
    const unsigned gx = gridDim.x, nwg = gx * gridDim.y;
    const unsigned bid = blockIdx.x + gx * blockIdx.y;
    const unsigned xcd = bid & 7u, q = nwg >> 3, r = nwg & 7u;
    const unsigned wg = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + (bid >> 3);
    qx = (int)(wg % gx);
    split = (int)(wg / gx);
}

// Tile pruning, lane-parallel: lane l tests training tile tb + l against the box of the wave's queries (squared distance
// between the boxes puts every exponent of the tile below the wave's bound -> skip), one ballot gives the visit mask of 64
// tiles.  The sweeps then walk the set bits only: a skipped tile costs 1/64 of a test and no fragment load (the first
// version tested tile by tile on wave-uniform values - 15 DP instructions and three loads per tile, skipped or not:
// a fifth of a kept tile's cost in the fp32 sweep and ALL of a skipped tile's).
template <typename BP>
__device__ __forceinline__ unsigned long long prune_visit_mask(BP tile_box, int pd, int64_t tb, int64_t t1,
                                                               const double (&wlo)[PBN_PRUNE_PD], const double (&whi)[PBN_PRUNE_PD], double wthr, int lane) {
    const int64_t t = tb + lane;
    bool keep = false;
    if (t < t1) {
        const BP bx = tile_box + t * 2 * pd;
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) {
                const double g1 = bx[k] - whi[k], g2 = wlo[k] - bx[pd + k];
                double g = g1 > g2 ? g1 : g2;
                g = g > 0.0 ? g : 0.0;
                d2 = __builtin_fma(g, g, d2);
            }
        keep = !(-0.5 * d2 < wthr);
    }
    return __ballot(keep);
}

// The same test against ONE 16-query group's own box and bound (fp64 pruned sweeps): a wave owns QG groups, consecutive in Morton
// order, and the box of all of them is up to twice as wide per axis as a group's own - at 3-4 dimensions, where a wave's box is
// as wide as the kernel's support, a third of the (tile, group) pairs of a visited tile lie beyond the group's own support.  The
// boxes are re-read per group (uniform addresses: scalar loads; the tile's box from L1) so that no box stays in registers.
template <typename BP>
__device__ __forceinline__ unsigned long long prune_group_mask(BP tile_box, BP qbox, int pd, int64_t tb, int64_t t1, double thr, int lane) {
    const int64_t t = tb + lane;
    bool keep = false;
    if (t < t1) {
        const BP bx = tile_box + t * 2 * pd;
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) {
                const double g1 = bx[k] - qbox[pd + k], g2 = qbox[k] - bx[pd + k];
                double g = g1 > g2 ? g1 : g2;
                g = g > 0.0 ? g : 0.0;
                d2 = __builtin_fma(g, g, d2);
            }
        keep = !(-0.5 * d2 < thr);
    }
    return __ballot(keep);
}

// ... with a second, nearer threshold: `near` = the tiles that hold a term above thr_near (the others of the returned mask are the
// far tiles of the fp32 tail path)
template <typename BP>
__device__ __forceinline__ unsigned long long prune_group_mask2(BP tile_box, BP qbox, int pd, int64_t tb, int64_t t1, double thr, double thr_near, int lane,
                                                                unsigned long long& near) {
    const int64_t t = tb + lane;
    bool keep = false, kn = false;
    if (t < t1) {
        const BP bx = tile_box + t * 2 * pd;
        double d2 = 0.0;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) {
                const double g1 = bx[k] - qbox[pd + k], g2 = qbox[k] - bx[pd + k];
                double g = g1 > g2 ? g1 : g2;
                g = g > 0.0 ? g : 0.0;
                d2 = __builtin_fma(g, g, d2);
            }
        keep = !(-0.5 * d2 < thr);
        kn = !(-0.5 * d2 < thr_near);
    }
    near = __ballot(kn);
    return __ballot(keep);
}

// One uniform test per (64-tile batch, query group): does the batch's box come within the drop threshold of the group's box at all?
template <typename BP>
__device__ __forceinline__ bool batch_in_reach(BP bb, BP qbox, int pd, double thr) {
    double d2 = 0.0;
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k)
        if (k < pd) {
            const double g1 = bb[k] - qbox[pd + k], g2 = qbox[k] - bb[pd + k];
            double g = g1 > g2 ? g1 : g2;
            g = g > 0.0 ? g : 0.0;
            d2 = __builtin_fma(g, g, d2);
        }
    return !(-0.5 * d2 < thr);
}

// GUARD of the pruned sum-only fp64 sweeps (round 6): is the LARGEST squared distance between the batch's box and the group's box at most
// PBN_OPEN_FAR2?  With boxes over all whitened dimensions every exponent of the batch's rows against the group's queries then lies at most
// PBN_OPEN_FAR2 / 2 below the offset, and exp2_magic needs no clamp for the batch (false for a box with a NaN or infinite side).
#define PBN_OPEN_FAR2 2200.0
template <typename BP>
__device__ __forceinline__ bool batch_all_near(BP bb, BP qbox, int pd) {
    double f2 = 0.0;
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k)
        if (k < pd) {
            const double h1 = bb[pd + k] - qbox[k], h2 = qbox[pd + k] - bb[k];
            const double h = h1 > h2 ? h1 : h2;
            f2 = __builtin_fma(h, h, f2);
        }
    return f2 <= PBN_OPEN_FAR2;
}

// ... and with the MOMENT pass (round 5): `mom` = the (tile, group) pairs whose contribution is taken from the tile's moments instead
// (kde_moment_group_kernel).  The tile's rows are z_t = c + delta_t, |delta_t| <= rho; for a query at u = z_q - c a row's term is
// 2^(-|u|^2 / 2) 2^(-|delta_t|^2 / 2) e^(s_t), s_t = a u.delta_t, a = ln 2, and e^s is replaced by its Taylor polynomial T_P(s).  By Lagrange's
// remainder |e^s - T_P(s)| <= |s|^(P+1) / (P+1)! max(1, e^s), so the row's error is at most |s|^(P+1) / (P+1)! times the larger of its own term
// and 2^(-|u|^2 / 2 - |delta_t|^2 / 2) - and BOTH are at most 2^(-d2min / 2), d2min the smallest distance between the tile's and the group's box (the
// centroid lies in the tile's box).  With |s| <= w = a |u|max rho (|u|max: the largest distance between the two boxes) a tile whose terms lie 2^E below the
// group's sum bound may therefore be expanded when E + log2(w^(P+1) / (P+1)!) <= -(margin + PBN_MOM_EXTRA): all expanded tiles together then err by at
// most N 2^-(margin + extra) of a sum - a quarter of what pruning may drop.  The bound is proved, the realised error is 5-6 orders smaller
// (tools/moment_prototype.py: 2e-13 of a sum on C3's folds).  Near tiles qualify through small |u|, far ones through small terms; the middle
// distances are what stays with the sweep.  Both kernels classify with this one function on the same inputs, so every (tile, group) pair is
// taken by exactly one of them.
#ifndef PBN_MOM_EXTRA
#define PBN_MOM_EXTRA 2.0
#endif
template <typename BP, typename RP>
__device__ __forceinline__ unsigned long long prune_group_mask3(BP tile_box, BP qbox, RP rad2, int pd, int64_t tb, int64_t t1, double thr, double thr_near,
                                                                double thr_mom, int lane, unsigned long long& near, unsigned long long& mom) {
#pragma clang fp contract(off)   // both kernels must take bit-identical decisions: no fused multiply-adds the inliner could place differently
    const int64_t t = tb + lane;
    bool keep = false, kn = false, km = false;
    if (t < t1) {
        const BP bx = tile_box + t * 2 * pd;
        double d2 = 0.0, f2 = 0.0;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) {
                const double g1 = bx[k] - qbox[pd + k], g2 = qbox[k] - bx[pd + k];
                double g = g1 > g2 ? g1 : g2;
                g = g > 0.0 ? g : 0.0;
                d2 = __builtin_fma(g, g, d2);   // (explicit fmas are kept as written)
                const double h1 = bx[pd + k] - qbox[k], h2 = qbox[pd + k] - bx[k];
                const double h = h1 > h2 ? h1 : h2;
                f2 = __builtin_fma(h, h, f2);
            }
        const double ex = -0.5 * d2;
        keep = !(ex < thr);
        kn = !(ex < thr_near);
        // w = ln 2 * |u|max * rho, rounded up; log2(w^9 / 9!) = 9 log2 w - log2 9! (v_sqrt_f32 / v_log_f32: 1 ulp, covered by the + 0.02)
        const float w = 0.69314724f * __builtin_amdgcn_sqrtf((float)f2 * 1.000001f * rad2[t]) * 1.000001f;
        const float logr = (float)(PBN_MOM_ORDER + 1) * __builtin_amdgcn_logf(w) - PBN_MOM_LOG2_FACT + 0.02f;
        km = keep && ((double)logr + ex <= thr_mom);
    }
    near = __ballot(kn);
    mom = __ballot(km);
    return __ballot(keep);
}

// WMUL (d mod 4 == 0, no free K slot for the norm): the training norms enter as WEIGHTS.  The accumulator starts from the
// per-query constant alone (a persistent register quad as the MFMA's C operand, as with FOLD) and holds
// x' = z_t.z_q - 1/2|z_q|^2 - m_q + bias; the term is 2^x' * w_t with w_t = 2^(-1/2|z_t|^2) precomputed by the pack kernel, and
// the multiply rides in the running sum's FMA: no add per value for the norm.  x' >= 0 for every term that matters (x' >=
// x' - 1/2|z_t|^2 >= 0), so the v_fract form still applies; x' can exceed the exponent range only when z_t.z_q is huge (a
// far-out query next to a far-out training row): 2^x' = inf (times w = 0: NaN) fails the per-tile test `ts < big`, and the
// rare path redoes the tile the classic way (norms added to the accumulator).  Rows with -1/2|z|^2 < -1000, whose weight
// would lose bits or underflow, carry w = NaN and always take that path.
// Pruned sweeps run ONE wave per workgroup: the kept tiles differ from wave to wave (each has its own query box), and a
// 4-wave workgroup holds its slots until its slowest wave is done - measured at 0.67-0.77 of the unpruned rate per visited
// tile at d = 2, 3 whatever the number of splits (tools/prune_visits.py); nothing in the kernel is shared between waves.
constexpr int sweep_block_threads(bool prune) { return prune ? 64 : 256; }
// The grid of a pruned sweep is one-dimensional and split-major (the workgroups in flight share a split's fragments in L2),
// with the splits taken from both ends of the Morton order inwards: the corner splits - sparse regions, where a query's
// whole neighbourhood lies in its own split and the workgroup visits nearly all of its tiles - start first.  Pure placement.
// (Measured and dropped: every query group's nearest splits first - all splits in flight at once, the L2 sharing is gone.)
__device__ __forceinline__ void pruned_block(const SweepArgs& a, int groups_per_block, unsigned b, int& qx, int& split) {
    const unsigned Gq = (unsigned)((a.nqtiles + groups_per_block - 1) / groups_per_block), Gs = (unsigned)a.nsplit_grid;
    const unsigned k = b / Gq;
    qx = (int)(b % Gq);
    split = (k & 1u) ? (int)(Gs - 1 - (k >> 1)) : (int)(k >> 1);
}

// The sweep proper.  `bid` is the workgroup's index inside ITS sweep: blockIdx.x for a stand-alone launch, the offset inside
// the unit for the grouped launches (kde_sweep_group_kernel), where `a` was assembled from the unit's record.
// EF32: 2^f of the main loop on the fp32 transcendental unit (exp2_f64_fract<true>; sweeps whose result is a sum)
// MOM (round 5): the moment pass runs beside this sweep (grouped fp64 sum-only launches of one- and two-variable units) - the sweep skips the
// pairs the pass takes (prune_group_mask3).  A template parameter, not a run-time branch:
// the kernel sits at its register limit and the extra paths cost the plain sweep 25 % when compiled in.
template <typename T, int KS, bool COND, int QG, bool FOLD, bool PRUNE, bool WMUL, bool EF32 = false, bool MOM = false>
__device__ __forceinline__ void kde_sweep_body(const SweepArgs& a, const unsigned bid) {
    static_assert(!WMUL || (!FOLD && !COND), "WMUL: plain sweeps without a free K slot only");
    using V = typename Tr<T>::vec4;
    // MAGIC: the accumulators of this sweep carry Tr<T>::magic() and 2^x is exp2_magic - unpruned sum-only fp64 sweeps (round 6)
    constexpr bool MAGIC = PBN_EXP2_MAGIC && EF32 && PBN_EXP2_F32 && PBN_EXP2_DEGREE <= 7 && sizeof(T) == 8 && !COND && (!PRUNE || PBN_MAGIC_PRUNED);
    // the per-query constant of the accumulators: norm (+ the magic constant, rounded to ITS grid once per query: what follows - the
    // integer offset, the bias - is exact), and 2^x of an accumulator
    auto cbase = [](T nyq) -> T { return MAGIC ? (T)(nyq + Tr<T>::magic()) : nyq; };
    auto ex2a = [](T v, T top) -> T {
        if constexpr (MAGIC) { (void)top; return Tr<T>::template ex2m<PBN_MAGIC_CLAMP != 0>(v); }
        else return Tr<T>::template ex2p<EF32>(v, top);
    };
    constexpr int WPB = sweep_block_threads(PRUNE) / 64;   // waves per workgroup
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lg = lane >> 4;
    int qx, split;
    if (PRUNE) pruned_block(a, WPB * QG, bid, qx, split);
    else xcd_block(qx, split);
    const int64_t qt0 = ((int64_t)qx * WPB + wave) * QG;
    if (qt0 >= a.nqtiles) return;  // no barriers in this kernel: idle waves just leave
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;

    const PBN_GLOBAL T* __restrict__ Ap = (const PBN_GLOBAL T*)a.Apack;
    const PBN_GLOBAL T* __restrict__ Np = (const PBN_GLOBAL T*)a.nxpack;
    const PBN_GLOBAL T* __restrict__ Wp = Np + a.ntiles * 16;   // WMUL: weights 2^norm behind the norms (PackArgs::write_w)
    const PBN_GLOBAL T* __restrict__ Xp = (const PBN_GLOBAL T*)a.Axpack;
    const PBN_GLOBAL T* __restrict__ Bp = (const PBN_GLOBAL T*)a.Bpack;
    const PBN_GLOBAL T* __restrict__ NYp = (const PBN_GLOBAL T*)a.nypack;
    const PBN_GLOBAL T* __restrict__ BXp = (const PBN_GLOBAL T*)a.Bxpack;
    const PBN_GLOBAL double* __restrict__ TBp = (const PBN_GLOBAL double*)a.tile_box;
    const PBN_GLOBAL double* __restrict__ QBp = (const PBN_GLOBAL double*)a.qtile_box;
    const PBN_GLOBAL double* __restrict__ QTp = (const PBN_GLOBAL double*)a.qtile_thr;
    const PBN_GLOBAL double* __restrict__ QLp = (const PBN_GLOBAL double*)a.qlb;

    // ---- query-side fragments and per-query state -------------------------------------------
    T b[QG][KS];
    T ny[QG], cm[QG], m[QG];
    V cmv[QG];  // FOLD: the accumulator's start value, cm in all four rows (the training norms ride in a K slot)
    double sum[QG];
    T bxb[QG], bx[QG], mj[QG];  // CKDE: extra-step B fragment (base / current), joint offset
    double sumj[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) b[g][ks] = Bp[(qt * KS + ks) * 64 + lane];
        ny[g] = NYp[qt * 16 + (lane & 15)];
        sum[g] = 0.0;
        if (COND) { bxb[g] = BXp[qt * 64 + lane]; sumj[g] = 0.0; }
    }

    const T ctop = Tr<T>::top();   // leading exp2 coefficient pinned in a VGPR for the whole kernel

    // ---- tile pruning: box of this wave's queries and the exponent below which a training tile cannot matter -------
    double wlo[PBN_PRUNE_PD] = {}, whi[PBN_PRUNE_PD] = {}, wthr = 0;
    const int pd = PRUNE ? a.pdims : 0;
    if (PRUNE) {
        wthr = INFINITY;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k) { wlo[k] = INFINITY; whi[k] = -INFINITY; }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
            const double th = QTp[qt];
            wthr = th < wthr ? th : wthr;
#pragma unroll
            for (int k = 0; k < PBN_PRUNE_PD; ++k)
                if (k < pd) {
                    const double l = QBp[qt * 2 * pd + k], h = QBp[qt * 2 * pd + pd + k];
                    wlo[k] = l < wlo[k] ? l : wlo[k];
                    whi[k] = h > whi[k] ? h : whi[k];
                }
        }
        wthr -= a.prune_margin;
    }
    // ---- prologue: offsets from the first tile (max of s2 over its 16 rows) ---------------------
    {
        T af[KS];
        V nx;
        T ax = 0;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) af[ks] = Ap[(t0 * KS + ks) * 64 + lane];
        if (!FOLD) nx = *(const PBN_GLOBAL V*)(Np + t0 * 16 + lg * 4);
        if (COND) ax = Xp[t0 * 64 + lane];
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            V acc = FOLD ? V{ny[g], ny[g], ny[g], ny[g]} : nx + ny[g];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = Tr<T>::mfma(af[ks], b[g][ks], acc);
            // The offsets are INTEGERS (base-2 units): then fract(s2 - m + bias) = fract(s2), so the argument of the 2^f
            // polynomial - and with it its ~2e-9 approximation error - belongs to the (training row, query) pair and not to
            // the offset, i.e. not to the split, the tile order or the pruning: sums taken in a different partition agree to
            // rounding (1e-13), not to the polynomial's error bound.  Scaling by 2^integer is exact.
            T mx = __builtin_ceil(colmax<T>(max4<T>(acc)));
            m[g] = mx;
            cm[g] = cbase(ny[g]) - mx + Tr<T>::bias();   // main-loop exponents are kept biased (Tr<T>::ex2p)
            if (FOLD || WMUL) cmv[g] = V{cm[g], cm[g], cm[g], cm[g]};
            if (COND) {
                V accj = Tr<T>::mfma(ax, bxb[g], acc);
                T mxj = __builtin_ceil(colmax<T>(max4<T>(accj)));
                mj[g] = mxj;
                bx[g] = (lg == 2) ? bxb[g] + (m[g] - mj[g]) : bxb[g];
            }
        }
    }

    // Pruned plain sweeps: the prepass knows a lower bound of every query's largest exponent over ALL training rows (its
    // Morton neighbours, the subsample sweep).  Where that bound lies above the first tile's maximum it becomes the offset: the
    // first tile of a Morton-ordered split is typically thousands of exponent units away from the queries, so the first
    // VISITED tile used to overflow 2^x and take the redo path once per (group, split); with the bound as the offset no term
    // can exceed it by more than the bound's slack (log2 of the subsample size, or the distance to the best Morton
    // neighbour), and the blind passes below never trip.  Such an offset has no term of this split behind it: an empty sum
    // stays empty (the query's largest term is never pruned, so the split that holds it has a positive sum).
    // (fused CKDE sweeps: the bound is the JOINT one, which also bounds the marginal maximum from below.)
    bool lbm[QG], lbmj[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) lbm[g] = lbmj[g] = false;
    if constexpr (PRUNE) {
        if (a.qlb) {
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
                const T lb = __builtin_ceil((T)QLp[qt * 16 + (lane & 15)]);
                // (a bound too large for T to hold to a fraction of a unit is not used: see kde_sweep_f16_kernel)
                const bool fin = (lb < (T)0 ? -lb : lb) < (sizeof(T) == 8 ? (T)0x1p50 : (T)0x1p22);
                lbm[g] = fin && lb > m[g];
                if (lbm[g]) {
                    m[g] = lb;
                    cm[g] = cbase(ny[g]) - lb + Tr<T>::bias();
                }
                if (FOLD || WMUL) cmv[g] = V{cm[g], cm[g], cm[g], cm[g]};
                if (COND) {
                    lbmj[g] = fin && lb > mj[g];
                    if (lbmj[g]) mj[g] = lb;
                    bx[g] = (lg == 2) ? bxb[g] + (m[g] - mj[g]) : bxb[g];
                }
            }
        }
    }

    // ---- main loop over training tiles: two tiles per iteration with ping-pong fragment buffers (no
    // register copies); the fragments of tile t+1 / t+2 are in flight while tile t is processed -------------
    auto load_tile = [&](int64_t t, T (&f)[KS], V& n, T& x) {
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) f[ks] = Ap[(t * KS + ks) * 64 + lane];
        if (!FOLD) n = *(const PBN_GLOBAL V*)((WMUL ? Wp : Np) + t * 16 + lg * 4);
        if (COND) x = Xp[t * 64 + lane];
    };
    // pruned plain fp64 sweeps: visit masks per query group (GMASK), bit `bit` of gm[g] = group g needs this tile
    constexpr bool GMASK = PRUNE && !COND && sizeof(T) == 8;
    unsigned long long gm[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) gm[g] = ~0ull;
    // FARP (round 4; sum-only pruned sweeps of the FOLD shapes, per-group masks): a tile ALL of whose terms lie more than
    // prune_margin - far_span (26 at 10^6 rows) below the group's sum bound contributes less than 2^-26 of the sum per term: its 2^x go
    // through the fp32 unit directly - cvt_f32_f64, v_exp_f32, v_add_f32: 4 issue slots per value instead of 8 - and are added in fp32
    // until the end of the blind batch.  The exponents are biased by +128 and the offsets lie within 6 units of the sum bound (the
    // prepass bounds), so x' <= 108: no overflow, and 2^x' carries the same 2^bias as the fp64 sums.  Relative error of such a term
    // 2^-24 |x'| ln 2 <= 5.3e-6, of the sum <= 5.3e-6 N 2^-26 = 8e-8 (the margin follows log2(N / 10^6), so the bound does not depend
    // on N).  A batch redone by the checked loop takes every tile through the full path.
    constexpr bool FARP = GMASK && FOLD && EF32 && PBN_FAR_F32;
    static_assert(!MOM || FARP, "the moment pass stands beside the FARP shapes only");
    unsigned long long gn[FARP ? QG : 1];
    float fs[FARP ? QG : 1];
#pragma unroll
    for (int g = 0; g < (FARP ? QG : 1); ++g) { gn[g] = ~0ull; fs[g] = 0.f; }
    // GUARDP (round 6): pruned MAGIC sweeps with the norm in a K slot and boxes over all whitened dimensions (d <= 3) - a 64-tile batch whose
    // box lies within PBN_OPEN_FAR2 of the boxes of the wave's groups runs exp2_magic without its clamp (batch_all_near, one lane-parallel
    // test per super-batch beside batch_in_reach: nothing per tile); gate_of = the group's side of the proof: offsets inside [-890, 16] (an
    // exponent is at most bias - offset), no NaN query, not the query tile with the padding rows
    constexpr bool GUARDP = MAGIC && PRUNE && GMASK && FOLD && KS == 1 && PBN_MAGIC_GUARD;
    // (taken per super-batch, from the offsets as they stand: a scalar carried around the whole walk - even one computed once before it - pushed
    // the visit masks out of the SGPRs into vector registers and lane-masked branches: 25 % of the kernel's time)
    auto gate_of = [&](int g) -> bool {
        return __builtin_amdgcn_readfirstlane(__all(a.box_full && qt0 + g < a.nqtiles - 1 && ny[g] == ny[g] && m[g] >= (T)-890 && m[g] <= (T)16)) != 0;
    };
    auto process_tile = [&](const int64_t t, const T (&af)[KS], const V& nx, const T ax, const int bit) {
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            if (GMASK && !((gm[g] >> bit) & 1ull)) continue;
            V acc;
            if (FOLD || WMUL) acc = cmv[g]; else acc = nx + cm[g];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = Tr<T>::mfma(af[ks], b[g][ks], acc);
            V accj;
            if (COND) accj = Tr<T>::mfma(ax, bx[g], acc);

            T e0 = ex2a(acc[0], ctop), e1 = ex2a(acc[1], ctop), e2 = ex2a(acc[2], ctop), e3 = ex2a(acc[3], ctop);
            T ts;
            if (WMUL) ts = __builtin_fma(e3, nx[3], __builtin_fma(e2, nx[2], __builtin_fma(e1, nx[1], e0 * nx[0])));   // nx holds the weights
            else ts = (e0 + e1) + (e2 + e3);
            T tsj = 0;
            bool bad = !(ts < Tr<T>::big());
            if (COND) {
                T j0 = ex2a(accj[0], ctop), j1 = ex2a(accj[1], ctop), j2 = ex2a(accj[2], ctop), j3 = ex2a(accj[3], ctop);
                tsj = (j0 + j1) + (j2 + j3);
                bad = bad || !(tsj < Tr<T>::big());
            }
            if (__builtin_expect(__any(bad), 0)) {
                // Rare wave-uniform slow path: raise the offsets to the tile maximum and redo the tile.
                if (WMUL) acc += *(const PBN_GLOBAL V*)(Np + t * 16 + lg * 4);   // the classic exponents: norms added
                T mx = __builtin_ceil(colmax<T>(max4<T>(acc)) - (MAGIC ? Tr<T>::magic() : (T)0) - Tr<T>::bias());
                if (mx > (T)0) {
                    m[g] += mx;
                    cm[g] = cbase(ny[g]) - m[g] + Tr<T>::bias();
                    if (FOLD || WMUL) cmv[g] = V{cm[g], cm[g], cm[g], cm[g]};
                    sum[g] *= exp2(-(double)mx);
                    acc -= mx;
                }
                // the SAME 2^x as the main loop (the exponents are still biased, the offsets integers): a term must come out
                // identical whichever path evaluates it, or sums taken in another tile order would differ by the polynomial's
                // error (the Morton-ordered sweeps come through here often, table-ordered ones hardly ever)
                e0 = ex2a(acc[0], ctop); e1 = ex2a(acc[1], ctop); e2 = ex2a(acc[2], ctop); e3 = ex2a(acc[3], ctop);
                ts = (e0 + e1) + (e2 + e3);
                if (COND) {
                    T mxj = __builtin_ceil(colmax<T>(max4<T>(accj)) - Tr<T>::bias());
                    if (mxj > (T)0) {
                        mj[g] += mxj;
                        sumj[g] *= exp2(-(double)mxj);
                        accj -= mxj;
                    }
                    bx[g] = (lg == 2) ? bxb[g] + (m[g] - mj[g]) : bxb[g];
                    T j0 = ex2a(accj[0], ctop), j1 = ex2a(accj[1], ctop), j2 = ex2a(accj[2], ctop), j3 = ex2a(accj[3], ctop);
                    tsj = (j0 + j1) + (j2 + j3);
                }
            }
            sum[g] += (double)ts;
            if (COND) sumj[g] += (double)tsj;
        }
    };

    T afA[KS], afB[KS];
    V nxA, nxB;
    T axA = 0, axB = 0;
    // blind accumulation (see "Unchecked passes" below): no overflow test, no separate add
    auto process_fast = [&](const T (&af)[KS], const V& nx, const int bit, auto clamp) {
        constexpr int CLAMPED = decltype(clamp)::value;   // 0: exp2_magic without its clamp (the chunk's / batch's exponents are proven inside +-1022)
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            if (GMASK && !((gm[g] >> bit) & 1ull)) continue;
            V acc;
            if (FOLD || WMUL) acc = cmv[g]; else acc = nx + cm[g];
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) acc = Tr<T>::mfma(af[ks], b[g][ks], acc);
            if constexpr (FARP) {
                if (!((gn[g] >> bit) & 1ull)) {   // a far tile of this group: the fp32 tail path
                    if constexpr (MAGIC) acc -= Tr<T>::magic();   // (exact: the accumulator sits on the constant's grid)
                    const float f0 = __builtin_amdgcn_exp2f((float)acc[0]), f1 = __builtin_amdgcn_exp2f((float)acc[1]);
                    const float f2 = __builtin_amdgcn_exp2f((float)acc[2]), f3 = __builtin_amdgcn_exp2f((float)acc[3]);
                    fs[g] += (f0 + f1) + (f2 + f3);
                    continue;
                }
            }
            T e0, e1, e2, e3;
            if constexpr (MAGIC && CLAMPED == 0) {
                e0 = Tr<T>::template ex2m<false>(acc[0]); e1 = Tr<T>::template ex2m<false>(acc[1]); e2 = Tr<T>::template ex2m<false>(acc[2]); e3 = Tr<T>::template ex2m<false>(acc[3]);
            } else {
                e0 = ex2a(acc[0], ctop); e1 = ex2a(acc[1], ctop); e2 = ex2a(acc[2], ctop); e3 = ex2a(acc[3], ctop);
            }
            if (WMUL) sum[g] = __builtin_fma(e3, nx[3], __builtin_fma(e2, nx[2], __builtin_fma(e1, nx[1], __builtin_fma(e0, nx[0], sum[g]))));
            else sum[g] += (e0 + e1) + (e2 + e3);
        }
    };
    if constexpr (PRUNE) {
        // 64 tiles per visit mask; inside a batch the kept tiles are processed two at a time with ping-pong fragment buffers
        auto run_batch = [&](int64_t tb, unsigned long long mask, auto blind) {
            constexpr int BMODE = decltype(blind)::value;   // 0: checked, 1: blind, 2: blind and without the clamp of exp2_magic (GUARDP)
            constexpr bool BLIND = BMODE != 0;
            // The prefetch of the next kept tile is UNCONDITIONAL (after the last one the current tile is simply loaded again):
            // with `if (mask) load` the two paths into the next MFMA differ in their number of loads in flight, and the compiler
            // must wait for ALL of them (s_waitcnt vmcnt(0)) - i.e. for the prefetch it has just issued - before every tile.
            int b = __builtin_ctzll(mask);
            mask &= mask - 1;
            load_tile(tb + b, afA, nxA, axA);
            for (;;) {
                const bool more = mask != 0;
                const int b2 = more ? __builtin_ctzll(mask) : b;
                mask &= mask - 1;
                load_tile(tb + b2, afB, nxB, axB);
                if constexpr (BLIND) process_fast(afA, nxA, b, std::integral_constant<int, BMODE == 2 ? 0 : 1>{}); else process_tile(tb + b, afA, nxA, axA, b);
                if (!more) break;
                const bool more2 = mask != 0;
                const int b3 = more2 ? __builtin_ctzll(mask) : b2;
                mask &= mask - 1;
                load_tile(tb + b3, afA, nxA, axA);
                if constexpr (BLIND) process_fast(afB, nxB, b2, std::integral_constant<int, BMODE == 2 ? 0 : 1>{}); else process_tile(tb + b2, afB, nxB, axB, b2);
                if (!more2) break;
                b = b3;
            }
        };
        // fp64 plain sweeps take a batch blind first (the offsets start from the prepass bounds: an overflow needs a term 896
        // exponent units above its query's bound) and redo it with the checked loop from the saved sums if a sum went bad
        constexpr bool FASTP = PBN_SWEEP_UNCHECKED && !COND && sizeof(T) == 8 && (FOLD || KS == 1);   // the shapes that stay <= 168 VGPRs
        if (a.count_redo && lane == 0) atomicAdd(&g_sweep_tiles, (unsigned long long)(t1 - t0) * QG);
        // Two levels: a SUPER-BATCH of 64 batches (4096 tiles) is classified first, lane = batch, against the batches' own boxes (grouped
        // sweeps: GSweepUnit::batch_box) - one round trip to L2 for 64 batches instead of one per batch, which is what the walk over a
        // split's tiles costs where most batches hold nothing for the wave (the test below is latency, not arithmetic).
        auto do_batch = [&](const int64_t tb, const unsigned gsel, const bool bare) {
            unsigned long long mask;
            if constexpr (GMASK) {
                mask = 0;
                if (MOM || a.group_masks) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
                        if (!((gsel >> g) & 1u)) {   // the whole batch beyond this group's reach
                            gm[g] = 0;
                            if constexpr (FARP) gn[g] = ~0ull;
                            continue;
                        }
                        if constexpr (FARP) {
                            if constexpr (MOM) {   // the moment pass takes the pairs it can expand: this sweep skips them
                                unsigned long long mx;
                                gm[g] = prune_group_mask3(TBp, QBp + qt * 2 * pd, (const PBN_GLOBAL float*)a.tile_rad2, pd, tb, t1, QTp[qt] - a.prune_margin,
                                                                  a.far_span > 0.0 ? QTp[qt] - (a.prune_margin - a.far_span) : -INFINITY,
                                                                  QTp[qt] - (a.prune_margin + PBN_MOM_EXTRA), lane, gn[g], mx);
                                gm[g] &= ~mx;
                            } else if (a.far_span > 0.0) {
                                gm[g] = prune_group_mask2(TBp, QBp + qt * 2 * pd, pd, tb, t1, QTp[qt] - a.prune_margin, QTp[qt] - (a.prune_margin - a.far_span), lane, gn[g]);
                            } else {
                                gm[g] = prune_group_mask(TBp, QBp + qt * 2 * pd, pd, tb, t1, QTp[qt] - a.prune_margin, lane);
                                gn[g] = ~0ull;
                            }
                        } else {
                            gm[g] = prune_group_mask(TBp, QBp + qt * 2 * pd, pd, tb, t1, QTp[qt] - a.prune_margin, lane);
                        }
                        mask |= gm[g];
                    }
                } else {
                    mask = prune_visit_mask(TBp, pd, tb, t1, wlo, whi, wthr, lane);
#pragma unroll
                    for (int g = 0; g < QG; ++g) gm[g] = mask;
                    if constexpr (FARP) {
#pragma unroll
                        for (int g = 0; g < QG; ++g) gn[g] = ~0ull;
                    }
                }
            } else {
                mask = prune_visit_mask(TBp, pd, tb, t1, wlo, whi, wthr, lane);
            }
            if (!mask) return;
            if (a.count_redo && lane == 0) {
                unsigned long long v = 0;
#pragma unroll
                for (int g = 0; g < QG; ++g) v += (unsigned long long)__builtin_popcountll(GMASK ? gm[g] : mask);
                atomicAdd(&g_sweep_visit, v);   // (tile, group) pairs visited, out of QG x tiles offered
            }
            if constexpr (FASTP) {
                double saved[QG];
#pragma unroll
                for (int g = 0; g < QG; ++g) saved[g] = sum[g];
                // a proven batch runs the blind loop without the clamp, any other (2-3 % on the bench tables: wide batch boxes, the last tile and
                // the last query tile of a unit, query groups whose offsets lie beyond -890) the SAME loop with it: the same instructions but one,
                // the same order of additions, the far tiles' fp32 tail in both - a batch's sum does not depend on whether it was proven
                // (PBN_MAGIC_GUARD=0 takes every batch through the second: bit-identical scores, tests/test_magic_exp2_gpu.py)
                if (GUARDP && bare) run_batch(tb, mask, std::integral_constant<int, 2>{});
                else run_batch(tb, mask, std::integral_constant<int, 1>{});
                if constexpr (FARP) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) { sum[g] += (double)fs[g]; fs[g] = 0.f; }   // (an overflowed fp32 tail arrives as inf: the batch is redone)
                }
                bool bad = false;
#pragma unroll
                for (int g = 0; g < QG; ++g) bad = bad || !(sum[g] < 0x1p1000);
                const bool redo = __any(bad);
                if (a.count_redo && lane == 0) { atomicAdd(&g_sweep_units, 1ull); if (redo) atomicAdd(&g_sweep_redo, 1ull); }
                if (redo) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) sum[g] = saved[g];
                    run_batch(tb, mask, std::integral_constant<int, 0>{});
                }
            } else {
                run_batch(tb, mask, std::integral_constant<int, 0>{});
            }
        };
        for (int64_t sb = t0; sb < t1; sb += 4096) {
            const int64_t bt = sb + 64 * lane;   // my batch's first tile
            unsigned long long bm = __ballot(bt < t1), bmg[QG];
            unsigned long long bopen = (GUARDP && (MOM || a.group_masks) && a.batch_box) ? ~0ull : 0ull;
#pragma unroll
            for (int g = 0; g < QG; ++g) bmg[g] = bm;
            if constexpr (GMASK) {
                if ((MOM || a.group_masks) && a.batch_box) {
                    const PBN_GLOBAL double* bb = (const PBN_GLOBAL double*)a.batch_box + ((int64_t)split * a.batches_per_split + ((bt - t0) >> 6)) * 2 * pd;
                    bm = 0;
#pragma unroll
                    for (int g = 0; g < QG; ++g) {
                        const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
                        bmg[g] = __ballot(bt < t1 && batch_in_reach(bb, QBp + qt * 2 * pd, pd, QTp[qt] - a.prune_margin));
                        bm |= bmg[g];
                        // open for the wave = proven for every group that reaches the batch (the batch with the table's last tile - padding
                        // rows, whose norm slot is not a distance - never)
                        if constexpr (GUARDP) bopen &= ~bmg[g] | (gate_of(g) ? __ballot(bt + 64 < a.ntiles && batch_all_near(bb, QBp + qt * 2 * pd, pd)) : 0ull);
                    }
                }
            }
            while (bm) {
                const int j = __builtin_ctzll(bm);
                bm &= bm - 1;
                unsigned gsel = 0;
#pragma unroll
                for (int g = 0; g < QG; ++g) gsel |= (unsigned)((bmg[g] >> j) & 1ull) << g;
                do_batch(sb + 64 * (int64_t)j, gsel, GUARDP && ((bopen >> j) & 1ull));
            }
        }
    } else {
        // Unchecked passes (fp64 plain sweeps): the per-(tile, group) overflow test and the separate add into the running sum are
        // 2 of the 42 DP instructions.  2^x overflows only when an exponent lies 896 above the offsets (a query whose neighbours
        // are ~25 bandwidths closer than the rows that set its offset) or, with the norms as weights, when z_t.z_q alone is
        // that large (a far-out row next to a far-out query; rows beyond |z|^2 = 2000 carry a NaN weight on purpose).  So the
        // sums are accumulated blind (weights riding in the FMA chain, or plain adds) and looked at once per chunk of 32 tiles:
        // a wave that finds an infinite / NaN sum restores the sums it saved in LDS at the start of the chunk and redoes the
        // chunk with the checked loop - 32 tiles, not the split: an outlier row costs its chunk, not every query block of its
        // split (measured with whole-split redo on the C2 data, whose diagonal bandwidths leave 5 rows beyond |z|^2 = 1780: 12 % of
        // the units redone, all on the XCDs that own those splits - 76 ms instead of 50).
        // (only where the second loop body leaves the kernel at 3 waves / SIMD: <= 168 VGPRs)
        constexpr bool FAST = PBN_SWEEP_UNCHECKED && !COND && sizeof(T) == 8 && (WMUL || (FOLD && KS <= 3) || (!FOLD && KS <= 2));
        auto checked_range = [&](int64_t ta, int64_t tb) {
            load_tile(ta, afA, nxA, axA);
            for (int64_t t = ta; t < tb; t += 2) {
                const bool second = t + 1 < tb;                       // wave-uniform
                load_tile(second ? t + 1 : t, afB, nxB, axB);
                process_tile(t, afA, nxA, axA, 0);
                load_tile(t + 2 < tb ? t + 2 : t, afA, nxA, axA);
                if (second) process_tile(t + 1, afB, nxB, axB, 0);
            }
        };
        if constexpr (FAST) {
            constexpr int CH = 32;
            __shared__ double sumsave[QG][256];
            // GUARD (round 6, MAGIC sweeps): exp2_magic WITHOUT its clamp - 5 instructions per value - for the chunks whose exponents are proven
            // inside +-1022 before they are computed.  With R = sqrt(max -norm) over the chunk's rows (SweepArgs::tile_r) and, per query, NQ =
            // -norm and a = the accumulator's constant (norm - offset + bias), Cauchy-Schwarz gives |z_t.z_q| <= 2 R sqrt(NQ) in exponent units:
            //   WMUL (x = z_t.z_q + a):                      |x| <= 1022  <=  R <= min(1022 - a, 1022 + a) / (2 sqrt(NQ))
            //   norms in the accumulator (x = a + NQ - d2/2): x <= a + NQ <= 1022 and x >= a + NQ - (R + sqrt(NQ))^2 >= -1022
            //                                                              <=  R <= sqrt(1022 + a + NQ) - sqrt(NQ)
            // rlim = the smallest such bound over the wave's queries (recomputed when a checked redo moves the offsets); a chunk is taken
            // unclamped when every one of its tiles has tile_r <= rlim.  Inside the bound the clamped and the unclamped form are the same
            // function: which chunks pass changes the time, not a bit of the result.
            constexpr bool GUARD = MAGIC && PBN_MAGIC_GUARD;
            const PBN_GLOBAL double* __restrict__ TRp = (const PBN_GLOBAL double*)a.tile_r;
            double rlim = -1.0;
            auto set_rlim = [&]() {
                double lim = INFINITY;
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const double aq = (double)(cm[g] - Tr<T>::magic()), nq = -(double)ny[g];
                    const double sq = __builtin_sqrt(nq);
                    double l;
                    if (WMUL) {
                        const double h = aq < 0.0 ? 1022.0 + aq : 1022.0 - aq;
                        l = sq > 0.0 ? 0.5 * h / sq : (h >= 0.0 ? INFINITY : -1.0);
                    } else {
                        const double top = aq + nq;
                        l = top <= 1022.0 ? __builtin_sqrt(1022.0 + top) - sq : -1.0;
                    }
                    lim = l < lim ? l : lim;   // (a NaN bound - NaN queries - never lowers the limit: their sums are NaN whatever the path)
                    if (!(l == l)) lim = -1.0;
                }
#pragma unroll
                for (int o = 1; o < 64; o <<= 1) { const double v = __shfl_xor(lim, o); lim = v < lim ? v : lim; }
                rlim = lim;
            };
            if constexpr (GUARD) { if (TRp) set_rlim(); }
            for (int64_t tc = t0; tc < t1; tc += CH) {
                const int64_t te = tc + CH < t1 ? tc + CH : t1;
#pragma unroll
                for (int g = 0; g < QG; ++g) sumsave[g][threadIdx.x] = sum[g];
                bool open = false;
                unsigned omask = 0;   // bit i: tile tc + i is proven (a chunk with an outlier row keeps its other tiles bare)
                if constexpr (GUARD) {
                    if (TRp) {
                        const int64_t tt = tc + lane < te ? tc + lane : te - 1;
                        const unsigned long long ob = __ballot(lane < CH && TRp[tt] <= rlim);
                        omask = (unsigned)ob;
                        open = omask == 0xffffffffu;
                    }
                }
                // (measurement aid: tiles taken without the clamp / tiles, through pbn_debug_sweep_visits)
                if (GUARD && a.count_redo && lane == 0) { atomicAdd(&g_sweep_tiles, (unsigned long long)(te - tc)); atomicAdd(&g_sweep_visit, (unsigned long long)__builtin_popcount(omask & (te - tc >= 32 ? ~0u : ((1u << (te - tc)) - 1u)))); }
                load_tile(tc, afA, nxA, axA);
                if (open) {
                    for (int64_t t = tc; t < te; t += 2) {
                        const bool second = t + 1 < te;                   // wave-uniform
                        load_tile(second ? t + 1 : t, afB, nxB, axB);
                        process_fast(afA, nxA, 0, std::integral_constant<int, 0>{});
                        load_tile(t + 2 < te ? t + 2 : t, afA, nxA, axA);
                        if (second) process_fast(afB, nxB, 0, std::integral_constant<int, 0>{});
                    }
                } else {
                    for (int64_t t = tc; t < te; t += 2) {
                        const bool second = t + 1 < te;                   // wave-uniform
                        const unsigned ob = omask >> (unsigned)(t - tc);
                        load_tile(second ? t + 1 : t, afB, nxB, axB);
                        if (ob & 1u) process_fast(afA, nxA, 0, std::integral_constant<int, 0>{});
                        else process_fast(afA, nxA, 0, std::integral_constant<int, 1>{});
                        load_tile(t + 2 < te ? t + 2 : t, afA, nxA, axA);
                        if (second) {
                            if (ob & 2u) process_fast(afB, nxB, 0, std::integral_constant<int, 0>{});
                            else process_fast(afB, nxB, 0, std::integral_constant<int, 1>{});
                        }
                    }
                }
                bool bad = false;
#pragma unroll
                for (int g = 0; g < QG; ++g) bad = bad || !(sum[g] < 0x1p1000);   // not merely finite: the epilogue adds 4 lanes' sums
                const bool redo = __any(bad);
                if (a.count_redo && lane == 0) { atomicAdd(&g_sweep_units, 1ull); if (redo) atomicAdd(&g_sweep_redo, 1ull); }
                if (redo) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) sum[g] = sumsave[g][threadIdx.x];
                    checked_range(tc, te);
                    if constexpr (GUARD) { if (TRp) set_rlim(); }   // the offsets may have moved
                }
            }
        } else {
            checked_range(t0, t1);
        }
    }

    // ---- epilogue: combine the 4 row-lanes of each query column, write (m, sum) partials ---------
    PBN_GLOBAL double* part = (PBN_GLOBAL double*)a.part;
    constexpr int P = COND ? 4 : 2;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        double s = sum[g];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        double sj = 0.0;
        if (COND) {
            sj = sumj[g];
            sj += __shfl_xor(sj, 16);
            sj += __shfl_xor(sj, 32);
        }
        // an empty sum still holds the term its offset came from (see kde_sweep_f16_kernel; 2^bias is that term here)
        // (only when the offset is a real exponent: a NaN / infinite offset - NaN queries, an all-padding split - keeps its sum)
        const bool mfin = (m[g] - m[g]) == (T)0;
        // (not with a moment pass beside this sweep: the tile the offset came from may be the other pass's - an empty sum is empty)
        if (s == 0.0 && mfin && !lbm[g] && !MOM) s = __builtin_ldexp(1.0, (int)Tr<T>::bias());
        if (COND && sj == 0.0 && (mj[g] - mj[g]) == (T)0 && !lbmj[g]) sj = __builtin_ldexp(1.0, (int)Tr<T>::bias());
        if (lg == 0 && qt0 + g < a.nqtiles) {
            PBN_GLOBAL double* o = part + ((int64_t)split * a.nqtiles * 16 + (qt0 + g) * 16 + lane) * P;
            o[0] = (double)m[g] - (double)Tr<T>::bias();   // the sums carry 2^bias
            o[1] = s;
            if (COND) { o[2] = (double)mj[g] - (double)Tr<T>::bias(); o[3] = sj; }
        }
    }
}

template <typename T, int KS, bool COND, int QG, bool FOLD, bool PRUNE, bool WMUL = false, bool EF32 = false>
__global__ __launch_bounds__(sweep_block_threads(PRUNE), PRUNE ? PBN_F64_PRUNE_WAVES : 2) void kde_sweep_kernel(SweepArgs a) {
    kde_sweep_body<T, KS, COND, QG, FOLD, PRUNE, WMUL, EF32>(a, blockIdx.x);
}

// Grouped launch (kde_group.hip): the flat grid covers the sweeps of MANY units back to back, unit-major, every unit's share
// rounded up to 64 workgroups so that one table entry per 64 workgroups names the unit.  Pruned plain fp64 sweeps only.
// (four waves per SIMD here, three for the stand-alone launches: the grouped sweeps spend more of their time in the tile walk, whose latency a
//  fourth wave covers - cv64 2.25 -> 2.19 s, C3 with 24 iterations 26.66 -> 26.04 s; the stand-alone handles lose 2-5 % at four, tools/r5_probe_q.sh)
#ifndef PBN_F64_GROUP_WAVES
#define PBN_F64_GROUP_WAVES 4
#endif
template <typename T, int KS, int QG, bool FOLD, bool WMUL, bool MOM = false>
__global__ __launch_bounds__(sweep_block_threads(true), PBN_F64_GROUP_WAVES) void kde_sweep_group_kernel(GSweepArgs g) {
    const int u = g.wg_unit[blockIdx.x >> 6];
    const GSweepUnit& su = g.units[u];
    const unsigned bid = (unsigned)((int64_t)blockIdx.x - su.wg0);
    if (bid >= (unsigned)su.nwg) return;
    SweepArgs a;
    a.Apack = su.Apack; a.nxpack = su.nxpack; a.Axpack = nullptr;
    a.Bpack = su.Bpack; a.nypack = su.nypack; a.Bxpack = nullptr; a.Bxnorm = nullptr;
    a.ntiles = su.ntiles; a.nqtiles = su.nqtiles; a.tiles_per_split = su.tps;
    a.fold = g.fold; a.count_redo = g.count_redo; a.wmul = g.wmul; a.box_full = su.box_full; a.tile_r = nullptr;
    a.prune = 1; a.pdims = su.pdims; a.prune_margin = g.prune_margin > 0.0 ? g.prune_margin : (double)su.margin;
    a.tile_box = su.tile_box; a.qtile_box = su.qtile_box; a.qtile_thr = su.qtile_thr; a.qlb = su.qlb;
    a.nsplit_grid = su.nsplit; a.part = su.part; a.group_masks = g.group_masks;
    a.far_span = g.far_span;
    a.tile_rad2 = su.tile_rad2; a.tile_mom = su.tile_mom; a.batch_box = su.batch_box; a.batches_per_split = su.nbps;
    kde_sweep_body<T, KS, false, QG, FOLD, true, WMUL, /*EF32: the engine's terms are sums*/ true, MOM>(a, bid);
}

// The moment pass of a grouped fp64 sum-only sweep of D = 1 or 2 dimensions (round 5).  Same flat grid and the same (unit, query block,
// split) mapping as kde_sweep_group_kernel, a wave owns the same 16-query groups - but here LANE = TILE: per 64-tile batch every lane with a
// pair loads the record of its own tile (structure of arrays: 47 | 10 coalesced loads, once per batch and group that has a pair in it), and the
// 16 queries of the group are taken one after the other - their coordinates and offsets are uniform (v_readlane from the lanes that hold
// them), the tile's coefficients per-lane registers.  Per (tile, query): 2 D + ~19 instructions + 44 | 8 FMAs of the Horner scheme, i.e. ~65
// fp64 issue slots per pair at D = 2 when all 64 lanes hold a pair (measured ~85 cycles at 56-62 busy lanes) against ~180 for the sweep's MFMA +
// 2^f form - and a lane idles only where ITS tile is not this group's (the first forms of this kernel - lane = query with the records through
// scalar loads, then 4 tile slots x 16 queries with per-lane record loads - paid for the union of the wave's groups' tiles resp. for 23 vector
// loads per four tiles: no faster than the sweep).  The exponent of the common factor is split as in the sweep (biased, integer offset from the
// prepass bound) and 2^x takes the sweep's own form (2^f of the fraction on the fp32 unit: the budget's first entry covers it; with the fp64
// polynomial the pass differed from the sweep by the fp32 unit's MEAN error, 1.3e-9 per term).  Running sums per (query, lane) in LDS, one
// cross-lane reduction per group; partials go behind the sweep's own (GSweepUnit::part_mom).
__device__ __forceinline__ double readlane_f64(double v, int l) {   // lane l's value, uniform (two v_readlane_b32 into scalar registers)
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), l), hi = __builtin_amdgcn_readlane(__double2hiint(v), l);
    return __hiloint2double(hi, lo);
}
#ifndef PBN_MOM_WAVES2
#define PBN_MOM_WAVES2 2   // D = 2: 45 coefficients per lane - 3 waves / SIMD (168 VGPRs) spill them
#endif
#ifndef PBN_MOM_UNROLL
#define PBN_MOM_UNROLL 16
#endif
#ifndef PBN_MOM_EXP_F32
#define PBN_MOM_EXP_F32 1
#endif
template <int D>
__global__ __launch_bounds__(64, D == 2 ? PBN_MOM_WAVES2 : 3) void kde_moment_group_kernel(GSweepArgs g) {
    // per (group, query, lane) running sums: in LDS - QG x 16 x 64 doubles per wave (as registers they cost the D = 2 kernel 4 %: 256 VGPRs and
    // scratch, profiles/r6/moment_probes.txt)
    __shared__ double accs[PBN_QG_PRUNE * 16][64];
    const int u = g.wg_unit[blockIdx.x >> 6];
    const GSweepUnit& su = g.units[u];
    const unsigned bid = (unsigned)((int64_t)blockIdx.x - su.wg0);
    if (bid >= (unsigned)su.nwg) return;
    constexpr int QG = PBN_QG_PRUNE;
    constexpr int NC = pbn_mom_coefs(D);
    const int lane = threadIdx.x & 63;
    const unsigned Gq = (unsigned)((su.nqtiles + QG - 1) / QG), Gs = (unsigned)su.nsplit;
    const unsigned kk = bid / Gq;
    const int qx = (int)(bid % Gq);
    const int split = (kk & 1u) ? (int)(Gs - 1 - (kk >> 1)) : (int)(kk >> 1);   // pruned_block's order
    const int64_t qt0 = (int64_t)qx * QG;
    if (qt0 >= su.nqtiles) return;
    const int64_t t0 = (int64_t)split * su.tps;
    const int64_t t1 = (t0 + su.tps < su.ntiles) ? t0 + su.tps : su.ntiles;
    const int pd = su.pdims;
    const int64_t ms = su.mom_stride;
    const PBN_GLOBAL double* __restrict__ TBp = (const PBN_GLOBAL double*)su.tile_box;
    const PBN_GLOBAL double* __restrict__ QBp = (const PBN_GLOBAL double*)su.qtile_box;
    const PBN_GLOBAL double* __restrict__ QTp = (const PBN_GLOBAL double*)su.qtile_thr;
    const PBN_GLOBAL double* __restrict__ QLp = (const PBN_GLOBAL double*)su.qlb;
    const PBN_GLOBAL float* __restrict__ R2p = (const PBN_GLOBAL float*)su.tile_rad2;
    const PBN_GLOBAL double* __restrict__ MOp = (const PBN_GLOBAL double*)su.tile_mom;
    const PBN_GLOBAL double* __restrict__ ZQp = (const PBN_GLOBAL double*)su.zq;
    const double margin = g.prune_margin > 0.0 ? g.prune_margin : (double)su.margin;
    PBN_GLOBAL double* part = (PBN_GLOBAL double*)su.part_mom;

    // Round 6: the wave's QG query groups share ONE walk over the tiles - a batch's boxes are tested for every group, the records of the lanes
    // that have a pair with ANY of them are loaded once (47 | 10 coalesced loads) and each group's 16 queries run against them: half the record
    // loads and half the exposed load latency per pair at QG = 2 (the kernel spent 28 % of its wave time in s_waitcnt).  Every (query, lane) sum
    // still meets its batches in ascending order: the partials are the ones of the group-by-group form, bit for bit.
    // The groups' 16 queries live in lanes 0..15 (copies in the other lanes): coordinates and the exponent offset - the prepass's lower bound of
    // the query's largest exponent, an integer as in the sweep.  A padding row (bound -inf) gets an offset that kills its terms.
    // (the exponents carry the magic constant of the sweep beside this pass: the same 2^x, exp2_magic - x below is that accumulator form)
    constexpr bool MOMM = PBN_MOM_EXP_F32 && PBN_EXP2_MAGIC && PBN_MAGIC_PRUNED && PBN_EXP2_F32 && PBN_EXP2_DEGREE <= 7;
    constexpr double MOMC = MOMM ? PBN_MAGIC_C : 0.0;
    double mqv[QG], cmv[QG], zv[QG][D], thr[QG];
    bool gok[QG], chk[QG];
#pragma unroll
    for (int gi = 0; gi < QG; ++gi) {
        gok[gi] = qt0 + gi < su.nqtiles;
        const int64_t qg = gok[gi] ? qt0 + gi : qt0;
        const int64_t q = qg * 16 + (lane & 15);
        const double lb = __builtin_ceil(QLp[q]);
        const bool qok = (lb < 0.0 ? -lb : lb) < 0x1p50;
        mqv[gi] = qok ? lb : 0.0;
        cmv[gi] = qok ? (Tr<double>::bias() + MOMC) - lb : -0x1p60;
        // x = -d2 / 2 + cmv <= cmv: a group none of whose queries can reach 900 exponent units (the prepass bound of its largest exponent lies
        // within ~870 units of 0: every query with a training row within 41 bandwidths) runs its 16 queries without the overflow test - one
        // basic block of 16 independent Horner schemes instead of 16 blocks with a branch between them
        chk[gi] = __any(cmv[gi] > 900.0 + MOMC) != 0;
#pragma unroll
        for (int k = 0; k < D; ++k) zv[gi][k] = qok ? ZQp[q * D + k] : 0.0;
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) accs[gi * 16 + qi][lane] = 0.0;
        thr[gi] = QTp[qg];
    }
    unsigned long long taken = 0;
    for (int64_t sb = t0; sb < t1; sb += 4096) {
        // (the sweep's two levels: 64 batches classified at once, lane = batch, then the batches in reach)
        const int64_t bt = sb + 64 * lane;
        unsigned long long bm[QG], bmu = 0;
#pragma unroll
        for (int gi = 0; gi < QG; ++gi) {
            if (su.batch_box) {
                const PBN_GLOBAL double* bb = (const PBN_GLOBAL double*)su.batch_box + ((int64_t)split * su.nbps + ((bt - t0) >> 6)) * 2 * pd;
                bm[gi] = __ballot(gok[gi] && bt < t1 && batch_in_reach(bb, QBp + (qt0 + (gok[gi] ? gi : 0)) * 2 * pd, pd, thr[gi] - margin));
            } else {
                bm[gi] = __ballot(gok[gi] && bt < t1);
            }
            bmu |= bm[gi];
        }
        while (bmu) {
            const int bj = __builtin_ctzll(bmu);
            const int64_t tb = sb + 64 * (int64_t)bj;
            bmu &= bmu - 1;
            unsigned long long m[QG], mu = 0;
#pragma unroll
            for (int gi = 0; gi < QG; ++gi) {
                m[gi] = 0;
                if ((bm[gi] >> bj) & 1ull) {
                    unsigned long long nr;
                    if (g.count_redo && lane == 0) atomicAdd(&g_mom_visits, 1ull);
                    const unsigned long long kept = prune_group_mask3(TBp, QBp + (qt0 + gi) * 2 * pd, R2p, pd, tb, t1, thr[gi] - margin,
                                                                      g.far_span > 0.0 ? thr[gi] - (margin - g.far_span) : -INFINITY,
                                                                      thr[gi] - (margin + PBN_MOM_EXTRA), lane, nr, m[gi]);
                    if (g.count_redo && lane == 0 && (kept & ~m[gi])) atomicAdd(&g_mom_left, 1ull);
                    if (g.count_redo && lane == 0 && m[gi]) { atomicAdd(&g_mom_pairs, (unsigned long long)__builtin_popcountll(m[gi])); atomicAdd(&g_mom_batches, 1ull); }
                }
                mu |= m[gi];
                taken += (unsigned long long)__builtin_popcountll(m[gi]);
            }
            if (!mu) continue;
            // my tile's record; a lane without a pair keeps zero coefficients and a centroid 10^10 units away: its polynomial is 0 and its
            // exponent -5e19, whose 2^x is an exact 0 (v_fract_f64 of an integer, the saturated v_cvt_i32_f64, v_ldexp_f64): it adds nothing,
            // without a select per query
            double c[D], cf[NC];
#pragma unroll
            for (int k = 0; k < D; ++k) c[k] = 1e10;
#pragma unroll
            for (int k = 0; k < NC; ++k) cf[k] = 0.0;
            if ((mu >> lane) & 1ull) {
                const PBN_GLOBAL double* __restrict__ rec = MOp + (tb + lane);
#pragma unroll
                for (int k = 0; k < D; ++k) c[k] = rec[(int64_t)k * ms];
#pragma unroll
                for (int k = 0; k < NC; ++k) cf[k] = rec[(int64_t)(D + k) * ms];
            }
#pragma unroll
            for (int gi = 0; gi < QG; ++gi) {
                if (!m[gi]) continue;
                const bool act = (m[gi] >> lane) & 1ull;
                // MASK: some lane holds a record for ANOTHER group of the wave and no pair with this one - its exponent is forced to -5e19 as
                // well (a select per query; not needed while the groups' masks agree, the common case for neighbouring groups)
                auto run = [&](auto checked, auto masked) {
                    constexpr bool CHECK = decltype(checked)::value, MASK = decltype(masked)::value;
#pragma unroll PBN_MOM_UNROLL
                    for (int qi = 0; qi < 16; ++qi) {
                        const double ux = readlane_f64(zv[gi][0], qi) - c[0];
                        double d2 = ux * ux, uy = 0.0;
                        if constexpr (D == 2) { uy = readlane_f64(zv[gi][1], qi) - c[1]; d2 = __builtin_fma(uy, uy, d2); }
                        double x = __builtin_fma(-0.5, d2, readlane_f64(cmv[gi], qi));
                        if constexpr (MASK) x = act ? x : -5e19;
                        if constexpr (CHECK) {
                            while (__builtin_expect(__any(x > 900.0 + MOMC), 0)) {
                                // the offset is a LOWER bound of the query's largest exponent: a far-out query (heavy tails) can sit thousands of
                                // units below a row its short neighbour scan missed.  Rebase the query (uniform: every lane's sum for it, and the
                                // offset it lives with from here on) by a fixed integer number of units
                                accs[gi * 16 + qi][lane] *= 0x1p-512;
                                if ((lane & 15) == qi) { cmv[gi] -= 512.0; mqv[gi] += 512.0; }
                                x = act ? x - 512.0 : -5e19;
                            }
                        }
                        // 2^x as in the sweep this pass stands in for: 2^f of the fraction on the fp32 unit (<= 1.4e-7 of the pair's contribution,
                        // the budget's first entry); x >= 0 for every pair that matters (the biased offset), a negative x comes out <= 2x too large
                        const double e = MOMM ? exp2_magic<true>(x) : PBN_MOM_EXP_F32 ? exp2_f64_fract<true>(x, 0.0) : Tr<double>::ex2_hi(x);
                        double pv;
                        if constexpr (D == 1) {
                            pv = cf[0];
#pragma unroll
                            for (int i = 1; i <= PBN_MOM_ORDER; ++i) pv = __builtin_fma(pv, ux, cf[i]);
                        } else {
                            int k = 0;
                            pv = 0.0;
#pragma unroll
                            for (int j = PBN_MOM_ORDER; j >= 0; --j) {
                                double qj = cf[k++];
#pragma unroll
                                for (int i = PBN_MOM_ORDER - j - 1; i >= 0; --i) qj = __builtin_fma(qj, ux, cf[k++]);
                                pv = __builtin_fma(pv, uy, qj);
                            }
                        }
                        accs[gi * 16 + qi][lane] = __builtin_fma(e, pv, accs[gi * 16 + qi][lane]);
                    }
                };
                if (chk[gi]) run(std::true_type{}, std::true_type{});
                else if (m[gi] != mu) run(std::false_type{}, std::true_type{});
                else run(std::false_type{}, std::false_type{});
            }
        }
    }
    if (lane == 0 && taken) atomicAdd(&g_mom_taken[D - 1], taken);
    // the groups' sums: add the 64 lanes' (tiles') parts per query, lane qi writes query qi
#pragma unroll
    for (int gi = 0; gi < QG; ++gi) {
        if (!gok[gi]) continue;
        double mine = 0.0;
#pragma unroll
        for (int qi = 0; qi < 16; ++qi) {
            double v = accs[gi * 16 + qi][lane];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v += __shfl_xor(v, off);
            if (lane == qi) mine = v;
        }
        if (lane < 16) {
            const int64_t q = (qt0 + gi) * 16 + lane;
            PBN_GLOBAL double* o = part + ((int64_t)split * su.nqtiles * 16 + q) * 2;
            o[0] = mqv[gi] - Tr<double>::bias();   // the sums carry 2^bias, as the sweep's
            o[1] = mine;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// fp32 path on the 16-bit matrix cores ("f16x2", round 6; rounds 1-5: "bf16x3"): v_mfma_f32_16x16x4_f32 runs on the same FMA units as
// the VALU (measured: no overlap, tools/microbench.hip), v_mfma_f32_16x16x32_f16 does not (a 16-cycle MFMA costs the VALU ~8 issue
// cycles).  Round 6 measured the f16x2 sweep POWER-bound (profiles/r6/f32_power_bound.txt: 19 % fewer cycles bought a 19 % lower
// clock; 2.0 PFLOP/s of dense bf16 MFMA work): what a pair value costs in wall time is its matrix work, so the contraction is halved -
// every whitened coordinate is split into TWO f16 pieces z^ = a1 + a2 (a1 = f16(z), a2 = f16(z - a1): 22 mantissa bits, |z - z^| <=
// 2^-22 |z|), the three products with weight >= 2^-11 (a1 b1, a1 b2, a2 b1) are exact in the f32 accumulator, the fourth (a2 b2,
// <= 2^-22 |a||b|) is dropped, and the norms are taken from the REPRESENTED z^: what the sweep evaluates is -1/2 |z^_t - z^_q|^2 up to
// the dropped products - an input perturbation of 2^-22 relative (fp32 inputs carry 2^-24 themselves) plus <= 2^-22 sum |a2 b2|, where
// bf16x3 paid 2^-24 |z|^2 of cancellation error with its norms from the unsplit z.  K = 3 d + 3 slots instead of 6 d + 3: ONE 32-slot
// MFMA per tile pair up to 9 dimensions (two before), two up to 20.
// f16 has 5 exponent bits: pieces are kept out of its subnormal range and inside its finite range by power-of-two slot scales -
//   coordinate k, slots 3k ... 3k+2:   training (a1, a1 2^-6, a2 2^6)   x   query (b1, b2 2^6, b1 2^-6);
//   a scalar that rides in slots (the training norm -1/2|z^_t|^2, the CKDE / W32 query offsets) is cut by split3s into three pieces
//   x = 2^15 p1 + 2^5 p2 + 2^-6 p3 against the constants (2^15, 2^5, 2^-6) on the other side: |x| <= 2^31, residual <= 2^-33 |x|;
//   a piece that would be subnormal is stored as zero (its value stays in the residual the next piece takes), so the result does not
//   depend on whether the matrix cores flush f16 subnormals.
// Query coordinates beyond +-65504 (54 000 bandwidths from the centre of the training set) are clamped and counted (PackArgs::far_count).
// Slot s of a row: s = 3 k + r (dimension k, role r) for s < 3 dm, then the three norm pieces; slot s lives in MFMA s / 32, lane group
// (s % 32) / 8, element s % 8.   Fragment arrays: [tile][NB][64 lanes][8 f16].  The query side -1/2|z^_q|^2 - m_q is the MFMA's C operand
// (a persistent register quad), so the VALU does nothing but v_exp_f32 and the sums.
// CKDE: one extra MFMA whose slots are the extra coordinate (0-2), its training norm against the constants (3-5) and the constants against
// the query norm + (m_marg - m_joint) (8-10, rewritten by the lanes of group 1 when an offset is raised).
// ------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void pack_rows_f16_kernel(PackArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    const int64_t npad = a.ntiles * 16;
    if (r >= npad) return;
    const int64_t tile = r >> 4;
    const int idx = (int)(r & 15);
    const int d = a.d, dm = a.dm;
    const int NB = a.KS;  // number of 32-slot MFMAs of the main contraction
    const bool valid = r < a.n;

    double xc[PBN_MAX_D];
    if (valid) {
        const int64_t rr = a.perm ? (int64_t)a.perm[a.perm_stride > 1 ? r * a.perm_stride : r] : r;
        const int64_t lr = rr < a.n0 ? a.row0 + rr : a.row1 + (rr - a.n0);
        const int64_t src = a.rows ? (int64_t)a.rows[lr] : lr;
        for (int j = 0; j < d; ++j) {
            const float* col = (const float*)a.base + (int64_t)a.cols[j] * a.ld;
            xc[j] = (double)col[src] - a.mu[j];
        }
    }
    hpiece p1[PBN_MAX_D], p2[PBN_MAX_D];
    double nrm = 0.0;
    bool far = false;
    for (int i = 0; i < dm; ++i) {
        double z = 0.0;
        if (valid) {
            const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)i * d;
            for (int j = 0; j <= i; ++j) z = __builtin_fma(w[j], xc[j], z);
        }
        bool cl;
        const double zr = split2(z, p1[i], p2[i], cl);
        far = far || cl;
        nrm = __builtin_fma(zr, zr, nrm);
    }
    float nv = (float)(-0.5 * nrm);
    if (!valid) nv = a.is_query ? 0.0f : (float)PBN_PAD_NORM;
    const hpiece zero = (hpiece)0.0f, c1 = (hpiece)PBN_H_C1, c2 = (hpiece)PBN_H_C2, c3 = (hpiece)PBN_H_C3;
    f16x2_store_row((hf8*)a.pack, NB, tile, idx, dm, p1, p2, nv, a.is_query != 0);
    if (a.is_query) ((float*)a.npack)[tile * 16 + idx] = nv;
    if (a.xpack) {
        double z = 0.0;
        if (valid) {
            const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)dm * d;
            for (int j = 0; j <= dm; ++j) z = __builtin_fma(w[j], xc[j], z);
        }
        hpiece e1, e2, h1, h2, h3;
        bool cl;
        const double zr = split2(z, e1, e2, cl);
        far = far || cl;
        const float hn = (float)(-0.5 * zr * zr);
        split3s(hn, h1, h2, h3);
        const hpiece e1s = h_piece((float)e1 * (1.0f / PBN_H_LO));
        hf8* xp = (hf8*)a.xpack;
        hf8 g0, g1, gz;
#pragma unroll
        for (int j = 0; j < 8; ++j) gz[j] = zero;
        g0 = gz; g1 = gz;
        if (!a.is_query) {
            g0[0] = e1; g0[1] = e1s; g0[2] = e2; g0[3] = h1; g0[4] = h2; g0[5] = h3;
            g1[0] = c1; g1[1] = c2; g1[2] = c3;
        } else {
            g0[0] = e1; g0[1] = e2; g0[2] = e1s; g0[3] = c1; g0[4] = c2; g0[5] = c3;
            g1[0] = h1; g1[1] = h2; g1[2] = h3;
            ((float*)a.xnorm)[tile * 16 + idx] = hn;  // base of the rewritable slots 8..10
        }
        xp[tile * 64 + 0 * 16 + idx] = g0;
        xp[tile * 64 + 1 * 16 + idx] = g1;
        xp[tile * 64 + 2 * 16 + idx] = gz;
        xp[tile * 64 + 3 * 16 + idx] = gz;
    }
    if (a.is_query && a.far_flag) a.far_flag[r] = (far && valid) ? 1 : 0;
}

// Queries beyond the f16 range (PackArgs::far_flag; tens of thousands of bandwidths from the training set): one 256-thread block per query tile;
// a flagged query is evaluated in fp64 against every training row DECODED from the fragments (a1 + a2 - the values the sweeps use), its
// coordinates recomputed unclamped from the table, and its partials are replaced: split 0 gets (max exponent, sum), the others (the same
// offset, 0).  Nothing but a flag test when no query is flagged.
template <bool COND>
__global__ __launch_bounds__(256) void kde_far_fix_kernel(PackArgs a, const hf8* __restrict__ Apack, const hf8* __restrict__ Axpack, int NB, int64_t n_train,
                                                          int64_t ntiles, double* __restrict__ part, int nsplit, int64_t nqtiles) {
    constexpr int P = COND ? 4 : 2;
    __shared__ double zq[PBN_MAX_D + 1];
    __shared__ double red[4][4];
    const int64_t qtile = blockIdx.x;
    for (int qi = 0; qi < 16; ++qi) {
        const int64_t r = qtile * 16 + qi;
        if (r >= a.n || !a.far_flag[r]) continue;   // (uniform over the block)
        const int d = a.d, dm = a.dm;
        __syncthreads();
        if (threadIdx.x == 0) {
            const int64_t rr = a.perm ? (int64_t)a.perm[a.perm_stride > 1 ? r * a.perm_stride : r] : r;
            const int64_t lr = rr < a.n0 ? a.row0 + rr : a.row1 + (rr - a.n0);
            const int64_t src = a.rows ? (int64_t)a.rows[lr] : lr;
            double xc[PBN_MAX_D];
            for (int j = 0; j < d; ++j) xc[j] = (double)((const float*)a.base + (int64_t)a.cols[j] * a.ld)[src] - a.mu[j];
            for (int i = 0; i < d; ++i) {
                const double* w = (a.Wdev ? a.Wdev : a.W) + (size_t)i * d;
                double z = 0.0;
                for (int j = 0; j <= i; ++j) z = __builtin_fma(w[j], xc[j], z);
                zq[i] = z;
            }
        }
        __syncthreads();
        double m = -INFINITY, s = 0.0, mj = -INFINITY, sj = 0.0;
        const int spd = f16x2_spd(dm);
        for (int64_t t = threadIdx.x; t < n_train; t += 256) {
            const int64_t tile = t >> 4;
            const int idx = (int)(t & 15);
            double e = 0.0;
            for (int k = 0; k < dm; ++k) {
                const int s1 = spd * k, s2 = spd * k + 2;
                const double a1 = (double)(float)Apack[(tile * NB + (s1 >> 5)) * 64 + ((s1 & 31) >> 3) * 16 + idx][s1 & 7];
                const double a2 = (double)(float)Apack[(tile * NB + (s2 >> 5)) * 64 + ((s2 & 31) >> 3) * 16 + idx][s2 & 7];
                const double df = a1 + a2 * (1.0 / (double)PBN_H_LO) - zq[k];
                e = __builtin_fma(-0.5 * df, df, e);
            }
            if (e > m) { s = s * exp2(m - e) + 1.0; m = e; } else s += exp2(e - m);
            if (COND) {
                const double x1 = (double)(float)Axpack[tile * 64 + idx][0], x2 = (double)(float)Axpack[tile * 64 + idx][2];
                const double df = x1 + x2 * (1.0 / (double)PBN_H_LO) - zq[dm];
                const double ej = __builtin_fma(-0.5 * df, df, e);
                if (ej > mj) { sj = sj * exp2(mj - ej) + 1.0; mj = ej; } else sj += exp2(ej - mj);
            }
        }
        auto merge = [](double& m1, double& s1, double m2, double s2) {
            if (m2 > m1) { s1 = s1 * exp2(m1 - m2) + s2; m1 = m2; } else if (m2 > -INFINITY) s1 += s2 * exp2(m2 - m1);
        };
        for (int off = 32; off >= 1; off >>= 1) {
            merge(m, s, __shfl_xor(m, off), __shfl_xor(s, off));
            if (COND) merge(mj, sj, __shfl_xor(mj, off), __shfl_xor(sj, off));
        }
        if ((threadIdx.x & 63) == 0) { double* o = red[threadIdx.x >> 6]; o[0] = m; o[1] = s; o[2] = mj; o[3] = sj; }
        __syncthreads();
        if (threadIdx.x == 0) {
            for (int w = 1; w < 4; ++w) {
                merge(m, s, red[w][0], red[w][1]);
                if (COND) merge(mj, sj, red[w][2], red[w][3]);
            }
            for (int sp = 0; sp < nsplit; ++sp) {
                double* o = part + ((int64_t)sp * nqtiles * 16 + r) * P;
                o[0] = m; o[1] = sp == 0 ? s : 0.0;
                if (COND) { o[2] = mj; o[3] = sp == 0 ? sj : 0.0; }
            }
        }
    }
}

// waves per SIMD the pruned fp32 sweeps are compiled for: 4 (<= 128 VGPRs; the fused CKDE shape needs 180 unconstrained and
// spills a few prologue / rare-path values to scratch, none in the tile loop).  One-wave workgroups walking irregular tile
// lists are latency-bound: 2 -> 4 resident waves is worth 12 % of C5's hill-climb and 12-14 % on the fp32 handles;
// 5 (96 VGPRs) spills inside the loop.
#ifndef PBN_F16_PRUNE_WAVES
#define PBN_F16_PRUNE_WAVES 4
#endif
#ifndef PBN_F16_WAVES
#define PBN_F16_WAVES 2   // the same for the unpruned fp32 sweeps of up to 10 dimensions (4 waves per workgroup: workgroups per CU)
#endif
// (Measured again in round 3 and dropped again: the tile sums of 8 / 16 consecutive tiles added in fp32 before they join the fp64 sums -
//  one v_add_f32 instead of v_cvt_f64_f32 + v_add_f64 per (tile, group).  The allocator answers with +35 VGPRs (138 -> 173: two
//  waves per SIMD instead of three): fp32 headline 14.5 -> 17.7 ms, 16.2 ms when held to three waves; tools/f32_variants.sh.)
#ifndef PBN_F16_PAIRSUM
#define PBN_F16_PAIRSUM 1
#endif
#ifndef PBN_F16_BLIND
#define PBN_F16_BLIND 1   // plain fp32 sweeps: batches / chunks of tiles without the per-tile overflow test, checked once at their end
#endif
// Measured (tools/lib_variants.sh, profiles/r3/bf16_blind_probe.txt): pruned fp32 slice sweeps -5...6 % (C5 9.25 -> 9.04 s), unpruned sweeps
// with one MFMA per tile pair -3.6 %; with two (d = 8 headline) +3 %: an unpruned split starts from the offsets of its own first tile, a near
// row later in the split overflows against them (whitened squared distances differ by hundreds), and every such chunk is swept twice - chunks
// of 256 / 1024 tiles 16.3 / 23.9 ms against 13.7.  So: always for the pruned sweeps (offsets from the prepass bounds: nothing to redo), chunks
// of 64 tiles for the unpruned sweeps - the two-MFMA ones only since their offsets look at 16 tiles spread over the split (PBN_F16_PROBES).
#ifndef PBN_F16_PROBES
#define PBN_F16_PROBES 16   // with them the two-MFMA unpruned sweep gains from the blind chunks too: d = 8 headline 13.89 -> 13.56 ms (4 probes: 13.82)
#endif
#ifndef PBN_F16_FSUM
#define PBN_F16_FSUM 1
#endif
#ifndef PBN_F16_PRUNE_SCHED
#define PBN_F16_PRUNE_SCHED 1   // pruned blind form: the (tile, 4 groups) stream placed by sched_group_barrier (process_tile)
#endif
#ifndef PBN_F16_BLIND_CHUNK
#define PBN_F16_BLIND_CHUNK 64
#endif
#ifndef PBN_F16_BLIND_NB2
#define PBN_F16_BLIND_NB2 1   // (0 without the probe tiles of PBN_F16_PROBES: see above)
#endif
template <int NB, bool COND, int QG, bool PRUNE>
__device__ __forceinline__ void kde_sweep_f16_body(const SweepArgs& a, const unsigned bid) {
    using V = f4;
    constexpr int WPB = sweep_block_threads(PRUNE) / 64;   // pruned: one wave per workgroup (see kde_sweep_kernel)
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lg = lane >> 4;
    int qx, split;
    if (PRUNE) pruned_block(a, WPB * QG, bid, qx, split); else xcd_block(qx, split);
    const int64_t qt0 = ((int64_t)qx * WPB + wave) * QG;
    if (qt0 >= a.nqtiles) return;
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;

    const PBN_GLOBAL hf8* __restrict__ Ap = (const PBN_GLOBAL hf8*)a.Apack;
    const PBN_GLOBAL hf8* __restrict__ Xp = (const PBN_GLOBAL hf8*)a.Axpack;
    const PBN_GLOBAL hf8* __restrict__ Bp = (const PBN_GLOBAL hf8*)a.Bpack;
    const PBN_GLOBAL float* __restrict__ NYp = (const PBN_GLOBAL float*)a.nypack;
    const PBN_GLOBAL hf8* __restrict__ BXp = (const PBN_GLOBAL hf8*)a.Bxpack;
    const PBN_GLOBAL float* __restrict__ XNp = (const PBN_GLOBAL float*)a.Bxnorm;
    const PBN_GLOBAL double* __restrict__ TBp = (const PBN_GLOBAL double*)a.tile_box;
    const PBN_GLOBAL double* __restrict__ QBp = (const PBN_GLOBAL double*)a.qtile_box;
    const PBN_GLOBAL double* __restrict__ QTp = (const PBN_GLOBAL double*)a.qtile_thr;
    const PBN_GLOBAL double* __restrict__ QLp = (const PBN_GLOBAL double*)a.qlb;

    hf8 b[QG][NB];
    float ny[QG], m[QG];
    V cmv[QG];
    double sum[QG];
    hf8 bx[QG];
    float xn[QG], mj[QG];
    double sumj[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) b[g][mb] = Bp[(qt * NB + mb) * 64 + lane];
        ny[g] = NYp[qt * 16 + (lane & 15)];
        sum[g] = 0.0;
        if (COND) { bx[g] = BXp[qt * 64 + lane]; xn[g] = XNp[qt * 16 + (lane & 15)]; sumj[g] = 0.0; }
    }
    // tile pruning, as in kde_sweep_kernel
    double wlo[PBN_PRUNE_PD] = {}, whi[PBN_PRUNE_PD] = {}, wthr = 0;
    const int pd = PRUNE ? a.pdims : 0;
    if (PRUNE) {
        wthr = INFINITY;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k) { wlo[k] = INFINITY; whi[k] = -INFINITY; }
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
            const double th = QTp[qt];
            wthr = th < wthr ? th : wthr;
#pragma unroll
            for (int k = 0; k < PBN_PRUNE_PD; ++k)
                if (k < pd) {
                    const double l = QBp[qt * 2 * pd + k], h = QBp[qt * 2 * pd + pd + k];
                    wlo[k] = l < wlo[k] ? l : wlo[k];
                    whi[k] = h > whi[k] ? h : whi[k];
                }
        }
        wthr -= a.prune_margin;
    }
    auto set_bx = [&](int g) {  // slots 8..10 (lane group 1, elements 0..2) <- split3s(xn + m - mj)
        if (lg == 1) {
            hpiece q1, q2, q3;
            split3s(xn[g] + (m[g] - mj[g]), q1, q2, q3);
            bx[g][0] = q1; bx[g][1] = q2; bx[g][2] = q3;
        }
    };
    auto load_tile = [&](int64_t t, hf8 (&f)[NB], hf8& x) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) f[mb] = Ap[(t * NB + mb) * 64 + lane];
        if (COND) x = Xp[t * 64 + lane];
    };
    auto mfma_main = [&](const hf8 (&f)[NB], int g, V c) {
#pragma unroll
        for (int mb = 0; mb < NB; ++mb) c = __builtin_amdgcn_mfma_f32_16x16x32_f16(f[mb], b[g][mb], c, 0, 0, 0);
        return c;
    };

    // ---- prologue: offsets from the first tile ------------------------------------------------------------
    {
        hf8 f[NB], x;
        load_tile(t0, f, x);
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            const V c0 = {ny[g], ny[g], ny[g], ny[g]};
            V acc = mfma_main(f, g, c0);
            const float mx = colmax<float>(max4<float>(acc));
            m[g] = mx;
            const float cm = ny[g] - mx;
            cmv[g] = V{cm, cm, cm, cm};
            if (COND) {
                V accj = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bx[g], acc, 0, 0, 0);  // slots 8..10 hold xn (m = mj = 0)
                mj[g] = colmax<float>(max4<float>(accj));
                set_bx(g);
            }
        }
        // plain unpruned sweeps: the offsets also look at PBN_F16_PROBES - 1 more tiles spread over the split - a split whose first 16
        // rows all lie far from a query otherwise meets rows hundreds of exponent units above its offset, and every such tile takes the
        // rescue path (or, in a blind chunk, costs the chunk a second pass)
        if constexpr (!COND && !PRUNE && PBN_F16_PROBES > 1) {
#pragma unroll 1
            for (int pz = 1; pz < PBN_F16_PROBES; ++pz) {
                load_tile(t0 + (t1 - t0) * pz / PBN_F16_PROBES, f, x);
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    const V c0 = {ny[g], ny[g], ny[g], ny[g]};
                    const V acc = mfma_main(f, g, c0);
                    const float mx = colmax<float>(max4<float>(acc));
                    if (mx > m[g]) {
                        m[g] = mx;
                        const float cm = ny[g] - mx;
                        cmv[g] = V{cm, cm, cm, cm};
                    }
                }
            }
        }
    }

    // pruned sweeps: offsets from the prepass bounds where they lie above the first tile's maximum (see kde_sweep_kernel)
    bool lbm[QG], lbmj[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) lbm[g] = lbmj[g] = false;
    if constexpr (PRUNE) {
        if (a.qlb) {
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
                const float lb = (float)QLp[qt * 16 + (lane & 15)];
                // a bound so large that fp32 cannot hold it to a fraction of a unit (a query ~2000 bandwidths out) is not used:
                // the first tile's offset comes with a term that is known to survive the rounding, the bound does not
                const bool fin = __builtin_fabsf(lb) < 0x1p22f;
                lbm[g] = fin && lb > m[g];
                if (lbm[g]) {
                    m[g] = lb;
                    const float cm = ny[g] - lb;
                    cmv[g] = V{cm, cm, cm, cm};
                }
                if (COND) {
                    lbmj[g] = fin && lb > mj[g];
                    if (lbmj[g]) mj[g] = lb;
                    set_bx(g);
                }
            }
        }
    }

    // Plain unpruned sweeps (the else branch): all groups' MFMAs are issued before the first exponential so that the matrix pipe works under the
    // VALU's exponentials, one overflow test per tile, the rare path redoes a group (C2 fp32: 15.2 -> 14.1 ms).  The
    // fused CKDE sweep and the pruned sweeps keep the group-by-group form: with two accumulator sets per group in flight,
    // or one wave per SIMD less, the other form loses (C5's sweeps 33 -> 40 s; pruned d = 1 plain sweep 8.3 -> 10.2 ms).
    // plain unpruned sweeps (PBN_F16_PAIRSUM): the sums of the two tiles of a loop iteration are added in fp32 and join the fp64 sums
    // together - one v_cvt_f64_f32 + v_add_f64 per group and TWO tiles
    constexpr bool PAIRSUM = !COND && !PRUNE && PBN_F16_PAIRSUM;
    float pend[PAIRSUM ? QG : 1];
#pragma unroll
    for (int g = 0; g < (PAIRSUM ? QG : 1); ++g) pend[g] = 0.f;
    // PBN_F16_FSUM (round 4): inside a BLIND batch / chunk (at most 64 tiles, looked at once at its end) the tile sums are added in fp32 and
    // join the fp64 sums once per batch - the v_cvt_f64_f32 + v_add_f64 per (tile, group) were 8 of the ~60 issue slots of a tile's four
    // groups.  At most 64 fp32 additions of positive terms: <= 4e-6 relative on a sum, against the fp32 bar of 1e-3.
    constexpr bool FSUM = !COND && PBN_F16_BLIND && PBN_F16_FSUM;
    float fs[FSUM ? QG : 1];
#pragma unroll
    for (int g = 0; g < (FSUM ? QG : 1); ++g) fs[g] = 0.f;
    auto flush_fs = [&]() {
        if constexpr (FSUM) {
#pragma unroll
            for (int g = 0; g < QG; ++g) { sum[g] += (double)fs[g]; fs[g] = 0.f; }
        }
    };
    // `blind` (plain sweeps, PBN_F16_BLIND): no overflow test and no rescue path - the caller looks at the fp64 sums once per batch / chunk
    // of tiles and redoes it checked if one of them went bad (as the fp64 sweeps do).  An exponent overflows only 128 units above its
    // query's offset, and the offsets start from the prepass bounds (pruned) or from a tile of the split itself.
    auto process_tile = [&](const hf8 (&f)[NB], const hf8& x, const int bit = 0, const bool flush = true, auto blind = std::false_type{}) {
        constexpr bool BLIND = decltype(blind)::value;
        if constexpr (COND || PRUNE) {
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                V acc = mfma_main(f, g, cmv[g]);
                V accj;
                if (COND) accj = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bx[g], acc, 0, 0, 0);
                float e0 = Tr<float>::ex2(acc[0]), e1 = Tr<float>::ex2(acc[1]), e2 = Tr<float>::ex2(acc[2]), e3 = Tr<float>::ex2(acc[3]);
                float ts = (e0 + e1) + (e2 + e3);
                float tsj = 0;
                bool bad = BLIND ? false : !(ts < Tr<float>::big());
                if (COND) {
                    float j0 = Tr<float>::ex2(accj[0]), j1 = Tr<float>::ex2(accj[1]), j2 = Tr<float>::ex2(accj[2]), j3 = Tr<float>::ex2(accj[3]);
                    tsj = (j0 + j1) + (j2 + j3);
                    bad = bad || !(tsj < Tr<float>::big());
                }
                if (!BLIND && __builtin_expect(__any(bad), 0)) {
                    float mx = colmax<float>(max4<float>(acc));
                    if (mx > 0.f) {
                        m[g] += mx;
                        const float cm = ny[g] - m[g];
                        cmv[g] = V{cm, cm, cm, cm};
                        sum[g] *= exp2(-(double)mx);
                        acc -= mx;
                    }
                    e0 = Tr<float>::ex2(acc[0]); e1 = Tr<float>::ex2(acc[1]); e2 = Tr<float>::ex2(acc[2]); e3 = Tr<float>::ex2(acc[3]);
                    ts = (e0 + e1) + (e2 + e3);
                    if (COND) {
                        float mxj = colmax<float>(max4<float>(accj));
                        if (mxj > 0.f) {
                            mj[g] += mxj;
                            sumj[g] *= exp2(-(double)mxj);
                            accj -= mxj;
                        }
                        set_bx(g);
                        float j0 = Tr<float>::ex2(accj[0]), j1 = Tr<float>::ex2(accj[1]), j2 = Tr<float>::ex2(accj[2]), j3 = Tr<float>::ex2(accj[3]);
                        tsj = (j0 + j1) + (j2 + j3);
                    }
                }
                if constexpr (BLIND && FSUM) fs[g] += ts;
                else sum[g] += (double)ts;
                if (COND) sumj[g] += (double)tsj;
            }
#if PBN_F16_PRUNE_SCHED
            // Round 6: the blind pruned form's stream placed - M0 M1 [E0] M2 [E1] M3 [E2] [E3], E = the four exponentials and four additions of a
            // group: no exponential reads an accumulator younger than one group's work, the MFMAs issue between the VALU blocks instead of four
            // in a row followed by the hazard's s_nops
            if constexpr (BLIND && !COND && NB == 1 && QG == 4) {
                __builtin_amdgcn_sched_group_barrier(0x008, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 4, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
            }
#endif
        } else {
            V acc[QG], accj[QG];
            float ts[QG], tsj[QG];
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                acc[g] = mfma_main(f, g, cmv[g]);
                if (COND) accj[g] = __builtin_amdgcn_mfma_f32_16x16x32_f16(x, bx[g], acc[g], 0, 0, 0);
            }
            bool bad = false;
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                const float e0 = Tr<float>::ex2(acc[g][0]), e1 = Tr<float>::ex2(acc[g][1]), e2 = Tr<float>::ex2(acc[g][2]), e3 = Tr<float>::ex2(acc[g][3]);
                ts[g] = (e0 + e1) + (e2 + e3);
                tsj[g] = 0;
                if constexpr (!BLIND) bad = bad || !(ts[g] < Tr<float>::big());
                if (COND) {
                    const float j0 = Tr<float>::ex2(accj[g][0]), j1 = Tr<float>::ex2(accj[g][1]), j2 = Tr<float>::ex2(accj[g][2]), j3 = Tr<float>::ex2(accj[g][3]);
                    tsj[g] = (j0 + j1) + (j2 + j3);
                    bad = bad || !(tsj[g] < Tr<float>::big());
                }
            }
            if (!BLIND && __builtin_expect(__any(bad), 0)) {
#pragma unroll
                for (int g = 0; g < QG; ++g) {
                    bool badg = !(ts[g] < Tr<float>::big());
                    if (COND) badg = badg || !(tsj[g] < Tr<float>::big());
                    if (!__any(badg)) continue;
                    float mx = colmax<float>(max4<float>(acc[g]));
                    if (mx > 0.f) {
                        m[g] += mx;
                        const float cm = ny[g] - m[g];
                        cmv[g] = V{cm, cm, cm, cm};
                        if constexpr (PAIRSUM) { sum[g] += (double)pend[g]; pend[g] = 0.f; }
                        sum[g] *= exp2(-(double)mx);
                        acc[g] -= mx;
                    }
                    const float e0 = Tr<float>::ex2(acc[g][0]), e1 = Tr<float>::ex2(acc[g][1]), e2 = Tr<float>::ex2(acc[g][2]), e3 = Tr<float>::ex2(acc[g][3]);
                    ts[g] = (e0 + e1) + (e2 + e3);
                    if (COND) {
                        float mxj = colmax<float>(max4<float>(accj[g]));
                        if (mxj > 0.f) {
                            mj[g] += mxj;
                            sumj[g] *= exp2(-(double)mxj);
                            accj[g] -= mxj;
                        }
                        set_bx(g);
                        const float j0 = Tr<float>::ex2(accj[g][0]), j1 = Tr<float>::ex2(accj[g][1]), j2 = Tr<float>::ex2(accj[g][2]), j3 = Tr<float>::ex2(accj[g][3]);
                        tsj[g] = (j0 + j1) + (j2 + j3);
                    }
                }
            }
#pragma unroll
            for (int g = 0; g < QG; ++g) {
                if constexpr (BLIND && FSUM) {
                    fs[g] += ts[g];
                } else if constexpr (PAIRSUM) {
                    if (flush) { sum[g] += (double)(pend[g] + ts[g]); pend[g] = 0.f; } else pend[g] = ts[g];
                } else {
                    sum[g] += (double)ts[g];
                }
                if (COND) sumj[g] += (double)tsj[g];
            }
        }
    };

    hf8 fA[NB], fB[NB], xA, xB;
    // (Measured and dropped, profiles/r3/prune_stream_probe.txt: the visit masks of the whole split taken first - lane w keeping the mask of
    //  batch w - and the kept tiles then walked as ONE stream across the batches through a ring of 3 or 4 tile fragments, the next set bit
    //  coming from scalar code on a v_readlane'd word.  The fp32 slice sweeps ran 9-12 % SLOWER (1.92 against 1.71 ms at 720 000 x 80 000,
    //  d = 2), C5 9.67 against 9.36 s, with 3 waves per SIMD 10.7 s: the loop is not waiting for its tiles - four waves per SIMD cover the
    //  one tile of prefetch - and the ring's 8-12 registers push the 128-register kernel into scratch.)
    if constexpr (PRUNE) {
        if (a.count_redo && lane == 0) atomicAdd(&g_sweep_tiles, (unsigned long long)(t1 - t0));
        for (int64_t tb = t0; tb < t1; tb += 64) {   // see kde_sweep_kernel
            // (one mask per WAVE here: per-group masks as in the fp64 kernel - prune_group_mask - were measured and dropped for the
            //  fp32 kernels, which live on occupancy and straight-line issue: 1e6 x 1e5 handles +15...20 %, C5 15.8 -> 16.4 s)
            const unsigned long long mask = prune_visit_mask(TBp, pd, tb, t1, wlo, whi, wthr, lane);
            if (!mask) continue;
            if (a.count_redo && lane == 0) atomicAdd(&g_sweep_visit, (unsigned long long)__builtin_popcountll(mask));
            auto run_batch = [&](unsigned long long mk, auto blind) {
                // unconditional prefetch of the next kept tile (see kde_sweep_body: a conditional one costs a vmcnt(0) per tile)
                int b = __builtin_ctzll(mk);
                mk &= mk - 1;
                load_tile(tb + b, fA, xA);
                for (;;) {
                    const bool more = mk != 0;
                    const int b2 = more ? __builtin_ctzll(mk) : b;
                    mk &= mk - 1;
                    load_tile(tb + b2, fB, xB);
                    process_tile(fA, xA, b, true, blind);
                    if (!more) break;
                    const bool more2 = mk != 0;
                    const int b3 = more2 ? __builtin_ctzll(mk) : b2;
                    mk &= mk - 1;
                    load_tile(tb + b3, fA, xA);
                    process_tile(fB, xB, b2, true, blind);
                    if (!more2) break;
                    b = b3;
                }
            };
            if constexpr (!COND && PBN_F16_BLIND) {
                double saved[QG];
#pragma unroll
                for (int g = 0; g < QG; ++g) saved[g] = sum[g];
                run_batch(mask, std::true_type{});
                flush_fs();
                bool bad = false;
#pragma unroll
                for (int g = 0; g < QG; ++g) bad = bad || !(sum[g] < 0x1p1000);
                if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) sum[g] = saved[g];
                    run_batch(mask, std::false_type{});
                }
            } else {
                run_batch(mask, std::false_type{});
            }
        }
    } else {
        auto run_range = [&](int64_t c0, int64_t c1, auto blind) {
            load_tile(c0, fA, xA);
            for (int64_t t = c0; t < c1; t += 2) {
                const bool second = t + 1 < c1;
                load_tile(second ? t + 1 : t, fB, xB);
                process_tile(fA, xA, 0, !second, blind);          // PAIRSUM: the first tile's sums wait for the second one's
                load_tile(t + 2 < c1 ? t + 2 : t, fA, xA);
                if (second) process_tile(fB, xB, 0, true, blind);
            }
        };
        if constexpr (!COND && PBN_F16_BLIND && (NB == 1 || PBN_F16_BLIND_NB2)) {
            constexpr int64_t CHUNK = PBN_F16_BLIND_CHUNK;   // tiles (an even number: the pair sums are flushed at its end)
            for (int64_t c0 = t0; c0 < t1; c0 += CHUNK) {
                const int64_t c1 = c0 + CHUNK < t1 ? c0 + CHUNK : t1;
                double saved[QG];
#pragma unroll
                for (int g = 0; g < QG; ++g) saved[g] = sum[g];
                run_range(c0, c1, std::true_type{});
                flush_fs();
                bool bad = false;
#pragma unroll
                for (int g = 0; g < QG; ++g) bad = bad || !(sum[g] < 0x1p1000);
                if (__builtin_expect(__any(bad), 0)) {
#pragma unroll
                    for (int g = 0; g < QG; ++g) sum[g] = saved[g];
                    run_range(c0, c1, std::false_type{});
                }
            }
        } else {
            run_range(t0, t1, std::false_type{});
        }
    }

    PBN_GLOBAL double* part = (PBN_GLOBAL double*)a.part;
    constexpr int P = COND ? 4 : 2;
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        double s = sum[g];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        double sj = 0.0;
        if (COND) {
            sj = sumj[g];
            sj += __shfl_xor(sj, 16);
            sj += __shfl_xor(sj, 32);
        }
        // The offsets are exponents of pairs of this split's first tile.  When the exponents are so large that their fp32
        // rounding (ulp(|e|) >> 1: queries ~10^6 bandwidths away) makes the second evaluation of that tile underflow, the
        // sum can come out empty although it holds at least the offset's own term: count that term.  (A split whose tiles
        // were all pruned gets the same term: below 2^-64 of the query's sum by the pruning rule.)
        // (not when the offset is a prepass bound: no term of this split stands behind it, an empty sum is empty)
        if (s == 0.0 && (m[g] - m[g]) == 0.f && !lbm[g]) s = 1.0;
        if (COND && sj == 0.0 && (mj[g] - mj[g]) == 0.f && !lbmj[g]) sj = 1.0;
        if (lg == 0 && qt0 + g < a.nqtiles) {
            PBN_GLOBAL double* o = part + ((int64_t)split * a.nqtiles * 16 + (qt0 + g) * 16 + lane) * P;
            o[0] = (double)m[g];
            o[1] = s;
            if (COND) { o[2] = (double)mj[g]; o[3] = sj; }
        }
    }
}

template <int NB, bool COND, int QG, bool PRUNE>
__global__ __launch_bounds__(sweep_block_threads(PRUNE), PRUNE ? PBN_F16_PRUNE_WAVES : (NB <= 2 ? PBN_F16_WAVES : 2)) void kde_sweep_f16_kernel(SweepArgs a) {
    kde_sweep_f16_body<NB, COND, QG, PRUNE>(a, blockIdx.x);
}

// grouped launch of the pruned plain fp32 sweeps (see kde_sweep_group_kernel)
template <int NB>
__global__ __launch_bounds__(sweep_block_threads(true), PBN_F16_PRUNE_WAVES) void kde_sweep_f16_group_kernel(GSweepArgs g) {
    const int u = g.wg_unit[blockIdx.x >> 6];
    const GSweepUnit& su = g.units[u];
    const unsigned bid = (unsigned)((int64_t)blockIdx.x - su.wg0);
    if (bid >= (unsigned)su.nwg) return;
    SweepArgs a;
    a.Apack = su.Apack; a.nxpack = su.nxpack; a.Axpack = nullptr;
    a.Bpack = su.Bpack; a.nypack = su.nypack; a.Bxpack = nullptr; a.Bxnorm = nullptr;
    a.ntiles = su.ntiles; a.nqtiles = su.nqtiles; a.tiles_per_split = su.tps;
    a.fold = 0; a.count_redo = g.count_redo; a.wmul = 0;
    a.prune = 1; a.pdims = su.pdims; a.prune_margin = g.prune_margin > 0.0 ? g.prune_margin : (double)su.margin;
    a.tile_box = su.tile_box; a.qtile_box = su.qtile_box; a.qtile_thr = su.qtile_thr; a.qlb = su.qlb;
    a.nsplit_grid = su.nsplit; a.part = su.part; a.group_masks = 0;
    kde_sweep_f16_body<NB, false, PBN_F16_QG_PRUNE, true>(a, bid);
}

// ------------------------------------------------------------------------------------------------
// W32 form of the plain unpruned fp32 sweep (round 6): the SAME packed f16x2 fragments contracted by v_mfma_f32_32x32x16_f16 -
// 32 training rows x 32 queries per accumulator, 16 pair values per lane and MFMA chain instead of 4.  An MFMA holds the SIMD's vector
// issue for 8 cycles whatever its shape (MI355X_MICROARCH.md, "vector-instruction ISSUE cost"), so the matrix side of a pair value
// costs 2 issue cycles at two 32-slot blocks (d = 5...9) instead of 4: the instruction-count bound of the d = 8 headline falls from
// 64 to 56 issue cycles per 256 pair values (v_exp_f32 8 + v_add_f32 4 per value, + the MFMAs).  Round 4 measured this shape
// compiler-scheduled and dropped it (14.1 against 13.05 ms: the four dependent 32-cycle MFMAs of a super-group are a longer chain than
// the wave's own exponentials cover).  Here the stream is PLACED: the loop is software-pipelined by one super-group - the chain of
// (tile pair, super-group s) issues while the 16 exponentials and 16 additions of the previous chain's accumulator run -, and
// __builtin_amdgcn_sched_group_barrier pins the order [MFMA, 4 x v_exp_f32, 4 x v_add_f32] x 4 per phase, so that no MFMA waits for
// its predecessor and no v_exp_f32 reads an accumulator younger than one phase.
//   * lane l: query column l % 32 of the super-group, half h = l / 32.  MFMA j of a chain takes slots 16 j + 8 h ... + 7 = block j / 2,
//     lane group 2 (j % 2) + h of the 16x16x32 fragment layout: the fragment arrays are read through another index map, nothing is repacked.
//     A training tile PAIR (rows 0-15 from one 16-row tile, 16-31 from another) feeds the A operand: lanes with (l % 32) < 16 read the first.
//   * the query side -1/2|z_q|^2 - m_q cannot be the C operand (16 registers per super-group): it rides in the three LAST slots of the
//     contraction (32 NB - 3 ...: split3 on the query side, rewritten when an offset moves; ones on the training side, written by
//     pack_rows_f16_kernel when 6 dm + 6 <= 32 NB - the 16x16 kernels meet zeros on the query side there), and C is the inline constant 0.
//   * accumulator row of register r: 8 (r / 4) + 4 h + r % 4 - registers 8...15 are the second tile of the pair (dropped for an odd tail).
//   * blind chunks of 64 tiles with fp32 tile sums, offsets from 16 probe tile pairs, checked redo: as kde_sweep_f16_body.
// Replaces kde/opencl_kernels/KDE.cl.src:115-121,143-170 for fp32 tables of 5...9 whitened dimensions.
// ------------------------------------------------------------------------------------------------
typedef float f16v __attribute__((ext_vector_type(16)));
#ifndef PBN_F16_W32_WAVES
#define PBN_F16_W32_WAVES 2
#endif
#ifndef PBN_F16_W32_SCHED
#define PBN_F16_W32_SCHED 1
#endif

template <int NB>
__global__ __launch_bounds__(256, PBN_F16_W32_WAVES) void kde_sweep_f16_w32_kernel(SweepArgs a) {
    constexpr int NJ = 2 * NB;   // MFMAs per chain
    constexpr int S = 2;         // super-groups of 32 queries per wave (= the 4 x 16 queries of the 16x16 kernel's wave: same grid)
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int half = lane >> 5, col = lane & 31, sub = col >> 4, idx = col & 15;
    int qx, split;
    xcd_block(qx, split);
    const int64_t qt0 = ((int64_t)qx * 4 + wave) * (2 * S);
    if (qt0 >= a.nqtiles) return;
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;

    const PBN_GLOBAL hf8* __restrict__ Ap = (const PBN_GLOBAL hf8*)a.Apack;
    const PBN_GLOBAL hf8* __restrict__ Bp = (const PBN_GLOBAL hf8*)a.Bpack;
    const PBN_GLOBAL float* __restrict__ NYp = (const PBN_GLOBAL float*)a.nypack;
    const int loff = half * 16 + idx;   // lane's place inside a (tile, block, j % 2) group of 32 fragment lanes

    hf8 b[S][NJ];
    float ny[S], m[S];
    double sum[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int64_t qt = qt0 + 2 * s + sub;
        qt = qt < a.nqtiles ? qt : a.nqtiles - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[s][j] = Bp[(qt * NB + (j >> 1)) * 64 + (j & 1) * 32 + loff];
        ny[s] = NYp[qt * 16 + idx];
        m[s] = 0.f;
        sum[s] = 0.0;
    }
    auto set_off = [&](int s) {   // slots 32 NB - 3 ... of the query side <- split3(-1/2|z_q|^2 - m_q)
        hpiece q1, q2, q3;
        split3s(ny[s] - m[s], q1, q2, q3);
        if (half == 1) { b[s][NJ - 1][5] = q1; b[s][NJ - 1][6] = q2; b[s][NJ - 1][7] = q3; }
    };
    auto load_pair = [&](int64_t ta, int64_t tb, hf8 (&f)[NJ]) {
        const int64_t t = sub ? tb : ta;
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[j] = Ap[(t * NB + (j >> 1)) * 64 + (j & 1) * 32 + loff];
    };
    auto chain = [&](const hf8 (&f)[NJ], int s) {
        f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[j], b[s][j], c, 0, 0, 0);
        return c;
    };
    auto colmax32 = [&](const f16v& v, int nr) {   // largest of the lane's first nr registers, then over the two halves of the column
        float mx = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r)
            if (r < nr) mx = v[r] > mx ? v[r] : mx;
        const float o = __shfl_xor(mx, 32);
        return mx > o ? mx : o;
    };

    // ---- offsets: the largest exponent of PBN_F16_PROBES tile pairs spread over the split (see kde_sweep_f16_body) ----
    {
#pragma unroll
        for (int s = 0; s < S; ++s) set_off(s);   // m = 0
        float mm[S];
#pragma unroll 1
        for (int pz = 0; pz < PBN_F16_PROBES; ++pz) {
            const int64_t ta = t0 + (t1 - t0) * pz / PBN_F16_PROBES;
            const int64_t tb = ta + 1 < t1 ? ta + 1 : ta;
            hf8 f[NJ];
            load_pair(ta, tb, f);
#pragma unroll
            for (int s = 0; s < S; ++s) {
                const float mx = colmax32(chain(f, s), 16);
                mm[s] = (pz == 0 || mx > mm[s]) ? mx : mm[s];
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) { m[s] = mm[s]; set_off(s); }
    }

    // ---- checked form (the redo of a chunk whose sums overflowed, and the odd tail): one tile pair, super-group by super-group ----
    auto checked_pair = [&](int64_t ta, int64_t tb, const bool second) {
        hf8 f[NJ];
        load_pair(ta, tb, f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f16v acc = chain(f, s);
            const int nr = second ? 16 : 8;
            auto tile_sum = [&]() {
                float ts = 0.f;
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (r < nr) ts += Tr<float>::ex2(acc[r]);
                return ts;
            };
            float ts = tile_sum();
            if (__builtin_expect(__any(!(ts < Tr<float>::big())), 0)) {
                const float mx = colmax32(acc, nr);
                if (mx > 0.f) {
                    m[s] += mx;
                    set_off(s);
                    sum[s] *= exp2(-(double)mx);
                    acc -= mx;
                }
                ts = tile_sum();
            }
            sum[s] += (double)ts;
        }
    };

    // ---- blind run of nbody x 4 tiles from c0: software-pipelined by one super-group, the stream placed by hand ----
    float fs[S];
    auto expsum = [&](const f16v& v) {   // 16 v_exp_f32 + 15 v_add_f32: four quads (e0 + e1) + (e2 + e3), added in order
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e0 = Tr<float>::ex2(v[4 * k]), e1 = Tr<float>::ex2(v[4 * k + 1]), e2 = Tr<float>::ex2(v[4 * k + 2]), e3 = Tr<float>::ex2(v[4 * k + 3]);
            q[k] = (e0 + e1) + (e2 + e3);
        }
        return ((q[0] + q[1]) + q[2]) + q[3];
    };
    auto place = [&]() {   // [MFMA, 4 trans, 4 VALU] x NJ
#if PBN_F16_W32_SCHED
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 16 / NJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 16 / NJ, 0);
        }
#endif
    };
    // contiguous tile pairs: the lane's byte offset inside a pair is fixed, the pair's base is wave-uniform (scalar address arithmetic)
    const uint32_t lane_b = (uint32_t)((sub * NB * 64 + loff) * 16);
    auto load_run = [&](int64_t t, hf8 (&f)[NJ]) {
        const PBN_GLOBAL char* base = (const PBN_GLOBAL char*)Ap + t * (int64_t)(NB * 64 * 16);
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[j] = *(const PBN_GLOBAL hf8*)(base + lane_b + (uint32_t)(((j >> 1) * 64 + (j & 1) * 32) * 16));
    };
    auto blind_run = [&](int64_t c0, int nbody) {
        hf8 fA[NJ], fB[NJ];
        f16v acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = -1000.f;   // the pipeline's first exponentials: 2^-1000 = 0
        load_run(c0, fA);
        int64_t t = c0;
#pragma unroll 1
        for (int i = 0; i < nbody; ++i, t += 4) {
            load_run(t + 2, fB);
            acc0 = chain(fA, 0);
            fs[1] += expsum(acc1);
            place();
            acc1 = chain(fA, 1);
            fs[0] += expsum(acc0);
            place();
            load_run(i + 1 < nbody ? t + 4 : t, fA);   // the last body's prefetch stays inside the run
            acc0 = chain(fB, 0);
            fs[1] += expsum(acc1);
            place();
            acc1 = chain(fB, 1);
            fs[0] += expsum(acc0);
            place();
        }
        fs[1] += expsum(acc1);
    };

    for (int64_t c0 = t0; c0 < t1; c0 += PBN_F16_BLIND_CHUNK) {
        const int64_t c1 = c0 + PBN_F16_BLIND_CHUNK < t1 ? c0 + PBN_F16_BLIND_CHUNK : t1;
        const int nbody = (int)((c1 - c0) >> 2);
        const int64_t cb = c0 + 4 * (int64_t)nbody;   // [cb, c1): at most three tiles, checked
        if (nbody) {
#pragma unroll
            for (int s = 0; s < S; ++s) fs[s] = 0.f;
            blind_run(c0, nbody);
            bool bad = false;
#pragma unroll
            for (int s = 0; s < S; ++s) bad = bad || !(fs[s] < Tr<float>::big());
            if (__builtin_expect(__any(bad), 0)) {
#pragma unroll 1
                for (int64_t t = c0; t < cb; t += 2) checked_pair(t, t + 1, true);
            } else {
#pragma unroll
                for (int s = 0; s < S; ++s) sum[s] += (double)fs[s];
            }
        }
#pragma unroll 1
        for (int64_t t = cb; t < c1; t += 2) checked_pair(t, t + 1 < c1 ? t + 1 : t, t + 1 < c1);
    }

    PBN_GLOBAL double* part = (PBN_GLOBAL double*)a.part;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double v = sum[s];
        v += __shfl_xor(v, 32);
        if (v == 0.0 && (m[s] - m[s]) == 0.f) v = 1.0;   // an empty sum holds at least the offset's own term (see kde_sweep_f16_body)
        const int64_t qt = qt0 + 2 * s + sub;
        if (half == 0 && qt < a.nqtiles) {
            PBN_GLOBAL double* o = part + ((int64_t)split * a.nqtiles * 16 + qt * 16 + idx) * 2;
            o[0] = (double)m[s];
            o[1] = v;
        }
    }
}

// ------------------------------------------------------------------------------------------------
// W32 form of the PRUNED plain fp32 sweeps (round 6; stand-alone handles and the grouped launches of the score engine: C5).  The kept tiles of
// a split - whatever batch they come from - are taken two at a time as the A operand of v_mfma_f32_32x32x16_f16 (any two 16-row tiles make a
// 32-row operand: lanes with (lane % 32) < 16 read the first), the wave's four 16-query groups are its two 32-query super-groups: per 2 048
// pair values 4 MFMAs instead of 8 (4 issue cycles per 256 values instead of 8), the stream software-pipelined by one super-group and placed as
// in kde_sweep_f16_w32_kernel.  The whole split is ONE blind region (fp32 tile sums: at most a few hundred pair sums per split, <= 3e-5
// relative, against the fp32 bar of 1e-3; offsets from the prepass bounds: an overflow is rare) - a wave whose sums came out bad walks its kept
// tiles again through the checked form.  An odd kept tile goes through the checked form too.  Visit masks: one per wave and 64-tile batch, as
// in kde_sweep_f16_body.
// ------------------------------------------------------------------------------------------------
#ifndef PBN_F16_W32P
#define PBN_F16_W32P 1
#endif
template <int NB>
__device__ __forceinline__ void kde_sweep_f16_w32p_body(const SweepArgs& a, const unsigned bid) {
    constexpr int NJ = 2 * NB, S = 2, QG = PBN_F16_QG_PRUNE;
    static_assert(QG == 2 * S, "the wave's query groups are its two 32-query super-groups");
    const int lane = threadIdx.x & 63;
    const int half = lane >> 5, col = lane & 31, sub = col >> 4, idx = col & 15;
    int qx, split;
    pruned_block(a, QG, bid, qx, split);
    const int64_t qt0 = (int64_t)qx * QG;
    if (qt0 >= a.nqtiles) return;
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;
    const PBN_GLOBAL hf8* __restrict__ Ap = (const PBN_GLOBAL hf8*)a.Apack;
    const PBN_GLOBAL hf8* __restrict__ Bp = (const PBN_GLOBAL hf8*)a.Bpack;
    const PBN_GLOBAL float* __restrict__ NYp = (const PBN_GLOBAL float*)a.nypack;
    const PBN_GLOBAL double* __restrict__ TBp = (const PBN_GLOBAL double*)a.tile_box;
    const PBN_GLOBAL double* __restrict__ QBp = (const PBN_GLOBAL double*)a.qtile_box;
    const PBN_GLOBAL double* __restrict__ QTp = (const PBN_GLOBAL double*)a.qtile_thr;
    const PBN_GLOBAL double* __restrict__ QLp = (const PBN_GLOBAL double*)a.qlb;
    const int loff = half * 16 + idx;

    hf8 b[S][NJ];
    float ny[S], m[S];
    double sum[S];
    bool lbm[S];
#pragma unroll
    for (int s = 0; s < S; ++s) {
        int64_t qt = qt0 + 2 * s + sub;
        qt = qt < a.nqtiles ? qt : a.nqtiles - 1;
#pragma unroll
        for (int j = 0; j < NJ; ++j) b[s][j] = Bp[(qt * NB + (j >> 1)) * 64 + (j & 1) * 32 + loff];
        ny[s] = NYp[qt * 16 + idx];
        m[s] = 0.f;
        sum[s] = 0.0;
        lbm[s] = false;
    }
    // the wave's query box and threshold (as kde_sweep_f16_body)
    double wlo[PBN_PRUNE_PD], whi[PBN_PRUNE_PD], wthr = INFINITY;
    const int pd = a.pdims;
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k) { wlo[k] = INFINITY; whi[k] = -INFINITY; }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        const int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
        const double th = QTp[qt];
        wthr = th < wthr ? th : wthr;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) {
                const double l = QBp[qt * 2 * pd + k], h = QBp[qt * 2 * pd + pd + k];
                wlo[k] = l < wlo[k] ? l : wlo[k];
                whi[k] = h > whi[k] ? h : whi[k];
            }
    }
    wthr -= a.prune_margin;

    auto set_off = [&](int s) {   // slots 32 NB - 3 ... of the query side <- split3s(-1/2|z_q|^2 - m_q)
        hpiece q1, q2, q3;
        split3s(ny[s] - m[s], q1, q2, q3);
        if (half == 1) { b[s][NJ - 1][5] = q1; b[s][NJ - 1][6] = q2; b[s][NJ - 1][7] = q3; }
    };
    auto load_pair = [&](int64_t ta, int64_t tb, hf8 (&f)[NJ]) {
        const int64_t t = sub ? tb : ta;
#pragma unroll
        for (int j = 0; j < NJ; ++j) f[j] = Ap[(t * NB + (j >> 1)) * 64 + (j & 1) * 32 + loff];
    };
    auto chain = [&](const hf8 (&f)[NJ], int s) {
        f16v c = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int j = 0; j < NJ; ++j) c = __builtin_amdgcn_mfma_f32_32x32x16_f16(f[j], b[s][j], c, 0, 0, 0);
        return c;
    };
    auto colmax32 = [&](const f16v& v, int nr) {
        float mx = v[0];
#pragma unroll
        for (int r = 1; r < 16; ++r)
            if (r < nr) mx = v[r] > mx ? v[r] : mx;
        const float o = __shfl_xor(mx, 32);
        return mx > o ? mx : o;
    };
    // ---- offsets: the first tile pair of the split, then the prepass bounds where they lie above (see kde_sweep_f16_body) ----
    {
#pragma unroll
        for (int s = 0; s < S; ++s) set_off(s);   // m = 0
        hf8 f[NJ];
        load_pair(t0, t0 + 1 < t1 ? t0 + 1 : t0, f);
#pragma unroll
        for (int s = 0; s < S; ++s) m[s] = colmax32(chain(f, s), 16);
        if (a.qlb) {
#pragma unroll
            for (int s = 0; s < S; ++s) {
                int64_t qt = qt0 + 2 * s + sub;
                qt = qt < a.nqtiles ? qt : a.nqtiles - 1;
                const float lb = (float)QLp[qt * 16 + idx];
                const bool fin = __builtin_fabsf(lb) < 0x1p22f;   // (a bound fp32 cannot hold to a fraction of a unit is not used)
                lbm[s] = fin && lb > m[s];
                if (lbm[s]) m[s] = lb;
            }
        }
#pragma unroll
        for (int s = 0; s < S; ++s) set_off(s);
    }
    if (a.count_redo && lane == 0) atomicAdd(&g_sweep_tiles, (unsigned long long)(t1 - t0));

    // ---- the kept tiles of the split, in order: one visit mask per 64-tile batch (uniform control flow: every lane tests its own tile) ----
    int64_t wtb = t0 - 64;
    unsigned long long wmask = 0;
    auto rewind = [&]() { wtb = t0 - 64; wmask = 0; };
    auto next_tile = [&]() -> int64_t {
        while (!wmask) {
            wtb += 64;
            if (wtb >= t1) return -1;
            wmask = prune_visit_mask(TBp, pd, wtb, t1, wlo, whi, wthr, lane);
            if (a.count_redo && lane == 0 && wmask) atomicAdd(&g_sweep_visit, (unsigned long long)__builtin_popcountll(wmask));
        }
        const int bit = __builtin_ctzll(wmask);
        wmask &= wmask - 1;
        return wtb + bit;
    };

    // ---- checked form: ONE tile (rows 0-15 of the pair's accumulator), super-group by super-group ----
    auto checked_single = [&](int64_t t) {
        hf8 f[NJ];
        load_pair(t, t, f);
#pragma unroll
        for (int s = 0; s < S; ++s) {
            f16v acc = chain(f, s);
            auto tile_sum = [&]() {
                float ts = 0.f;
#pragma unroll
                for (int r = 0; r < 8; ++r) ts += Tr<float>::ex2(acc[r]);
                return ts;
            };
            float ts = tile_sum();
            if (__builtin_expect(__any(!(ts < Tr<float>::big())), 0)) {
                const float mx = colmax32(acc, 8);
                if (mx > 0.f) {
                    m[s] += mx;
                    set_off(s);
                    sum[s] *= exp2(-(double)mx);
                    acc -= mx;
                }
                ts = tile_sum();
            }
            sum[s] += (double)ts;
        }
    };

    // ---- blind walk: pairs of kept tiles, software-pipelined by one super-group ----
    float fs[S];
#pragma unroll
    for (int s = 0; s < S; ++s) fs[s] = 0.f;
    auto expsum = [&](const f16v& v) {
        float q[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const float e0 = Tr<float>::ex2(v[4 * k]), e1 = Tr<float>::ex2(v[4 * k + 1]), e2 = Tr<float>::ex2(v[4 * k + 2]), e3 = Tr<float>::ex2(v[4 * k + 3]);
            q[k] = (e0 + e1) + (e2 + e3);
        }
        return ((q[0] + q[1]) + q[2]) + q[3];
    };
    auto place = [&]() {
#pragma unroll
        for (int j = 0; j < NJ; ++j) {
            __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
            __builtin_amdgcn_sched_group_barrier(0x400, 16 / NJ, 0);
            __builtin_amdgcn_sched_group_barrier(0x002, 16 / NJ, 0);
        }
    };
    int64_t single = -1;
    {
        hf8 fA[NJ], fB[NJ];
        f16v acc0, acc1;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc1[r] = -1000.f;
        int64_t pa = next_tile(), pb = pa >= 0 ? next_tile() : -1;
        bool have = pb >= 0;
        if (!have) single = pa;
        if (have) load_pair(pa, pb, fA);
        while (have) {
            // the next pair (unconditional prefetch: a conditional load costs a vmcnt(0) per pair - see kde_sweep_f16_body)
            int64_t na = next_tile(), nb = na >= 0 ? next_tile() : -1;
            const bool more = nb >= 0;
            if (!more) single = na;
            load_pair(more ? na : pa, more ? nb : pb, fB);
            acc0 = chain(fA, 0);
            fs[1] += expsum(acc1);
            place();
            acc1 = chain(fA, 1);
            fs[0] += expsum(acc0);
            place();
            if (!more) break;
            pa = next_tile();
            pb = pa >= 0 ? next_tile() : -1;
            have = pb >= 0;
            if (!have) single = pa;
            load_pair(have ? pa : na, have ? pb : nb, fA);
            acc0 = chain(fB, 0);
            fs[1] += expsum(acc1);
            place();
            acc1 = chain(fB, 1);
            fs[0] += expsum(acc0);
            place();
        }
        fs[1] += expsum(acc1);
    }
    bool bad = false;
#pragma unroll
    for (int s = 0; s < S; ++s) bad = bad || !(fs[s] < Tr<float>::big());
    if (__builtin_expect(__any(bad), 0)) {   // the split again, tile by tile, checked (the odd tile included)
        rewind();
#pragma unroll 1
        for (int64_t t = next_tile(); t >= 0; t = next_tile()) checked_single(t);
    } else {
#pragma unroll
        for (int s = 0; s < S; ++s) sum[s] += (double)fs[s];
        if (single >= 0) checked_single(single);
    }

    PBN_GLOBAL double* part = (PBN_GLOBAL double*)a.part;
#pragma unroll
    for (int s = 0; s < S; ++s) {
        double v = sum[s];
        v += __shfl_xor(v, 32);
        if (v == 0.0 && (m[s] - m[s]) == 0.f && !lbm[s]) v = 1.0;   // (see kde_sweep_f16_body: not when the offset is a prepass bound)
        const int64_t qt = qt0 + 2 * s + sub;
        if (half == 0 && qt < a.nqtiles) {
            PBN_GLOBAL double* o = part + ((int64_t)split * a.nqtiles * 16 + qt * 16 + idx) * 2;
            o[0] = (double)m[s];
            o[1] = v;
        }
    }
}

template <int NB>
__global__ __launch_bounds__(sweep_block_threads(true), PBN_F16_PRUNE_WAVES) void kde_sweep_f16_w32p_kernel(SweepArgs a) {
    kde_sweep_f16_w32p_body<NB>(a, blockIdx.x);
}
template <int NB>
__global__ __launch_bounds__(sweep_block_threads(true), PBN_F16_PRUNE_WAVES) void kde_sweep_f16_w32p_group_kernel(GSweepArgs g) {
    const int u = g.wg_unit[blockIdx.x >> 6];
    const GSweepUnit& su = g.units[u];
    const unsigned bid = (unsigned)((int64_t)blockIdx.x - su.wg0);
    if (bid >= (unsigned)su.nwg) return;
    SweepArgs a;
    a.Apack = su.Apack; a.nxpack = su.nxpack; a.Axpack = nullptr;
    a.Bpack = su.Bpack; a.nypack = su.nypack; a.Bxpack = nullptr; a.Bxnorm = nullptr;
    a.ntiles = su.ntiles; a.nqtiles = su.nqtiles; a.tiles_per_split = su.tps;
    a.fold = 0; a.count_redo = g.count_redo; a.wmul = 0;
    a.prune = 1; a.pdims = su.pdims; a.prune_margin = g.prune_margin > 0.0 ? g.prune_margin : (double)su.margin;
    a.tile_box = su.tile_box; a.qtile_box = su.qtile_box; a.qtile_thr = su.qtile_thr; a.qlb = su.qlb;
    a.nsplit_grid = su.nsplit; a.part = su.part; a.group_masks = 0;
    kde_sweep_f16_w32p_body<NB>(a, bid);
}

// ------------------------------------------------------------------------------------------------
// kde_cdf: CKDE::cdf.  The reference (CKDE.hpp:560-735 + KDE.cl.src:376-468) materialises, per tile of 64 test rows,
// the N x 64 weight matrix W (marginal KDE terms), the N x 64 conditional means, their normal cdf, the element-wise
// product and two column sums.  Here: the weights are the marginal sweep's 2^(s2 - m) (same MFMA + offset machinery),
// the conditional mean is linear, (x_q - mu_t(e_q)) / sigma_c = u_q - u_t with u = (x - b.e) / sigma_c precomputed per
// row, so a pair costs one erfc; cdf_q = sum_t w_t Phi(u_q - u_t) / sum_t w_t.  No evidence: w_t = 1.
// ------------------------------------------------------------------------------------------------
template <typename T>
__device__ __forceinline__ T half_erfc(T x);
template <>
__device__ __forceinline__ double half_erfc<double>(double x) { return 0.5 * erfc(x); }

// 1/2 erfc(x) for the fp64 CKDE::cdf kernel, branch-free: erfc(|x|) = erfcx(|x|) exp(-x^2) with erfcx from a table of
// degree-7 polynomials on [i/8, (i+1)/8) (48 intervals up to 6, relative error 6.5e-14: tools/erfcx_table.py; beyond 6
// the exponential alone is below 2^-52), the table staged in LDS, the exponential by the sweep's own 2^x.  About a third
// of the instructions of the library erfc.
#define PBN_ERFCX_INTERVALS 48
__device__ const double ERFCX_TABLE[PBN_ERFCX_INTERVALS * 8] = {
#include "erfcx_table.inc"
};
__device__ __forceinline__ double half_erfc_table(double x, const double* __restrict__ tab) {
    const double a = __builtin_fabs(x);
    const double ac = __builtin_fmin(a, 5.999999999);
    int idx;
    const double scaled = ac * 8.0;
    asm("v_cvt_i32_f64 %0, %1" : "=v"(idx) : "v"(scaled));                 // truncation = floor: ac >= 0
    const double r = __builtin_fma((double)idx, -0.125, ac) - 0.0625;     // centred in the interval
    const double* c = tab + idx * 8;
    double p = c[7];
    p = __builtin_fma(p, r, c[6]);
    p = __builtin_fma(p, r, c[5]);
    p = __builtin_fma(p, r, c[4]);
    p = __builtin_fma(p, r, c[3]);
    p = __builtin_fma(p, r, c[2]);
    p = __builtin_fma(p, r, c[1]);
    p = __builtin_fma(p, r, c[0]);
    const double e = exp2_f64<8>(-(a * a) * 0x1.71547652b82fep+0);        // exp(-a^2)
    const double h = 0.5 * p * e;
    return x >= 0.0 ? h : 1.0 - h;
}
template <>
__device__ __forceinline__ float half_erfc<float>(float x) { return 0.5f * erfcf(x); }

// MODE 0: weights only (CKDE::sample), 1: weights x normal cdf (CKDE::cdf), 2: sum w and sum sqrt(w) with the offset
// pinned at 0 (UCV: K_2H = sqrt of the un-normalised K_H; self pairs keep every exponent <= 0)
template <typename T, int KS, int QG, int MODE>
__global__ __launch_bounds__(256, 2) void kde_cdf_kernel(CdfArgs a) {
    constexpr bool CDF = MODE == 1;
    // KS == 0: the number of K steps is a run-time value (more than 16 evidence variables / UCV dimensions): the fragments of both
    // sides are read at every step instead of living in registers
    constexpr bool RT = KS == 0;
    constexpr int KSR = RT ? 1 : KS;
    const int ksn = RT ? a.KS : KS;
    using V = typename Tr<T>::vec4;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int lg = lane >> 4;
    int qx, split;
    xcd_block(qx, split);
    constexpr bool TABLE = CDF && sizeof(T) == 8;
    __shared__ double etab[TABLE ? PBN_ERFCX_INTERVALS * 8 : 1];
    if (TABLE) {   // before any wave leaves: the barrier needs them all
        for (int e = threadIdx.x; e < PBN_ERFCX_INTERVALS * 8; e += 256) etab[e] = ERFCX_TABLE[e];
        __syncthreads();
    }
    const int64_t qt0 = ((int64_t)qx * 4 + wave) * QG;
    if (qt0 >= a.nqtiles) return;
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;
    const T* __restrict__ Ap = (const T*)a.Apack;
    const T* __restrict__ Np = (const T*)a.nxpack;
    const T* __restrict__ Up = (const T*)a.utrain;
    const T* __restrict__ Bp = (const T*)a.Bpack;
    const T* __restrict__ NYp = (const T*)a.nypack;
    const T* __restrict__ UQp = (const T*)a.uquery;

    T b[QG][KSR], ny[QG], cm[QG], m[QG], uq[QG];
    int64_t qtg[QG];
    double sw[QG], sc[QG];
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        int64_t qt = qt0 + g < a.nqtiles ? qt0 + g : a.nqtiles - 1;
        qtg[g] = qt;
        if constexpr (!RT) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) b[g][ks] = Bp[(qt * KS + ks) * 64 + lane];
        }
        ny[g] = NYp[qt * 16 + (lane & 15)];
        uq[g] = CDF ? UQp[qt * 16 + (lane & 15)] : (T)0;
        sw[g] = 0.0; sc[g] = 0.0;
    }
    {   // offsets from the first tile
        const V nx = *(const V*)(Np + t0 * 16 + lg * 4);
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            V acc = nx + ny[g];
            if constexpr (RT) {
                for (int ks = 0; ks < ksn; ++ks) acc = Tr<T>::mfma(Ap[(t0 * ksn + ks) * 64 + lane], Bp[(qtg[g] * ksn + ks) * 64 + lane], acc);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = Tr<T>::mfma(Ap[(t0 * KS + ks) * 64 + lane], b[g][ks], acc);
            }
            m[g] = MODE == 2 ? (T)0 : colmax<T>(max4<T>(acc));
            cm[g] = ny[g] - m[g];
        }
    }
    for (int64_t t = t0; t < t1; ++t) {
        T af[KSR];
        if constexpr (!RT) {
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) af[ks] = Ap[(t * KS + ks) * 64 + lane];
        }
        const V nx = *(const V*)(Np + t * 16 + lg * 4);
        V ut = {0, 0, 0, 0};
        if (CDF) ut = *(const V*)(Up + t * 16 + lg * 4);
#pragma unroll
        for (int g = 0; g < QG; ++g) {
            V acc = nx + cm[g];
            if constexpr (RT) {
                for (int ks = 0; ks < ksn; ++ks) acc = Tr<T>::mfma(Ap[(t * ksn + ks) * 64 + lane], Bp[(qtg[g] * ksn + ks) * 64 + lane], acc);
            } else {
#pragma unroll
                for (int ks = 0; ks < KS; ++ks) acc = Tr<T>::mfma(af[ks], b[g][ks], acc);
            }
            T w0 = Tr<T>::ex2_hi(acc[0]), w1 = Tr<T>::ex2_hi(acc[1]), w2 = Tr<T>::ex2_hi(acc[2]), w3 = Tr<T>::ex2_hi(acc[3]);
            T ts = (w0 + w1) + (w2 + w3);
            if (MODE != 2 && __builtin_expect(__any(!(ts < Tr<T>::big())), 0)) {
                const T mx = colmax<T>(max4<T>(acc));
                if (mx > (T)0) {
                    m[g] += mx;
                    cm[g] = ny[g] - m[g];
                    const double f = exp2(-(double)mx);
                    sw[g] *= f; sc[g] *= f;
                    acc -= mx;
                }
                w0 = Tr<T>::ex2_hi(acc[0]); w1 = Tr<T>::ex2_hi(acc[1]); w2 = Tr<T>::ex2_hi(acc[2]); w3 = Tr<T>::ex2_hi(acc[3]);
                ts = (w0 + w1) + (w2 + w3);
            }
            // Phi((x_q - mu_t)/sigma_c) = 1/2 erfc((u_t - u_q)), u pre-divided by sqrt 2 (KDE.cl.src:448-456)
            sw[g] += (double)ts;
            if (MODE == 2) sc[g] += (double)((sqrt(w0) + sqrt(w1)) + (sqrt(w2) + sqrt(w3)));
            if (CDF) {
                auto phi = [&](T v) -> T {
                    if constexpr (TABLE) return (T)half_erfc_table((double)v, etab);
                    else return half_erfc<T>(v);
                };
                const T c = (w0 * phi(ut[0] - uq[g]) + w1 * phi(ut[1] - uq[g])) + (w2 * phi(ut[2] - uq[g]) + w3 * phi(ut[3] - uq[g]));
                sc[g] += (double)c;
            }
        }
    }
#pragma unroll
    for (int g = 0; g < QG; ++g) {
        double s = sw[g], c = sc[g];
        s += __shfl_xor(s, 16); s += __shfl_xor(s, 32);
        c += __shfl_xor(c, 16); c += __shfl_xor(c, 32);
        if (lg == 0 && qt0 + g < a.nqtiles) {
            double* o = a.part + ((int64_t)split * a.nqtiles * 16 + (qt0 + g) * 16 + lane) * 4;
            o[0] = (double)m[g]; o[1] = s; o[2] = c; o[3] = 0.0;
        }
    }
}

__global__ __launch_bounds__(256) void kde_cdf_finish_kernel(const double* __restrict__ part, int nsplit, int64_t nqtiles, int64_t nq,
                                                              double* __restrict__ out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const double* p = part + q * 4;
    const int64_t stride = nqtiles * 16 * 4;
    double m = p[0], sw = p[1], sc = p[2];
    for (int sp = 1; sp < nsplit; ++sp) {
        const double* pp = p + sp * stride;
        const double M = m > pp[0] ? m : pp[0];
        const double f1 = exp2(m - M), f2 = exp2(pp[0] - M);
        sw = sw * f1 + pp[1] * f2;
        sc = sc * f1 + pp[2] * f2;
        m = M;
    }
    out[q] = sc / sw;
}

// ------------------------------------------------------------------------------------------------
// kde_finish: per query merge the split partials (fixed order), logl = lognorm + ln2*(m + log2 sum)
// [CKDE: joint - marginal], optional logl store, deterministic block tree sum.
// ------------------------------------------------------------------------------------------------
template <bool COND>
__global__ __launch_bounds__(256) void kde_finish_kernel(FinishArgs a) {
    constexpr int P = COND ? 4 : 2;
    constexpr double LN2 = 0.693147180559945309417232121458;
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double val = 0.0, val_marg = 0.0;
    if (q < a.nq) {
        const double* p = a.part + q * P;
        const int64_t stride = a.nqtiles * 16 * P;
        // two passes: the largest offset first, then the sums scaled to it in split order - one 2^x per partial and no
        // dependent chain (the running-rescale form cost two library exp2 per split in sequence: with the 157 splits of a
        // 90 000 x 10 000 sweep this kernel took 65 us against the sweep's 250).  Integer offsets (the fp64 sweeps' own) make every
        // factor an exact power of two, so the result is the one of the running form bit for bit.
        double m = p[0], mjj = COND ? p[2] : 0.0;
        for (int sp = 1; sp < a.nsplit; ++sp) {
            const double* pp = p + sp * stride;
            const double m2 = pp[0];
            m = m > m2 ? m : m2;
            if (COND) { const double m3 = pp[2]; mjj = mjj > m3 ? mjj : m3; }
        }
        double s = 0.0, sj = 0.0;
#pragma unroll 4
        for (int sp = 0; sp < a.nsplit; ++sp) {
            const double* pp = p + sp * stride;
            s += pp[1] * exp2(pp[0] - m);
            if (COND) sj += pp[3] * exp2(pp[2] - mjj);
        }
        double l = a.lognorm + LN2 * (m + log2(s));
        if (COND) {
            const double lj = a.lognorm + LN2 * (mjj + log2(sj)), lm = a.lognorm_marg + LN2 * (m + log2(s));
            l = lj - lm;
            if (a.block_sums_marg) { l = lj; val_marg = lm; }   // the two sums separately (score engine's set cache)
        }
        if (a.logl) a.logl[a.scatter ? (int64_t)a.scatter[q] : q] = l;
        val = l;
    }
    __shared__ double red[256];
    red[threadIdx.x] = val;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.block_sums) a.block_sums[blockIdx.x] = red[0];
    if (COND && a.block_sums_marg) {
        __syncthreads();
        red[threadIdx.x] = val_marg;
        __syncthreads();
#pragma unroll
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) a.block_sums_marg[blockIdx.x] = red[0];
    }
}

// Final fixed-order reduction of the per-block sums (replaces the multi-pass sum1d of
// opencl_config.hpp:344-397 with one launch).
__global__ __launch_bounds__(256) void reduce_final_kernel(const double* __restrict__ in, int64_t n, double* out) {
    __shared__ double red[256];
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 256) v += in[i];
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) *out = red[0];
}

__global__ __launch_bounds__(256) void diff_kernel(double* __restrict__ out, const double* __restrict__ a, const double* __restrict__ b, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) out[i] = a[i] - b[i];
}

// ------------------------------------------------------------------------------------------------
// launchers
// ------------------------------------------------------------------------------------------------
void launch_diff(double* out, const double* a, const double* b, int64_t n, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(diff_kernel, dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, st, out, a, b, n);
    HIP_CHECK(hipGetLastError());
}

void launch_pack_classic(const PackArgs& a, int dtype, hipStream_t st) {
    const int64_t npad = a.ntiles * 16;
    if (npad == 0) return;
    dim3 grid((unsigned)ceil_div(npad, 256)), block(256);
    if (dtype == PBN_F64 && a.src_f32) hipLaunchKernelGGL((pack_rows_kernel<double, float>), grid, block, 0, st, a);
    else if (dtype == PBN_F64) hipLaunchKernelGGL(pack_rows_kernel<double>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(pack_rows_kernel<float>, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------
// Wide models (more than 32 whitened dimensions): generic pack and sweep, see kde_kernels.hpp
// ------------------------------------------------------------------------------------------------
template <typename TS>
__global__ __launch_bounds__(256) void pack_rows_wide_kernel(WidePackArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= a.ntiles * 16) return;
    const int64_t tile = r >> 4;
    const int idx = (int)(r & 15);
    const int d = a.d, KS = a.KS;
    const bool valid = r < a.n;
    int64_t src = 0;
    if (valid) {
        const int64_t lr = r < a.n0 ? a.row0 + r : a.row1 + (r - a.n0);
        src = a.rows ? (int64_t)a.rows[lr] : lr;
    }
    double nrm = 0.0;
    for (int i = 0; i < KS * 4; ++i) {
        double z = 0.0;
        if (valid && i < a.dm) {
            // z_i = sum_{j <= i} W[i][j] (x_j - mu_j): the row's coordinates are re-read per i (L1 / L2 hits) instead of living in a
            // per-thread array of unknown size
            const double* w = a.W + (size_t)i * a.ldw;
            for (int j = 0; j <= i; ++j) {
                const double x = (double)((const TS*)a.base + (int64_t)a.cols[j] * a.ld)[src] - a.mu[j];
                z = __builtin_fma(w[j], x, z);
            }
        }
        nrm = __builtin_fma(z, z, nrm);
        a.pack[(tile * KS + (i >> 2)) * 64 + (i & 3) * 16 + idx] = z;
    }
    double nv = -0.5 * nrm;
    if (!valid) nv = a.is_query ? 0.0 : PBN_PAD_NORM;
    if (a.is_query) {
        a.npack[tile * 16 + idx] = nv;
    } else {
        const int lg = idx & 3, i = idx >> 2;   // f64 C-row order: crow(lg, i) == idx
        a.npack[tile * 16 + lg * 4 + i] = nv;
    }
    if (a.upack) {   // CKDE::cdf: standardised "x - b.e" of the row, in the norm's layout
        double u = 0.0;
        if (valid)
            for (int j = 0; j < d; ++j) u = __builtin_fma(a.wu[j], (double)((const TS*)a.base + (int64_t)a.cols[j] * a.ld)[src] - a.mu[j], u);
        if (a.is_query) a.upack[tile * 16 + idx] = u;
        else a.upack[tile * 16 + (idx & 3) * 4 + (idx >> 2)] = u;
    }
}

void launch_pack_wide(const WidePackArgs& a, hipStream_t st) {
    const int64_t npad = a.ntiles * 16;
    if (npad == 0) return;
    dim3 grid((unsigned)ceil_div(npad, 256)), block(256);
    if (a.src_f32) hipLaunchKernelGGL(pack_rows_wide_kernel<float>, grid, block, 0, st, a);
    else hipLaunchKernelGGL(pack_rows_wide_kernel<double>, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// one wave = one group of 16 queries; the B fragments come from memory at every K step (the wave's 16 queries are the same for all
// tiles: L1 hits), the offset is raised tile by tile (online logsumexp, integer offsets), 2^x by the degree-8 polynomial
__global__ __launch_bounds__(256) void kde_sweep_wide_kernel(SweepArgs a, int KS) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, lg = lane >> 4;
    const int64_t qt = (int64_t)blockIdx.x * 4 + wave;
    const int split = blockIdx.y;
    if (qt >= a.nqtiles) return;
    const int64_t t0 = (int64_t)split * a.tiles_per_split;
    const int64_t t1 = (t0 + a.tiles_per_split < a.ntiles) ? t0 + a.tiles_per_split : a.ntiles;
    const PBN_GLOBAL double* __restrict__ Ap = (const PBN_GLOBAL double*)a.Apack;
    const PBN_GLOBAL double* __restrict__ Np = (const PBN_GLOBAL double*)a.nxpack;
    const PBN_GLOBAL double* __restrict__ Bp = (const PBN_GLOBAL double*)a.Bpack + qt * KS * 64 + lane;
    const double ny = ((const PBN_GLOBAL double*)a.nypack)[qt * 16 + (lane & 15)];
    double m = -INFINITY, sum = 0.0;
    for (int64_t t = t0; t < t1; ++t) {
        d4 acc = *(const PBN_GLOBAL d4*)(Np + t * 16 + lg * 4) + ny;
        const PBN_GLOBAL double* __restrict__ At = Ap + t * KS * 64 + lane;
        for (int ks = 0; ks < KS; ++ks) acc = Tr<double>::mfma(At[ks * 64], Bp[ks * 64], acc);
        const double vmax = colmax<double>(max4<double>(acc));   // uniform over the four lanes of a query column
        if (vmax > m) {
            const double nm = __builtin_ceil(vmax);
            sum *= exp2(m - nm);   // m = -inf: the sum is still 0
            m = nm;
        }
        sum += (Tr<double>::ex2_hi(acc[0] - m) + Tr<double>::ex2_hi(acc[1] - m)) + (Tr<double>::ex2_hi(acc[2] - m) + Tr<double>::ex2_hi(acc[3] - m));
    }
    sum += __shfl_xor(sum, 16);
    sum += __shfl_xor(sum, 32);
    if (lg == 0) {
        PBN_GLOBAL double* o = (PBN_GLOBAL double*)a.part + ((int64_t)split * a.nqtiles * 16 + qt * 16 + lane) * 2;
        o[0] = m;
        o[1] = sum;
    }
}

void launch_sweep_wide(const SweepArgs& a, int KS, int nsplit, hipStream_t st) {
    if (a.nqtiles == 0) return;
    dim3 grid((unsigned)ceil_div(a.nqtiles, 4), (unsigned)nsplit), block(256);
    hipLaunchKernelGGL(kde_sweep_wide_kernel, grid, block, 0, st, a, KS);
    HIP_CHECK(hipGetLastError());
}

void launch_far_fix(const PackArgs& q, const void* Apack, const void* Axpack, int NB, int64_t n_train, int64_t ntiles, double* part, int nsplit, int64_t nqtiles,
                    bool cond, hipStream_t st) {
    if (!q.far_flag || nqtiles == 0) return;
    const dim3 grid((unsigned)nqtiles), block(256);
    if (cond) hipLaunchKernelGGL(kde_far_fix_kernel<true>, grid, block, 0, st, q, (const hf8*)Apack, (const hf8*)Axpack, NB, n_train, ntiles, part, nsplit, nqtiles);
    else hipLaunchKernelGGL(kde_far_fix_kernel<false>, grid, block, 0, st, q, (const hf8*)Apack, (const hf8*)Axpack, NB, n_train, ntiles, part, nsplit, nqtiles);
    HIP_CHECK(hipGetLastError());
}

void launch_max_norm2(const PackArgs& a, int src_dtype, double* dev_out, hipStream_t st) {
    if (a.n <= 0) return;
    dim3 grid((unsigned)ceil_div(a.n, 256)), block(256);
    if (src_dtype == PBN_F64) hipLaunchKernelGGL(max_norm2_kernel<double>, grid, block, 0, st, a, (unsigned long long*)dev_out);
    else hipLaunchKernelGGL(max_norm2_kernel<float>, grid, block, 0, st, a, (unsigned long long*)dev_out);
    HIP_CHECK(hipGetLastError());
}

template <typename T, int CDF>
static void launch_cdf_t(const CdfArgs& a, int KS, dim3 grid, hipStream_t st) {
    dim3 block(256);
    switch (KS) {
        case 1: hipLaunchKernelGGL((kde_cdf_kernel<T, 1, 2, CDF>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((kde_cdf_kernel<T, 2, 2, CDF>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((kde_cdf_kernel<T, 3, 2, CDF>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((kde_cdf_kernel<T, 4, 2, CDF>), grid, block, 0, st, a); break;
        default:
            if constexpr (sizeof(T) == 8) hipLaunchKernelGGL((kde_cdf_kernel<T, 0, 2, CDF>), grid, block, 0, st, a);   // runtime-sized (a.KS)
            else throw invalid_error("CKDE::cdf / sample / UCV: more than 16 dimensions take fp64 fragments");
    }
    HIP_CHECK(hipGetLastError());
}

// utrain == nullptr: weights only (sum w per split; used by CKDE::sample to locate the sampled instance).
void launch_cdf(const CdfArgs& a_in, int dtype, int KS, int nsplit, hipStream_t st) {
    CdfArgs a = a_in;
    a.KS = KS;
    dim3 grid((unsigned)ceil_div(a.nqtiles, 4 * 2), (unsigned)nsplit);
    const bool cdf = a.utrain != nullptr;
    if (dtype == PBN_F64) { if (cdf) launch_cdf_t<double, 1>(a, KS, grid, st); else launch_cdf_t<double, 0>(a, KS, grid, st); }
    else                  { if (cdf) launch_cdf_t<float, 1>(a, KS, grid, st); else launch_cdf_t<float, 0>(a, KS, grid, st); }
}

// UCV pair sums: part[split][query] = (0, sum_t w, sum_t sqrt w, 0) with w = 2^(s2(t, q)); then the two totals over the
// first nq queries and all splits, fixed order.
__global__ __launch_bounds__(256) void ucv_block_sums_kernel(const double* __restrict__ part, int nsplit, int64_t nqtiles, int64_t nq,
                                                              double* __restrict__ block_w, double* __restrict__ block_r) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double w = 0.0, r = 0.0;
    if (q < nq) {
        const int64_t stride = nqtiles * 16 * 4;
        for (int sp = 0; sp < nsplit; ++sp) {
            w += part[sp * stride + q * 4 + 1];
            r += part[sp * stride + q * 4 + 2];
        }
    }
    __shared__ double red[256];
    for (int pass = 0; pass < 2; ++pass) {
        red[threadIdx.x] = pass ? r : w;
        __syncthreads();
#pragma unroll
        for (int s = 128; s > 0; s >>= 1) {
            if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
            __syncthreads();
        }
        if (threadIdx.x == 0) (pass ? block_r : block_w)[blockIdx.x] = red[0];
        __syncthreads();
    }
}

void launch_ucv(const CdfArgs& a_in, int dtype, int KS, int nsplit, int64_t nq, double* block_scratch, double* dev_out2, hipStream_t st) {
    CdfArgs a = a_in;
    a.KS = KS;
    dim3 grid((unsigned)ceil_div(a.nqtiles, 4 * 2), (unsigned)nsplit);
    if (dtype == PBN_F64) launch_cdf_t<double, 2>(a, KS, grid, st); else launch_cdf_t<float, 2>(a, KS, grid, st);
    const int64_t nblocks = ceil_div(nq, 256);
    hipLaunchKernelGGL(ucv_block_sums_kernel, dim3((unsigned)nblocks), dim3(256), 0, st, a.part, nsplit, a.nqtiles, nq, block_scratch,
                       block_scratch + nblocks);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const double*)block_scratch, nblocks, dev_out2);
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const double*)(block_scratch + nblocks), nblocks, dev_out2 + 1);
    HIP_CHECK(hipGetLastError());
}

void launch_cdf_finish(const double* part, int nsplit, int64_t nqtiles, int64_t nq, double* dev_out, hipStream_t st) {
    if (nq == 0) return;
    hipLaunchKernelGGL(kde_cdf_finish_kernel, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, st, part, nsplit, nqtiles, nq, dev_out);
    HIP_CHECK(hipGetLastError());
}

// Exponent distance below the queries' sum bound beyond which a training tile is skipped.  PBN_PRUNE_MARGIN (fp64, 52) /
// PBN_PRUNE_MARGIN_F32 (fp32, 40) are the values AT 10^6 TRAINING ROWS; for n rows the margin is that + log2(n / 10^6), so that the
// bound of what pruning can drop - at most n terms of 2^-margin of the sum each - is the same fraction of the sum whatever the size
// of the training set: 10^6 x 2^-52 = 2.2e-10 (fp32: 10^6 x 2^-40 = 9.1e-7).  With a constant margin the bound grew linearly with n
// (and was needlessly tight for the 10^4-10^5-row folds and slices of the score engine: 90 000 rows -> 48.5, 450 000 -> 50.9;
// 4 x 10^6 -> 54).  PBN_PRUNE_MARGIN_ADAPT=0 keeps the constant.
// Round 4: sweeps whose result is a SUM over the test rows (slogl, the score engine's terms: `sum_only`) carry a per-term arithmetic
// error of 1.4e-7 anyway (2^f on the fp32 transcendental unit), fp32 tables one of ~1e-5 (2^-24 |z|^2): their margins are set so that
// the dropped-mass bound matches - fp64 sums 43 at 10^6 rows (1.1e-7 of a sum), fp32 36 (1.5e-5) - while per-row logl outputs keep 52
// (2.2e-10).  cv64 3.06 -> 2.74 s, C3's first iteration 14.1 -> 12.6 s, C5 9.1 -> 8.7 s with the same operator sequences
// (profiles/r4/margin_probe4.txt).  PBN_PRUNE_MARGIN pins the fp64 value for both kinds, PBN_PRUNE_MARGIN_SUM the sum-only one,
// PBN_PRUNE_MARGIN_F32 the fp32 one; read per call (a host getenv per sweep launch) so that tests can pin them inside one process.
double prune_margin(int dtype, int64_t n_train, bool sum_only) {
    const bool adapt = PBN_TUNE(PRUNE_MARGIN_ADAPT, 1) != 0;
    double base;
    if (use_f16x2(dtype)) base = knob_double("PBN_PRUNE_MARGIN_F32", (double)PBN_PRUNE_MARGIN_F32);
    else base = knob_double("PBN_PRUNE_MARGIN", sum_only ? knob_double("PBN_PRUNE_MARGIN_SUM", (double)PBN_PRUNE_MARGIN_SUM) : (double)PBN_PRUNE_MARGIN);
    if (!adapt || n_train <= 0) return base;
    const double m = base + std::log2((double)n_train / 1e6);
    return m < 8.0 ? 8.0 : m;
}

bool use_f16x2(int dtype) {
    static const int v = PBN_TUNE(F32_F16X2, 1);   // (0: fp32 tables on the f32 MFMA kernels - the round-1 path, kept for comparisons)
    return v != 0 && dtype == PBN_F32;
}

int f16x2_mfmas(int dm) { return f16x2_blocks(dm); }   // f16x2: three (four where they fit) slots per dimension + the training norm

// the W32 form of the plain unpruned fp32 sweep: one or two 32-slot blocks whose last three slots are free (up to 8 / 19 whitened dimensions)
bool f16x2_w32(int dm, int NB) {
    return knob_int("PBN_F32_W32", 1) != 0 && (NB == 1 || NB == 2) && f16x2_spd(dm) * dm + 6 <= 32 * NB;   // read per call: tests compare the two forms in one process
}

// ... and of the pruned plain fp32 sweeps (stand-alone handles, grouped launches): one 32-slot block whose last three slots are free
bool f16x2_w32p(int dm, int NB) {
    return PBN_F16_W32P != 0 && knob_int("PBN_F32_W32", 1) != 0 && NB == 1 && f16x2_spd(dm) * dm + 6 <= 32;
}

void launch_pack(const PackArgs& a, int dtype, hipStream_t st) {
    const int64_t npad = a.ntiles * 16;
    if (npad == 0) return;
    dim3 grid((unsigned)ceil_div(npad, 256)), block(256);
    if (use_f16x2(dtype)) {
        hipLaunchKernelGGL(pack_rows_f16_kernel, grid, block, 0, st, a);
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (dtype == PBN_F64 && a.src_f32)
        hipLaunchKernelGGL((pack_rows_kernel<double, float>), grid, block, 0, st, a);
    else if (dtype == PBN_F64)
        hipLaunchKernelGGL(pack_rows_kernel<double>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(pack_rows_kernel<float>, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// one launch site of kde_sweep_kernel: the plain fp64 shapes exist twice - FAST (SweepArgs::fast: the sweep's result is a sum over the
// test rows) and exact-polynomial (per-row logl outputs); CKDE-fused and fp32 shapes only in the second form
#define PBN_LAUNCH_SWEEP(KSv, CONDv, QGv, FOLDv, PRUNEv, WMULv)                                                                            \
    do {                                                                                                                                 \
        if constexpr (sizeof(T) == 8 && !(CONDv)) {                                                                                      \
            if (a.fast) {                                                                                                                \
                hipLaunchKernelGGL((kde_sweep_kernel<T, KSv, CONDv, QGv, FOLDv, PRUNEv, WMULv, true>), grid, block, 0, st, a);             \
                break;                                                                                                                   \
            }                                                                                                                            \
        }                                                                                                                                \
        hipLaunchKernelGGL((kde_sweep_kernel<T, KSv, CONDv, QGv, FOLDv, PRUNEv, WMULv, false>), grid, block, 0, st, a);                    \
    } while (0)

template <typename T, bool COND, bool FOLD>
static void launch_sweep_tf(const SweepArgs& a, int KS, dim3 grid, hipStream_t st) {
    constexpr int QG = SweepQG<sizeof(T) == 8, COND>::value;
    dim3 block(256);
    if (a.prune) {   // fp64, at most 6 marginal dimensions (KS <= 2)
        if constexpr (sizeof(T) == 8) {
            constexpr int QGP = COND ? PBN_QG_PRUNE_COND : PBN_QG_PRUNE;   // query groups per wave of the pruned kernels
            block = dim3(sweep_block_threads(true));
            grid = dim3((unsigned)(ceil_div(a.nqtiles, QGP) * a.nsplit_grid));   // one wave per workgroup, placed by pruned_block
            if constexpr (!COND && !FOLD) {
                if (a.wmul && KS <= 2) {   // 4 / 8 marginal dimensions: the pruned shapes without a free K slot
                    if (KS == 1) PBN_LAUNCH_SWEEP(1, false, QGP, false, true, true);
                    else PBN_LAUNCH_SWEEP(2, false, QGP, false, true, true);
                    HIP_CHECK(hipGetLastError());
                    return;
                }
            }
            if (KS == 1) PBN_LAUNCH_SWEEP(1, COND, QGP, FOLD, true, false);
            else if (KS == 2) PBN_LAUNCH_SWEEP(2, COND, QGP, FOLD, true, false);
            else throw invalid_error("KDE: pruned sweeps cover at most 8 whitened dimensions");
            HIP_CHECK(hipGetLastError());
            return;
        } else {
            throw invalid_error("KDE: pruned sweeps are fp64 only");
        }
    }
    if constexpr (sizeof(T) == 8 && !COND && !FOLD) {
        if (a.wmul) {
            switch (KS) {
                case 1: PBN_LAUNCH_SWEEP(1, false, QG, false, false, true); break;
                case 2: PBN_LAUNCH_SWEEP(2, false, QG, false, false, true); break;
                default: throw invalid_error("KDE: weighted-norm sweeps cover at most 8 whitened dimensions");
            }
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    if constexpr (sizeof(T) == 8 && !FOLD) {   // 17-32 dimensions: fp64, norms added per value, two query groups per wave (sweep_qg)
        switch (KS) {
            case 5: PBN_LAUNCH_SWEEP(5, COND, 2, false, false, false); HIP_CHECK(hipGetLastError()); return;
            case 6: PBN_LAUNCH_SWEEP(6, COND, 2, false, false, false); HIP_CHECK(hipGetLastError()); return;
            case 7: PBN_LAUNCH_SWEEP(7, COND, 2, false, false, false); HIP_CHECK(hipGetLastError()); return;
            case 8: PBN_LAUNCH_SWEEP(8, COND, 2, false, false, false); HIP_CHECK(hipGetLastError()); return;
            default: break;
        }
    }
    switch (KS) {
        case 1: PBN_LAUNCH_SWEEP(1, COND, QG, FOLD, false, false); break;
        case 2: PBN_LAUNCH_SWEEP(2, COND, QG, FOLD, false, false); break;
        case 3: PBN_LAUNCH_SWEEP(3, COND, QG, FOLD, false, false); break;
        case 4: PBN_LAUNCH_SWEEP(4, COND, QG, FOLD, false, false); break;
        default: throw invalid_error("KDE: this many whitened dimensions per sweep are not supported for the table's type");
    }
    HIP_CHECK(hipGetLastError());
}
template <typename T, bool COND>
static void launch_sweep_t(const SweepArgs& a, int KS, dim3 grid, hipStream_t st) {
    if (a.fold) launch_sweep_tf<T, COND, true>(a, KS, grid, st); else launch_sweep_tf<T, COND, false>(a, KS, grid, st);
}

void launch_prune_keys(const PackArgs& a, int dtype, int zd, int kd, double* zrow, uint32_t* keys, int32_t* iota, hipStream_t st) {
    if (a.n == 0) return;
    const dim3 grid((unsigned)ceil_div(a.n, 256)), block(256);
    const double inv_cell = 1.0 / prune_key_cell(kd);
    static const int hnd = PBN_TUNE(PRUNE_HILBERT_ND, 1);   // Hilbert order at three / four key dimensions too (two: always)
    if (dtype == PBN_F64 && a.src_f32) hipLaunchKernelGGL((prune_keys_kernel<double, float>), grid, block, 0, st, a, zd, kd, zrow, keys, iota, inv_cell, hnd);
    else if (dtype == PBN_F64) hipLaunchKernelGGL(prune_keys_kernel<double>, grid, block, 0, st, a, zd, kd, zrow, keys, iota, inv_cell, hnd);
    else hipLaunchKernelGGL(prune_keys_kernel<float>, grid, block, 0, st, a, zd, kd, zrow, keys, iota, inv_cell, hnd);
    HIP_CHECK(hipGetLastError());
}
void launch_tile_boxes(const double* zrow, const int32_t* perm, int64_t n, int zd, int pd, double* box, double* zsorted, hipStream_t st) {
    if (n == 0) return;
    hipLaunchKernelGGL(tile_box_kernel, dim3((unsigned)ceil_div(ceil_div(n, 16) * 16, 256)), dim3(256), 0, st, zrow, perm, n, zd, pd, box, zsorted);
    HIP_CHECK(hipGetLastError());
}
// one 64-lane block per (split, batch): the bounding box of up to 64 tile boxes (the first level of the pruned sweeps' tile walk)
__global__ __launch_bounds__(64) void batch_box_kernel(const double* __restrict__ tile_box, int pd, int64_t ntiles, int64_t tps, int nbps, double* __restrict__ out) {
    const int split = blockIdx.x / nbps, k = blockIdx.x - split * nbps;
    const int64_t t0 = (int64_t)split * tps, t1 = t0 + tps < ntiles ? t0 + tps : ntiles;
    const int64_t t = t0 + 64 * (int64_t)k + (int)threadIdx.x;
    double lo[PBN_PRUNE_PD], hi[PBN_PRUNE_PD];
#pragma unroll
    for (int i = 0; i < PBN_PRUNE_PD; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
    if (t < t1) {
        const double* bx = tile_box + t * 2 * pd;
#pragma unroll
        for (int i = 0; i < PBN_PRUNE_PD; ++i)
            if (i < pd) { lo[i] = bx[i]; hi[i] = bx[pd + i]; }
    }
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < PBN_PRUNE_PD; ++i) {
            const double l = __shfl_xor(lo[i], off), h = __shfl_xor(hi[i], off);
            lo[i] = l < lo[i] ? l : lo[i];
            hi[i] = h > hi[i] ? h : hi[i];
        }
    }
    if (threadIdx.x == 0) {
        double* bb = out + (int64_t)blockIdx.x * 2 * pd;
        for (int i = 0; i < pd; ++i) { bb[i] = lo[i]; bb[pd + i] = hi[i]; }
    }
}
void launch_batch_boxes(const double* tile_box, int pd, int64_t ntiles, int64_t tiles_per_split, int nsplit, double* out, hipStream_t st) {
    const int nbps = (int)ceil_div(tiles_per_split, 64);
    if (ntiles == 0 || nsplit <= 0) return;
    hipLaunchKernelGGL(batch_box_kernel, dim3((unsigned)((int64_t)nsplit * nbps)), dim3(64), 0, st, tile_box, pd, ntiles, tiles_per_split, nbps, out);
    HIP_CHECK(hipGetLastError());
}
void launch_query_prepass(const double* zq_row, const int32_t* qperm, int64_t nq, const uint32_t* qkeys_sorted, const double* ztrain_sorted,
                          const uint32_t* tkeys_sorted, int64_t n, int zd, int pd, double* qbox, double* qthr, double* qlb, hipStream_t st,
                          const double* subpart, int P, int which, double log2_nsub, const double* tile_box) {
    if (nq == 0) return;
    static const int sum_bound = PBN_TUNE(GROUP_SUM_BOUND, 1);
    static const int tile_window = std::max(0, PBN_TUNE(GROUP_TILE_WINDOW, 256));
    hipLaunchKernelGGL(query_prepass_kernel, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, st, zq_row, qperm, nq, qkeys_sorted, ztrain_sorted,
                       tkeys_sorted, n, zd, pd, qbox, qthr, qlb, subpart, P, which, log2_nsub, sum_bound, tile_box, tile_window);
    HIP_CHECK(hipGetLastError());
}

bool sweep_folds_norm(int dtype, bool cond, int KS, int dm) {
    static const int v = PBN_TUNE(SWEEP_FOLD, 1);
    return v != 0 && !use_f16x2(dtype) && dm % 4 != 0 && KS <= 4;   // more than 16 dimensions: one form only
}

bool sweep_weights_norm(int dtype, bool cond, int KS, int dm) {
    static const int v = PBN_TUNE(SWEEP_WMUL, 1);
    return v != 0 && dtype == PBN_F64 && !cond && dm % 4 == 0 && KS <= 2;   // KS 3, 4: 169 / 181 VGPRs, a wave per SIMD lost
}

int sweep_qg(int dtype, bool cond, int KS, bool prune) {
    if (prune && dtype == PBN_F64) return cond ? PBN_QG_PRUNE_COND : PBN_QG_PRUNE;
    if (prune && use_f16x2(dtype) && !cond) return PBN_F16_QG_PRUNE;   // (what the grids of the pruned launches - stand-alone and grouped - are sized with)
    if (KS > 4) return 2;   // more than 16 (fp32: 20) dimensions: two query groups per wave (fragment registers); KS = MFMAs per tile pair
    if (dtype == PBN_F64) return cond ? SweepQG<true, true>::value : SweepQG<true, false>::value;
    return cond ? SweepQG<false, true>::value : SweepQG<false, false>::value;
}

static std::atomic<unsigned long long> g_w32_launches{0};   // measurement aid (pbn_debug_w32_launches): launches of the W32 form
template <bool COND>
static void launch_sweep_f16(const SweepArgs& a, int NB, dim3 grid, hipStream_t st) {
    dim3 block(256);
    if (a.prune) {   // at most 6 marginal dimensions: 27 f16 slots, one MFMA (two are kept instantiated)
        block = dim3(sweep_block_threads(true));
        constexpr int QGP = PBN_F16_QG_PRUNE;
        grid = dim3((unsigned)(ceil_div(a.nqtiles, QGP) * a.nsplit_grid));   // one wave (QGP query groups) per workgroup, placed by pruned_block
        if constexpr (!COND) {
            if (a.w32 && NB == 1) {   // paired kept tiles on 32x32x16 MFMAs (kde_sweep_f16_w32p_body)
                ++g_w32_launches;
                hipLaunchKernelGGL((kde_sweep_f16_w32p_kernel<1>), grid, block, 0, st, a);
                HIP_CHECK(hipGetLastError());
                return;
            }
        }
        if (NB == 1) hipLaunchKernelGGL((kde_sweep_f16_kernel<1, COND, QGP, true>), grid, block, 0, st, a);
        else if (NB == 2) hipLaunchKernelGGL((kde_sweep_f16_kernel<2, COND, QGP, true>), grid, block, 0, st, a);
        else throw invalid_error("KDE: pruned fp32 sweeps cover at most 10 whitened dimensions");
        HIP_CHECK(hipGetLastError());
        return;
    }
    if constexpr (!COND) {
        if (a.w32 && (NB == 1 || NB == 2)) {   // same grid: a wave's four 16-query groups are its two 32-query super-groups
            ++g_w32_launches;
            if (NB == 1) hipLaunchKernelGGL((kde_sweep_f16_w32_kernel<1>), grid, block, 0, st, a);
            else hipLaunchKernelGGL((kde_sweep_f16_w32_kernel<2>), grid, block, 0, st, a);
            HIP_CHECK(hipGetLastError());
            return;
        }
    }
    switch (NB) {
        case 1: hipLaunchKernelGGL((kde_sweep_f16_kernel<1, COND, 4, false>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((kde_sweep_f16_kernel<2, COND, 4, false>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((kde_sweep_f16_kernel<3, COND, 4, false>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((kde_sweep_f16_kernel<4, COND, 4, false>), grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL((kde_sweep_f16_kernel<5, COND, 2, false>), grid, block, 0, st, a); break;   // 21-32 dimensions (sweep_qg: 2)
        case 6: hipLaunchKernelGGL((kde_sweep_f16_kernel<6, COND, 2, false>), grid, block, 0, st, a); break;
        case 7: hipLaunchKernelGGL((kde_sweep_f16_kernel<7, COND, 2, false>), grid, block, 0, st, a); break;
        default: throw invalid_error("KDE: more than 32 whitened dimensions per sweep are not supported");
    }
    HIP_CHECK(hipGetLastError());
}

void launch_sweep(const SweepArgs& a_in, int dtype, int KS, bool cond, int nsplit, hipStream_t st) {
    SweepArgs a = a_in;
    a.nsplit_grid = nsplit;
    dim3 grid((unsigned)ceil_div(a.nqtiles, 4 * sweep_qg(dtype, cond, KS, a.prune != 0)), (unsigned)nsplit);
    if (use_f16x2(dtype)) {  // KS carries the number of 32-slot f16 MFMAs
        if (cond) launch_sweep_f16<true>(a, KS, grid, st); else launch_sweep_f16<false>(a, KS, grid, st);
        return;
    }
    if (dtype == PBN_F64) {
        if (cond) launch_sweep_t<double, true>(a, KS, grid, st); else launch_sweep_t<double, false>(a, KS, grid, st);
    } else {
        if (cond) launch_sweep_t<float, true>(a, KS, grid, st); else launch_sweep_t<float, false>(a, KS, grid, st);
    }
}

void launch_moment_grouped(const GSweepArgs& g, int d, hipStream_t st) {
    if (g.total_wg == 0) return;
    const dim3 grid((unsigned)g.total_wg), block(64);
    if (d == 1) hipLaunchKernelGGL(kde_moment_group_kernel<1>, grid, block, 0, st, g);
    else if (d == 2) hipLaunchKernelGGL(kde_moment_group_kernel<2>, grid, block, 0, st, g);
    else throw invalid_error("moment pass: one or two dimensions");
    HIP_CHECK(hipGetLastError());
}

// fold: d mod 4 != 0 (norm in a free K slot); wmul: d mod 4 == 0 (norms as weights) - the two pruned plain fp64 shapes
void launch_sweep_grouped(const GSweepArgs& g, int dtype, int KS, hipStream_t st) {
    if (g.total_wg == 0) return;
    if (g.total_wg > 0x7fffffffll) throw invalid_error("grouped sweeps: grid too large");
    const dim3 grid((unsigned)g.total_wg), block(sweep_block_threads(true));
    if (use_f16x2(dtype)) {   // KS carries the number of 32-slot f16 MFMAs
        if (KS == 1 && g.w32) { ++g_w32_launches; hipLaunchKernelGGL((kde_sweep_f16_w32p_group_kernel<1>), grid, block, 0, st, g); }
        else if (KS == 1) hipLaunchKernelGGL((kde_sweep_f16_group_kernel<1>), grid, block, 0, st, g);
        else if (KS == 2) hipLaunchKernelGGL((kde_sweep_f16_group_kernel<2>), grid, block, 0, st, g);
        else throw invalid_error("grouped fp32 sweeps: at most 10 whitened dimensions");
        HIP_CHECK(hipGetLastError());
        return;
    }
    if (dtype != PBN_F64 || KS < 1 || KS > 2 || (g.fold != 0) == (g.wmul != 0)) throw invalid_error("grouped sweeps: fp64 / fp32 on the f16 cores, at most 8 whitened dimensions");
    constexpr int QGP = PBN_QG_PRUNE;
    if (g.fold) {
        if (g.moments && (KS != 1 || !g.group_masks)) throw invalid_error("grouped sweeps: the moment pass stands beside one- and two-variable units with per-group masks");
        if (KS == 1 && g.moments) hipLaunchKernelGGL((kde_sweep_group_kernel<double, 1, QGP, true, false, true>), grid, block, 0, st, g);
        else if (KS == 1) hipLaunchKernelGGL((kde_sweep_group_kernel<double, 1, QGP, true, false>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((kde_sweep_group_kernel<double, 2, QGP, true, false>), grid, block, 0, st, g);
    } else {
        if (KS == 1) hipLaunchKernelGGL((kde_sweep_group_kernel<double, 1, QGP, false, true>), grid, block, 0, st, g);
        else hipLaunchKernelGGL((kde_sweep_group_kernel<double, 2, QGP, false, true>), grid, block, 0, st, g);
    }
    HIP_CHECK(hipGetLastError());
}

void launch_finish(const FinishArgs& a, bool cond, double* dev_sum_out, hipStream_t st, double* dev_sum_marg_out) {
    const int64_t nblocks = ceil_div(a.nq, 256);
    if (nblocks == 0) return;
    dim3 grid((unsigned)nblocks), block(256);
    if (cond)
        hipLaunchKernelGGL(kde_finish_kernel<true>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(kde_finish_kernel<false>, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
    if (dev_sum_out) {
        hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const double*)a.block_sums, nblocks, dev_sum_out);
        HIP_CHECK(hipGetLastError());
    }
    if (dev_sum_marg_out && a.block_sums_marg) {
        hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(256), 0, st, (const double*)a.block_sums_marg, nblocks, dev_sum_marg_out);
        HIP_CHECK(hipGetLastError());
    }
}

}  // namespace pbn

extern "C" void pbn_debug_w32_launches(unsigned long long* n, int reset) {
    if (n) *n = pbn::g_w32_launches.load();
    if (reset) pbn::g_w32_launches.store(0);
}
extern "C" void pbn_debug_sweep_redo(unsigned long long* redo, unsigned long long* units, int reset) {
    unsigned long long z = 0;
    if (redo) (void)hipMemcpyFromSymbol(redo, HIP_SYMBOL(pbn::g_sweep_redo), sizeof z);
    if (units) (void)hipMemcpyFromSymbol(units, HIP_SYMBOL(pbn::g_sweep_units), sizeof z);
    if (reset) { (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_sweep_redo), &z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_sweep_units), &z, sizeof z); }
}
// measurement aid like the above: tiles visited / tiles offered to the waves of the pruned fp64 sweeps since the last reset
// measurement aid like the above: tiles visited / tiles offered to the waves of the pruned fp64 sweeps since the last reset
extern "C" void pbn_debug_moment_visits(unsigned long long* visits) {
    if (visits) (void)hipMemcpyFromSymbol(visits, HIP_SYMBOL(pbn::g_mom_visits), sizeof(unsigned long long));
}
extern "C" void pbn_debug_moment_left(unsigned long long* left, int reset) {
    unsigned long long z = 0;
    if (left) (void)hipMemcpyFromSymbol(left, HIP_SYMBOL(pbn::g_mom_left), sizeof z);
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_mom_left), &z, sizeof z);
}
extern "C" void pbn_debug_moment_totals(unsigned long long* pairs_d1, unsigned long long* pairs_d2, int reset) {
    unsigned long long v[2] = {0, 0}, z[2] = {0, 0};
    (void)hipMemcpyFromSymbol(v, HIP_SYMBOL(pbn::g_mom_taken), sizeof v);
    if (pairs_d1) *pairs_d1 = v[0];
    if (pairs_d2) *pairs_d2 = v[1];
    if (reset) (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_mom_taken), z, sizeof z);
}
extern "C" void pbn_debug_moment_pairs(unsigned long long* pairs, unsigned long long* batches, int reset) {
    unsigned long long z = 0;
    if (pairs) (void)hipMemcpyFromSymbol(pairs, HIP_SYMBOL(pbn::g_mom_pairs), sizeof z);
    if (batches) (void)hipMemcpyFromSymbol(batches, HIP_SYMBOL(pbn::g_mom_batches), sizeof z);
    if (reset) { (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_mom_pairs), &z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_mom_batches), &z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_mom_visits), &z, sizeof z); }
}
extern "C" void pbn_debug_sweep_visits(unsigned long long* visited, unsigned long long* tiles, int reset) {
    unsigned long long z = 0;
    if (visited) (void)hipMemcpyFromSymbol(visited, HIP_SYMBOL(pbn::g_sweep_visit), sizeof z);
    if (tiles) (void)hipMemcpyFromSymbol(tiles, HIP_SYMBOL(pbn::g_sweep_tiles), sizeof z);
    if (reset) { (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_sweep_visit), &z, sizeof z); (void)hipMemcpyToSymbol(HIP_SYMBOL(pbn::g_sweep_tiles), &z, sizeof z); }
}
