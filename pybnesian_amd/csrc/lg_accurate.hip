// Ill-conditioned LinearGaussianCPD fits: moments in double-double, solved in double-double.
//
// Reference: MLE<LinearGaussianCPD> with >= 3 parents runs Eigen's ColPivHouseholderQR on the N x (p+1) design matrix
// (/root/reference/pybnesian/learning/parameters/mle_LinearGaussianCPD.hpp:152-193), error ~ kappa * eps.  The score engine
// solves the normal equations on one-pass fp64 moments instead (scoring.hip, lg_fit): kappa^2 * eps.  That is invisible on
// ordinary tables and wrong by more than the 1e-6 parity bar when parents are nearly collinear (kappa >~ 1e5) or the fit is
// nearly exact (1 - R^2 <~ 1e-8).  lg_fit flags those cases (smallest pivot ratio of its diagonally pivoted Cholesky, RSS /
// SSE_y) and the callers come here: ONE more pass over the candidate's rows accumulates the RAW moments sum x_i,
// sum x_i x_j in double-double (TwoProd by FMA: every product exact, TwoSum accumulation: ~1e-32 relative), and the
// centring, the pivoted Cholesky, the solves and the residual sum of squares (a Schur complement) run in double-double on the
// host on (p+1)^2 numbers.  The result is the least-squares solution of the DATA to full double precision - at least as
// accurate as the reference's Householder QR - at the cost of a rare extra pass; no N x (p+1) copy, no QR kernel.
// Rank decision: a pivot below (eps (p+1))^2 of the largest one marks a dependent column (coefficient 0), the square of
// ColPivHouseholderQR's default threshold on |R_kk| / |R_00| (the reference pivots the uncentred matrix with its ones column;
// here the centred one: the decisions can only differ for kappa ~ 1e15).
#include <algorithm>
#include <cmath>
#include <numeric>
#include <vector>

#include "common.hpp"
#include "scoring_internal.hpp"

// The error-free transformations below (TwoSum, TwoProd) are exact only if every operation is rounded exactly as
// written: hipcc's default for device code, -ffp-contract=fast, would fuse the product of TwoProd into the first addition of
// the following TwoSum (s = fma(x, y, acc) instead of acc + round(x y)) and silently turn the double-double accumulation
// back into plain double.
#pragma STDC FP_CONTRACT OFF

namespace pbn {
namespace score {

namespace {

struct dd {
    double hi, lo;
};
__host__ __device__ inline dd two_sum(double a, double b) {
    const double s = a + b, bb = s - a;
    return {s, (a - (s - bb)) + (b - bb)};
}
__host__ __device__ inline dd quick_two_sum(double a, double b) {
    const double s = a + b;
    return {s, b - (s - a)};
}
__host__ __device__ inline dd two_prod(double a, double b) {
    const double p = a * b;
    return {p, __builtin_fma(a, b, -p)};
}
__host__ __device__ inline dd dd_add(dd a, dd b) {
    dd s = two_sum(a.hi, b.hi);
    const dd t = two_sum(a.lo, b.lo);
    s.lo += t.hi;
    s = quick_two_sum(s.hi, s.lo);
    s.lo += t.lo;
    return quick_two_sum(s.hi, s.lo);
}
__host__ __device__ inline dd dd_add_prod(dd acc, double a, double b) { return dd_add(acc, two_prod(a, b)); }
inline dd dd_neg(dd a) { return {-a.hi, -a.lo}; }
inline dd dd_sub(dd a, dd b) { return dd_add(a, dd_neg(b)); }
inline dd dd_mul(dd a, dd b) {
    dd p = two_prod(a.hi, b.hi);
    p.lo += a.hi * b.lo + a.lo * b.hi;
    return quick_two_sum(p.hi, p.lo);
}
inline dd dd_div(dd a, dd b) {
    const double q1 = a.hi / b.hi;
    dd r = dd_sub(a, dd_mul(b, {q1, 0.0}));
    const double q2 = r.hi / b.hi;
    r = dd_sub(r, dd_mul(b, {q2, 0.0}));
    const double q3 = r.hi / b.hi;
    dd q = quick_two_sum(q1, q2);
    return dd_add(q, {q3, 0.0});
}
inline dd dd_sqrt(dd a) {
    if (!(a.hi > 0.0)) return {0.0, 0.0};
    const double x = 1.0 / std::sqrt(a.hi), ax = a.hi * x;
    const dd diff = dd_sub(a, two_prod(ax, ax));
    return dd_add({ax, 0.0}, {diff.hi * (x * 0.5), 0.0});
}

constexpr int DD_MAX_D = 16;   // variable + up to 15 parents
constexpr int DD_BLOCKS = 128;

struct DdGramArgs {
    const void* base;
    int64_t ld;
    int cols[DD_MAX_D];
    int d;
    int64_t row0, n0, row1;   // logical row r -> r < n0 ? row0 + r : row1 + (r - n0), then through `rows` if given
    const int32_t* rows;
    int64_t n;
    double* partial;          // [tile][block][20][2]: 16 products + 4 sums, (hi, lo)
};

__device__ inline dd wave_sum_dd(dd v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        dd o;
        o.hi = __shfl_xor(v.hi, off);
        o.lo = __shfl_xor(v.lo, off);
        v = dd_add(v, o);
    }
    return v;
}

// grid (DD_BLOCKS, tiles): tile = (I, J), I <= J, of 4-column groups; 16 products x_{4I+i} x_{4J+j} + the sums of group J
template <typename T>
__global__ __launch_bounds__(256) void gram_dd_kernel(DdGramArgs a) {
    int I = 0, J = 0;
    {
        const int ng = (a.d + 3) / 4;
        int t = blockIdx.y;
        for (I = 0; I < ng; ++I) {
            if (t < ng - I) { J = I + t; break; }
            t -= ng - I;
        }
    }
    const T* ci[4];
    const T* cj[4];
    bool vi[4], vj[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        vi[k] = 4 * I + k < a.d;
        vj[k] = 4 * J + k < a.d;
        ci[k] = (const T*)a.base + (int64_t)a.cols[vi[k] ? 4 * I + k : 0] * a.ld;
        cj[k] = (const T*)a.base + (int64_t)a.cols[vj[k] ? 4 * J + k : 0] * a.ld;
    }
    dd acc[20];
#pragma unroll
    for (int k = 0; k < 20; ++k) acc[k] = {0.0, 0.0};
    for (int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x; r < a.n; r += (int64_t)gridDim.x * 256) {
        int64_t src = r < a.n0 ? a.row0 + r : a.row1 + (r - a.n0);
        if (a.rows) src = a.rows[src];
        double x[4], y[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            x[k] = vi[k] ? (double)ci[k][src] : 0.0;
            y[k] = vj[k] ? (double)cj[k][src] : 0.0;
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) acc[i * 4 + j] = dd_add_prod(acc[i * 4 + j], x[i], y[j]);
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[16 + j] = dd_add(acc[16 + j], {y[j], 0.0});
    }
    __shared__ double red[4][20][2];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 20; ++k) {
        const dd v = wave_sum_dd(acc[k]);
        if (lane == 0) { red[wave][k][0] = v.hi; red[wave][k][1] = v.lo; }
    }
    __syncthreads();
    if (threadIdx.x < 20) {
        const int k = threadIdx.x;
        dd v = {red[0][k][0], red[0][k][1]};
        for (int w = 1; w < 4; ++w) v = dd_add(v, {red[w][k][0], red[w][k][1]});
        double* o = a.partial + (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 20 + k) * 2;
        o[0] = v.hi;
        o[1] = v.lo;
    }
}

}  // namespace

// Raw moments of `d` columns over the described rows in double-double, then the exact least-squares fit.
double lg_fit_accurate(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n0, int64_t row1, int64_t n,
                       const int32_t* dev_rows, double* beta) {
    const int p = d - 1;
    if (d > DD_MAX_D) throw invalid_error("lg_fit_accurate: more than 15 parents");
    pbn_ctx* ctx = t->ctx;
    const int ng = (d + 3) / 4, tiles = ng * (ng + 1) / 2;
    const int nblocks = (int)std::max<int64_t>(1, std::min<int64_t>(DD_BLOCKS, ceil_div(n, 256)));
    const size_t count = (size_t)tiles * nblocks * 40;
    ctx->scratch_red.reserve(count);
    DdGramArgs a{};
    a.base = t->data; a.ld = t->ld; a.d = d;
    for (int i = 0; i < d; ++i) a.cols[i] = cols[i];
    a.row0 = row0; a.n0 = n0; a.row1 = row1; a.rows = dev_rows; a.n = n; a.partial = ctx->scratch_red.p;
    const dim3 grid((unsigned)nblocks, (unsigned)tiles), block(256);
    {
        KernelTimer kt(ctx, PBN_K_GRAM);
        if (t->dtype == PBN_F64) hipLaunchKernelGGL(gram_dd_kernel<double>, grid, block, 0, ctx->stream, a);
        else hipLaunchKernelGGL(gram_dd_kernel<float>, grid, block, 0, ctx->stream, a);
        HIP_CHECK(hipGetLastError());
    }
    std::vector<double> h(count);
    HIP_CHECK(hipMemcpyAsync(h.data(), ctx->scratch_red.p, count * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // raw moments, blocks added in order
    std::vector<dd> S(d, dd{0, 0}), G((size_t)d * d, dd{0, 0});
    int tile = 0;
    for (int I = 0; I < ng; ++I)
        for (int J = I; J < ng; ++J, ++tile) {
            dd acc[20];
            for (auto& v : acc) v = {0, 0};
            for (int b = 0; b < nblocks; ++b) {
                const double* o = h.data() + ((size_t)tile * nblocks + b) * 40;
                for (int k = 0; k < 20; ++k) acc[k] = dd_add(acc[k], {o[2 * k], o[2 * k + 1]});
            }
            for (int i = 0; i < 4; ++i)
                for (int j = 0; j < 4; ++j) {
                    const int r = 4 * I + i, c = 4 * J + j;
                    if (r < d && c < d) G[r + (size_t)c * d] = G[c + (size_t)r * d] = acc[i * 4 + j];
                }
            if (I == 0)   // the sums of group J ride with every tile (*, J); take them from (0, J)
                for (int j = 0; j < 4; ++j)
                    if (4 * J + j < d) S[4 * J + j] = acc[16 + j];
        }
    // centred SSE and means
    const dd N = {(double)n, 0.0};
    std::vector<dd> C((size_t)d * d), mu(d);
    for (int i = 0; i < d; ++i) mu[i] = dd_div(S[i], N);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) C[i + (size_t)j * d] = dd_sub(G[i + (size_t)j * d], dd_mul(S[i], mu[j]));
    // diagonally pivoted Cholesky of the parents' block (indices 1..p), rank by the squared QR threshold
    std::vector<int> piv(p);
    std::iota(piv.begin(), piv.end(), 0);
    std::vector<dd> L((size_t)p * p, dd{0, 0}), diag(p);
    for (int i = 0; i < p; ++i) diag[i] = C[(i + 1) + (size_t)(i + 1) * d];
    const double thr = std::pow(2.220446049250313e-16 * (double)std::min<int64_t>(n, d), 2.0);
    int rank = 0;
    double first = 0.0;
    for (int k = 0; k < p; ++k) {
        int best = k;
        for (int j = k + 1; j < p; ++j)
            if (diag[piv[j]].hi > diag[piv[best]].hi) best = j;
        std::swap(piv[k], piv[best]);
        // rows of L follow the pivot order: swap the already computed parts
        for (int c2 = 0; c2 < k; ++c2) std::swap(L[k + (size_t)c2 * p], L[best + (size_t)c2 * p]);
        const dd pk = diag[piv[k]];
        if (k == 0) first = pk.hi;
        if (!(pk.hi > thr * first) || !std::isfinite(pk.hi)) break;
        const dd lkk = dd_sqrt(pk);
        L[k + (size_t)k * p] = lkk;
        for (int i = k + 1; i < p; ++i) {
            dd v = C[(piv[i] + 1) + (size_t)(piv[k] + 1) * d];
            for (int c2 = 0; c2 < k; ++c2) v = dd_sub(v, dd_mul(L[i + (size_t)c2 * p], L[k + (size_t)c2 * p]));
            v = dd_div(v, lkk);
            L[i + (size_t)k * p] = v;
            diag[piv[i]] = dd_sub(diag[piv[i]], dd_mul(v, v));
        }
        rank = k + 1;
    }
    // forward / backward solves on the leading `rank` pivots; RSS = C_yy - |w|^2
    std::vector<dd> w(rank), z(rank);
    dd rss = C[0];
    for (int i = 0; i < rank; ++i) {
        dd v = C[0 + (size_t)(piv[i] + 1) * d];
        for (int c2 = 0; c2 < i; ++c2) v = dd_sub(v, dd_mul(L[i + (size_t)c2 * p], w[c2]));
        w[i] = dd_div(v, L[i + (size_t)i * p]);
        rss = dd_sub(rss, dd_mul(w[i], w[i]));
    }
    for (int i = rank - 1; i >= 0; --i) {
        dd v = w[i];
        for (int c2 = i + 1; c2 < rank; ++c2) v = dd_sub(v, dd_mul(L[c2 + (size_t)i * p], z[c2]));
        z[i] = dd_div(v, L[i + (size_t)i * p]);
    }
    std::vector<dd> b(p, dd{0, 0});
    for (int i = 0; i < rank; ++i) b[piv[i]] = z[i];
    dd b0 = mu[0];
    for (int j = 0; j < p; ++j) b0 = dd_sub(b0, dd_mul(b[j], mu[j + 1]));
    beta[0] = b0.hi;
    for (int j = 0; j < p; ++j) beta[j + 1] = b[j].hi;
    if (n <= p + 1) return INF;
    return std::max(rss.hi, 0.0) / ((double)n - p - 1);
}

}  // namespace score
}  // namespace pbn
