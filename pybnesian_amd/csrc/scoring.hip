// Score engine: BIC / BGe / CV-likelihood / hold-out likelihood local scores, batched.
//
// Reference (paths under /root/reference/pybnesian/): learning/scores/bic.cpp:12-27, bge.hpp:154-234,
// bge.cpp:106-144, cv_likelihood.cpp:11-25, holdout_likelihood.cpp:14-23, validated_likelihood.hpp:14-60,
// learning/parameters/mle_LinearGaussianCPD.hpp:11-221, factors/continuous/LinearGaussianCPD.cpp:92-149,
// dataset/crossvalidation_adaptator.hpp:15-58, dataset/holdout_adaptator.hpp:17-61.
//
// MI355X design (DESIGN.md "score engine"):
//  * The table is permuted ONCE on device into split order (libstdc++ std::shuffle of std::mt19937{seed},
//    exactly the reference's fold membership): CV fold f is the contiguous row range [limits[f],
//    limits[f+1]) and its training set the two ranges around it, so no per-call arrow::Take is needed.
//  * One f64-MFMA Gram pass per fold gives pilot-shifted moments (N, S, G) of every fold; moments are
//    additive, so the statistics of "all rows but fold f" are a subtraction.  Every LinearGaussian quantity
//    (MLE, BIC, BGe, and the Gaussian test log-likelihood of a fold, which is a quadratic form in the test
//    fold's moments) is then O(p^3) host arithmetic on <= (p+1)^2 numbers: the reference's O(N p^2) QR per
//    candidate disappears.  (Normal equations instead of column-pivoted QR: see DESIGN.md "numerics".)
//  * CKDE candidates: bandwidth from the training-fold moments (no extra pass), then per fold
//    pack(train) -> pack(test) -> fused joint+marginal sweep -> finish, all enqueued on the stream from
//    scratch arenas; one synchronisation and one D2H of all fold sums per batch.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <map>
#include <numeric>
#include <random>

#include "common.hpp"
#include "hostmath.hpp"
#include "kde_group.hpp"
#include "kde_model.hpp"
#include "scoring_internal.hpp"
#include "stats_kernels.hpp"

using namespace pbn;
using namespace pbn::score;

namespace pbn {
namespace score {

// Shifted Gram of up to 64 columns over a contiguous row range -> raw S (d) and G (d*d col-major).
void gram_raw(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const int32_t* dev_rows,
              const double* shift_dev, double* S, double* G) {
    pbn_ctx* ctx = t->ctx;
    const int nct = (d + 15) / 16;
    const int WS = gram_ws(nct);
    int nblocks = (int)std::min<int64_t>(4 * ctx->num_cus, std::max<int64_t>(1, ceil_div(n, 256)));
    int64_t rpb = ceil_div(std::max<int64_t>(n, 1), nblocks);
    rpb = (rpb + 63) / 64 * 64;
    nblocks = (int)std::max<int64_t>(1, ceil_div(n, rpb));
    ctx->scratch_red.reserve((size_t)nblocks * WS + WS);
    double* partial = ctx->scratch_red.p;
    double* total = partial + (size_t)nblocks * WS;
    GramArgs a{};
    a.base = t->data; a.ld = t->ld; a.n_cols = d; a.row0 = row0; a.rows = dev_rows; a.n = n;
    for (int i = 0; i < d; ++i) a.gc.cols[i] = cols[i];
    a.rows_per_block = rpb; a.shift = shift_dev; a.partial = partial; a.num_cus = ctx->num_cus;
    { KernelTimer kt(ctx, PBN_K_GRAM); launch_gram(a, t->dtype, nblocks, total, ctx->stream); }
    std::vector<double> h((size_t)WS);
    HIP_CHECK(hipMemcpyAsync(h.data(), total, (size_t)WS * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double* Sx = h.data() + (WS - nct * 16);
    for (int i = 0; i < d; ++i) S[i] = Sx[i];
    int p = 0;
    for (int I = 0; I < nct; ++I)
        for (int J = I; J < nct; ++J, ++p) {
            const double* tile = h.data() + (size_t)p * 256;
            for (int e = 0; e < 256; ++e) {
                const int reg = e >> 6, lane = e & 63;
                const int r = I * 16 + (lane >> 4) + 4 * reg, c = J * 16 + (lane & 15);
                if (r >= d || c >= d) continue;
                if (I == J && r > c) continue;  // keep the upper triangle of diagonal tiles
                G[r + (size_t)c * d] = tile[e];
                G[c + (size_t)r * d] = tile[e];
            }
        }
}

// Moments of all n columns over each of the row ranges [r0[i], r1[i]), i in [s0, s1): ONE segmented Gram launch per block (pair)
// of <= 64 columns.  A range is cut into pieces of SEG_ROWS rows (one workgroup each) that are added in order: the result for a
// range depends on its rows only - not on the other ranges of the launch, not on who else computes what.
// PBN_MOMENT_SEG_ROWS (default 4096, a multiple of 128; 2M x 64 doubles: 231 / 212 / 196 / 203 us at 1024 / 2048 / 4096 / 8192 - pieces on
// 512 resident workgroup slots): the piece size is part of the summation order - like PBN_MOMENT_SUPERBLOCKS it must
// be the same on every rank of a job, and another value gives other last bits.
static int64_t seg_rows() {
    static const int64_t v = [] {
        const int64_t r = PBN_TUNE(MOMENT_SEG_ROWS, 4096);
        return std::max<int64_t>(128, r / 128 * 128);
    }();
    return v;
}
void compute_stats_segments(const pbn_scoredata* sd, const std::vector<int64_t>& r0, const std::vector<int64_t>& r1, size_t s0, size_t s1,
                            std::vector<Stats>& out) {
    const int n = sd->n;
    const int64_t SEG_ROWS = seg_rows();
    for (size_t i = s0; i < s1; ++i) { out[i].zero(n); out[i].N = r1[i] - r0[i]; }
    if (s1 <= s0) return;
    const pbn_table* t = sd->table();
    pbn_ctx* ctx = t->ctx;
    const int nseg = (int)(s1 - s0);
    std::vector<int32_t> blk, off(nseg + 1, 0);
    for (int g = 0; g < nseg; ++g) {
        for (int64_t r = r0[s0 + g]; r < r1[s0 + g]; r += SEG_ROWS) {
            const int32_t slot = (int32_t)(blk.size() / 4);   // the piece's partial: pieces of a segment are consecutive slots
            blk.push_back(g); blk.push_back((int32_t)r); blk.push_back((int32_t)std::min(r + SEG_ROWS, r1[s0 + g])); blk.push_back(slot);
        }
        off[g + 1] = (int32_t)(blk.size() / 4);
    }
    const int nblk = (int)(blk.size() / 4);
    if (nblk == 0) return;
    if (t->n_rows > 0x7fffffffll) throw invalid_error("score data: more than 2^31 rows");
    dev_buf<int32_t> dblk(blk.size() + off.size());
    HIP_CHECK(hipMemcpyAsync(dblk.p, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(dblk.p + blk.size(), off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    auto run = [&](const std::vector<int>& cols) {
        const int d = (int)cols.size(), nct = (d + 15) / 16, WS = gram_ws(nct);
        ctx->scratch_red.reserve(((size_t)nblk + (size_t)nseg) * WS);
        double* partial = ctx->scratch_red.p;
        double* outd = partial + (size_t)nblk * WS;
        GramArgs a{};
        a.base = t->data; a.ld = t->ld; a.n_cols = d; a.row0 = 0; a.rows = nullptr; a.n = t->n_rows;
        for (int i = 0; i < d; ++i) a.gc.cols[i] = cols[i];
        a.rows_per_block = SEG_ROWS; a.blk = dblk.p; a.shift = sd->shift_dev.p; a.partial = partial; a.num_cus = ctx->num_cus;
        { KernelTimer kt(ctx, PBN_K_GRAM); launch_gram_segments(a, t->dtype, nblk, dblk.p + blk.size(), nseg, outd, ctx->stream); }
        std::vector<double> h((size_t)nseg * WS);
        HIP_CHECK(hipMemcpyAsync(h.data(), outd, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int g = 0; g < nseg; ++g) {
            const double* w = h.data() + (size_t)g * WS;
            Stats& st = out[s0 + g];
            for (int i = 0; i < d; ++i) st.S[cols[i]] = w[WS - nct * 16 + i];
            int p = 0;
            for (int I = 0; I < nct; ++I)
                for (int J = I; J < nct; ++J, ++p) {
                    const double* tile = w + (size_t)p * 256;
                    for (int e = 0; e < 256; ++e) {
                        const int reg = e >> 6, lane = e & 63;
                        const int r = I * 16 + (lane >> 4) + 4 * reg, c = J * 16 + (lane & 15);
                        if (r >= d || c >= d || (I == J && r > c)) continue;
                        st.G[cols[r] + (size_t)cols[c] * n] = tile[e];
                        st.G[cols[c] + (size_t)cols[r] * n] = tile[e];
                    }
                }
        }
    };
    if (n <= 64) {
        std::vector<int> cols(n);
        std::iota(cols.begin(), cols.end(), 0);
        run(cols);
    } else {
        const int nb = (n + 31) / 32;
        for (int bi = 0; bi < nb; ++bi)
            for (int bj = bi + 1; bj < nb; ++bj) {
                std::vector<int> cols;
                for (int c = bi * 32; c < std::min(n, bi * 32 + 32); ++c) cols.push_back(c);
                for (int c = bj * 32; c < std::min(n, bj * 32 + 32); ++c) cols.push_back(c);
                run(cols);
            }
    }
    HIP_CHECK(hipStreamSynchronize(ctx->stream));   // dblk goes out of scope
}

// means / centred SSE of a column subset from moments.
void subset_moments(const pbn_scoredata* sd, const Stats& st, const int* cols, int d, double* mu, double* sse) {
    const int n = sd->n;
    const double N = (double)st.N;
    for (int i = 0; i < d; ++i) mu[i] = sd->shift[cols[i]] + (st.N > 0 ? st.S[cols[i]] / N : 0.0);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i)
            sse[i + (size_t)j * d] =
                st.G[cols[i] + (size_t)cols[j] * n] - (st.N > 0 ? st.S[cols[i]] * st.S[cols[j]] / N : 0.0);
}

void stats_minus(const Stats& a, const Stats& b, Stats& out) {
    out.N = a.N - b.N;
    out.S.resize(a.S.size());
    out.G.resize(a.G.size());
    for (size_t i = 0; i < a.S.size(); ++i) out.S[i] = a.S[i] - b.S[i];
    for (size_t i = 0; i < a.G.size(); ++i) out.G[i] = a.G[i] - b.G[i];
}

// ---- MLE<LinearGaussianCPD> from (N, means, SSE) of [y, x1..xp] (mle_LinearGaussianCPD.hpp:11-221) -----
// beta: p+1 (intercept first); returns the unbiased variance (inf when N <= p+1).
double lg_fit(int64_t N, int p, const double* mu, const double* S /* (p+1)^2 col-major */, double* beta, bool* suspect) {
    if (suspect) *suspect = false;
    const int d = p + 1;
    auto s = [&](int i, int j) { return S[i + (size_t)j * d]; };
    const double rows = (double)N;
    if (p == 0) {
        beta[0] = mu[0];
        if (N == 1) return INF;
        return s(0, 0) / (rows - 1);
    }
    if (p == 1) {
        const double var_x = s(1, 1) / (rows - 1);
        if (var_x < MACHINE_TOL) {
            beta[0] = mu[0]; beta[1] = 0;
            return N <= 2 ? INF : s(0, 0) / (rows - 2);
        }
        const double b = s(0, 1) / s(1, 1);
        beta[0] = mu[0] - b * mu[1]; beta[1] = b;
        if (N <= 2) return INF;
        const double rss = s(0, 0) - 2 * b * s(0, 1) + b * b * s(1, 1);
        return std::max(rss, 0.0) / (rows - 2);
    }
    if (p == 2) {
        const double v1 = s(1, 1) / (rows - 1), v2 = s(2, 2) / (rows - 1), c12 = s(1, 2) / (rows - 1);
        const bool singular1 = v1 < MACHINE_TOL;
        const bool singular2 = v2 < MACHINE_TOL || std::fabs(c12 / std::sqrt(v1 * v2)) > (1 - MACHINE_TOL);
        double b1 = 0, b2 = 0;
        if (singular1) {
            if (!singular2) b2 = s(0, 2) / s(2, 2);
        } else if (singular2) {
            b1 = s(0, 1) / s(1, 1);
        } else {
            const double cy1 = s(0, 1) / (rows - 1), cy2 = s(0, 2) / (rows - 1);
            const double den = v1 * v2 - c12 * c12;
            b1 = (v2 * cy1 - c12 * cy2) / den;
            b2 = (cy2 - b1 * c12) / v2;
        }
        beta[0] = mu[0] - b1 * mu[1] - b2 * mu[2]; beta[1] = b1; beta[2] = b2;
        if (N <= 3) return INF;
        const double rss = s(0, 0) - 2 * b1 * s(0, 1) - 2 * b2 * s(0, 2) + b1 * b1 * s(1, 1) + 2 * b1 * b2 * s(1, 2) +
                           b2 * b2 * s(2, 2);
        return std::max(rss, 0.0) / (rows - 3);
    }
    // p >= 3: normal equations on the centred moments by a diagonally pivoted Cholesky - the pivot order of
    // ColPivHouseholderQR (largest remaining column norm first, mle_LinearGaussianCPD.hpp:171) - a pivot that is not
    // positive relative to its column's scale marks a dependent column (coefficient 0, like the rank-deficient QR solve).
    // RSS is the Schur complement S_yy - |L^-1 b|^2 (error ~ eps S_yy, whatever the size of the coefficients), not the
    // quadratic form in beta.  What this cannot deliver is reported through `suspect`: nearly collinear parents (smallest
    // pivot ratio below 1e-6: coefficients good to ~1e-9 at best) or a nearly exact fit (RSS < 1e-8 S_yy): the callers
    // then refit from the rows in double-double (lg_accurate.hip).
    std::vector<double> A((size_t)p * p), L((size_t)p * p, 0.0), w(p, 0.0), z(p, 0.0), diag(p), scale(p);
    std::vector<int> piv(p);
    for (int j = 0; j < p; ++j)
        for (int i = 0; i < p; ++i) A[i + (size_t)j * p] = s(i + 1, j + 1);
    std::iota(piv.begin(), piv.end(), 0);
    for (int i = 0; i < p; ++i) diag[i] = scale[i] = A[i + (size_t)i * p];
    int rank = 0;
    double min_ratio = 1.0;
    for (int k = 0; k < p; ++k) {
        int best = k;
        for (int j = k + 1; j < p; ++j)
            if (diag[piv[j]] > diag[piv[best]]) best = j;
        std::swap(piv[k], piv[best]);
        for (int c2 = 0; c2 < k; ++c2) std::swap(L[k + (size_t)c2 * p], L[best + (size_t)c2 * p]);
        const double pk = diag[piv[k]];
        if (!(pk > scale[piv[k]] * p * 2.220446049250313e-16) || !std::isfinite(pk)) break;  // the rest depends on the columns taken
        min_ratio = std::min(min_ratio, pk / scale[piv[k]]);
        const double lkk = std::sqrt(pk);
        L[k + (size_t)k * p] = lkk;
        for (int i = k + 1; i < p; ++i) {
            double t = A[piv[i] + (size_t)piv[k] * p];
            for (int c2 = 0; c2 < k; ++c2) t -= L[i + (size_t)c2 * p] * L[k + (size_t)c2 * p];
            t /= lkk;
            L[i + (size_t)k * p] = t;
            diag[piv[i]] -= t * t;
        }
        rank = k + 1;
    }
    double rss = s(0, 0);
    for (int i = 0; i < rank; ++i) {  // forward: w = L^-1 b
        double t = s(0, piv[i] + 1);
        for (int c2 = 0; c2 < i; ++c2) t -= L[i + (size_t)c2 * p] * w[c2];
        w[i] = t / L[i + (size_t)i * p];
        rss -= w[i] * w[i];
    }
    for (int i = rank - 1; i >= 0; --i) {  // backward
        double t = w[i];
        for (int c2 = i + 1; c2 < rank; ++c2) t -= L[c2 + (size_t)i * p] * z[c2];
        z[i] = t / L[i + (size_t)i * p];
    }
    for (int j = 0; j < p; ++j) beta[j + 1] = 0.0;
    for (int i = 0; i < rank; ++i) beta[piv[i] + 1] = z[i];
    double b0 = mu[0];
    for (int j = 0; j < p; ++j) b0 -= beta[j + 1] * mu[j + 1];
    beta[0] = b0;
    if (suspect) *suspect = rank < p || min_ratio < 1e-6 || !(rss > 1e-8 * s(0, 0));
    if (N <= p + 1) return INF;
    return std::max(rss, 0.0) / (rows - p - 1);
}

// BIC of a LinearGaussianCPD (bic.cpp:12-27)
double bic_lg(int64_t N, int p, double variance) {
    if (variance < MACHINE_TOL || std::isinf(variance)) return -INF;
    const double rows = (double)N;
    const double loglik = 0.5 * (1 + (double)p - rows) - 0.5 * rows * LOG_2PI - rows * 0.5 * std::log(variance);
    return loglik - std::log(rows) * 0.5 * (p + 2);
}

// Gaussian log-likelihood of the rows behind `test` moments under (beta, variance)
// (LinearGaussianCPD.cpp:92-149 summed): -1/2 n log(2 pi var) - RSS / (2 var), RSS as a quadratic form.
double lg_slogl_from_moments(const pbn_scoredata* sd, const Stats& test, const int* cols, int p, const double* beta,
                             double variance) {
    const int n = sd->n;
    const double Nt = (double)test.N;
    // residual r = (y - s_y) - sum_j beta_j (x_j - s_j) - c,  c = beta0 - s_y + sum_j beta_j s_j
    double c = beta[0] - sd->shift[cols[0]];
    for (int j = 1; j <= p; ++j) c += beta[j] * sd->shift[cols[j]];
    auto G = [&](int i, int j) { return test.G[cols[i] + (size_t)cols[j] * n]; };
    double rss = G(0, 0);
    double lin = test.S[cols[0]];
    for (int j = 1; j <= p; ++j) {
        rss -= 2 * beta[j] * G(0, j);
        lin -= beta[j] * test.S[cols[j]];
    }
    for (int i = 1; i <= p; ++i)
        for (int j = 1; j <= p; ++j) rss += beta[i] * beta[j] * G(i, j);
    rss += -2 * c * lin + Nt * c * c;
    rss = std::max(rss, 0.0);
    return -0.5 * Nt * (std::log(variance) + LOG_2PI) - 0.5 * rss / variance;
}

// The same sum from the ROWS (one streaming pass, lg_logl_kernel): for fits whose coefficients are huge (nearly collinear
// parents) the quadratic form above cancels - eps beta^2 G against a residual sum of squares that is tiny beside it.
double lg_slogl_from_rows(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const int32_t* dev_rows,
                          const double* beta, double variance) {
    if (n == 0) return 0.0;
    pbn_ctx* ctx = t->ctx;
    const int64_t nblocks = ceil_div(n, 256);
    ctx->scratch_misc.reserve((size_t)(nblocks + 1) * sizeof(double));
    double* bs = (double*)ctx->scratch_misc.p;
    LgArgs a{};
    a.base = t->data; a.ld = t->ld; a.p = d - 1; a.row0 = row0; a.n = n; a.rows = dev_rows;
    for (int i = 0; i < d; ++i) { a.gc.cols[i] = cols[i]; a.beta[i] = beta[i]; }
    a.inv_std = 1.0 / std::sqrt(variance);
    a.cte = -0.5 * std::log(variance) - 0.5 * LOG_2PI;
    a.logl = nullptr; a.block_sums = bs; a.want_cdf = 0;
    launch_lg_logl(a, t->dtype, ctx->stream);
    std::vector<double> hb((size_t)nblocks);
    HIP_CHECK(hipMemcpyAsync(hb.data(), bs, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    double s = 0.0;
    for (double v : hb) s += v;  // fixed order
    return s;
}

// BGe (bge.hpp:154-234).  params: iss_mu, iss_w, total_nodes, then optionally nu[n] (per table column).
double bge_score(const pbn_scoredata* sd, const Stats& st, const int* cols, int p, double iss_mu, double iss_w,
                 int total_nodes, const double* nu_all) {
    const int d = p + 1;
    std::vector<double> mu(d), sse((size_t)d * d);
    subset_moments(sd, st, cols, d, mu.data(), sse.data());
    const double N = (double)st.N;
    const double t = iss_mu * (iss_w - total_nodes - 1) / (iss_mu + 1);
    double logprob = 0.5 * (std::log(iss_mu) - std::log(N + iss_mu));
    logprob += std::lgamma(0.5 * (N + iss_w - total_nodes + p + 1)) - std::lgamma(0.5 * (iss_w - total_nodes + p + 1));
    logprob -= 0.5 * N * LOG_PI;
    std::vector<double> diff(d);
    for (int i = 0; i < d; ++i) diff[i] = nu_all ? mu[i] - nu_all[cols[i]] : 0.0;
    const double cte_r = (N * iss_mu) / (N + iss_mu);
    if (p == 0) {
        logprob += 0.5 * (iss_w - total_nodes + 1) * std::log(t);
        const double r = t + sse[0] + cte_r * diff[0] * diff[0];
        logprob -= 0.5 * (N + iss_w - total_nodes + 1) * std::log(r);
        return logprob;
    }
    logprob += 0.5 * (iss_w - total_nodes + 2 * p + 1) * std::log(t);
    std::vector<double> R((size_t)d * d), Rp((size_t)p * p);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) R[i + (size_t)j * d] = sse[i + (size_t)j * d] + (i == j ? t : 0.0) + cte_r * diff[i] * diff[j];
    for (int j = 0; j < p; ++j)
        for (int i = 0; i < p; ++i) Rp[i + (size_t)j * p] = R[(i + 1) + (size_t)(j + 1) * d];
    logprob -= 0.5 * (N + iss_w - total_nodes + p + 1) * std::log(hm::determinant(R.data(), d));
    logprob += 0.5 * (N + iss_w - total_nodes + p) * std::log(hm::determinant(Rp.data(), p));
    return logprob;
}

}  // namespace score
}  // namespace pbn

namespace pbn { namespace score {
bool lg_guard_on() {
    return knob_int("PBN_LG_GUARD", 1) != 0;   // read per call: the tests switch it
}
} }

static bool score_memo_on() {
    static const bool v = PBN_TUNE(SCORE_MEMO, 1) != 0;
    return v;
}

extern "C" {

// region moments = their segments added in segment order (see scoredata_create_impl)
static void regions_from_segments(pbn_scoredata* sd) {
    const int n = sd->n;
    if (sd->k > 0) { sd->fold.resize(sd->k); for (auto& f : sd->fold) f.zero(n); } else sd->all.zero(n);
    if (sd->n_hold > 0) sd->hold.zero(n);
    const int hold_region = sd->k > 0 ? sd->k : 1;
    for (size_t i = 0; i < sd->seg.size(); ++i) {
        const int r = sd->seg_region[i];
        Stats& dst = (sd->n_hold > 0 && r == hold_region) ? sd->hold : (sd->k > 0 ? sd->fold[r] : sd->all);
        dst.add(sd->seg[i]);
    }
    if (sd->k > 0) {
        sd->all.zero(n);
        for (int f = 0; f < sd->k; ++f) {
            sd->fold[f].N = sd->limits[f + 1] - sd->limits[f];
            sd->all.add(sd->fold[f]);
        }
    } else {
        sd->all.N = sd->n_cv;
    }
    if (sd->n_hold > 0) sd->hold.N = sd->n_hold;
}

// Row layout of the splits: HoldOut (dataset/holdout_adaptator.hpp:24-61), CrossValidation
// (dataset/crossvalidation_adaptator.hpp:17-57) and, for PBN_SPLIT_VALIDATED, the CV of the hold-out training part
// with the same seed (validated_likelihood.hpp:19-20).  Host only.
struct SplitLayout {
    std::vector<int32_t> idx, limits;
    int64_t n_cv = 0, n_hold = 0;
    int k = 0;
};
static SplitLayout split_layout(int64_t rows, int split, int k, uint32_t seed, double test_ratio) {
    SplitLayout L;
    SplitLayout* sd = &L;
    std::vector<int32_t> idx((size_t)rows);
    std::iota(idx.begin(), idx.end(), 0);
    sd->n_cv = rows;
    if (split == PBN_SPLIT_HOLDOUT || split == PBN_SPLIT_VALIDATED) {
        // holdout_adaptator.hpp:24-61
        if (test_ratio <= 0 || test_ratio >= 1.0) throw invalid_error("test_ratio must be a number between 0 and 1.");
        std::mt19937 rng{seed};
        std::shuffle(idx.begin(), idx.end(), rng);
        const int64_t test_rows = (int64_t)std::round((double)rows * test_ratio);
        const int64_t train_rows = rows - test_rows;
        if (test_rows == 0 || train_rows == 0)
            throw invalid_error("Wrong test_ratio (" + std::to_string(test_ratio) + "selected for HoldOut.\nGenerated train instances: " +
                                std::to_string(train_rows) + "\nGenerated test instances: " + std::to_string(test_rows));
        sd->n_cv = train_rows;
        sd->n_hold = test_rows;
    }
    if (split == PBN_SPLIT_CV || split == PBN_SPLIT_VALIDATED) {
        // crossvalidation_adaptator.hpp:17-57 on the (hold-out) training part; for VALIDATED the CV object is
        // built on training_data() with the same seed (validated_likelihood.hpp:19-20)
        const int64_t n = sd->n_cv;
        if (k <= 1 || k > n)
            throw invalid_error("Cannot split " + std::to_string(n) + " instances into " + std::to_string(k) + " folds.");
        std::vector<int32_t> local((size_t)n);
        std::iota(local.begin(), local.end(), 0);
        std::mt19937 rng{seed};
        std::shuffle(local.begin(), local.end(), rng);
        std::vector<int32_t> tmp(idx.begin(), idx.begin() + n);
        for (int64_t i = 0; i < n; ++i) idx[i] = tmp[local[i]];
        const int fold_size = (int)(n / k), extra = (int)(n % k);
        sd->k = k;
        sd->limits.assign(1, 0);
        int cur = 0;
        for (int i = 0; i < extra; ++i) { cur += fold_size + 1; sd->limits.push_back(cur); }
        for (int i = extra; i < k; ++i) { cur += fold_size; sd->limits.push_back(cur); }
    }
    L.idx = std::move(idx);
    return L;
}

static int scoredata_create_impl(pbn_ctx* ctx, const pbn_table* table, int split, int k, uint32_t seed, double test_ratio,
                                 int rank, int world, pbn_scoredata** out) {
    return guarded(mu_of(ctx), [&] {
        if (!ctx || !table || !out) throw invalid_error("pbn_scoredata_create: null argument");
        if (world < 1 || rank < 0 || rank >= world) throw invalid_error("pbn_scoredata_create_sharded: rank / world out of range");
        if (table->n_cols <= 0) throw invalid_error("pbn_scoredata_create: table has no columns");
        HIP_CHECK(hipSetDevice(ctx->device));
        auto sd = std::make_unique<pbn_scoredata>();
        sd->ctx = ctx; sd->dtype = table->dtype; sd->n = table->n_cols; sd->split = split; sd->src = table;
        const int64_t rows = table->n_rows;
        SplitLayout lay = split_layout(rows, split, k, seed, test_ratio);
        sd->n_cv = lay.n_cv; sd->n_hold = lay.n_hold; sd->k = lay.k; sd->limits = lay.limits;
        sd->perm = std::move(lay.idx);   // (2M rows: 8 MB - moved, not copied twice)
        const std::vector<int32_t>& idx = sd->perm;
        if (split != PBN_SPLIT_NONE) {
            pbn_table* pt = nullptr;
            int rc = pbn_table_take(table, idx.data(), rows, &pt);
            if (rc != PBN_OK) throw device_error(pbn_last_error());
            sd->perm_table = pt;
        }
        // pilot shifts over the CV region of the (permuted) table
        const pbn_table* t = sd->table();
        sd->shift_dev.alloc((size_t)sd->n);
        for (int c0 = 0; c0 < sd->n; c0 += 64) {
            GramCols gc{};
            const int d = std::min(64, sd->n - c0);
            for (int i = 0; i < d; ++i) gc.cols[i] = c0 + i;
            launch_pilot(t->data, t->ld, gc, d, 0, nullptr, sd->n_cv, t->dtype, sd->shift_dev.p, ctx->stream);
        }
        sd->shift.resize(sd->n);
        HIP_CHECK(hipMemcpyAsync(sd->shift.data(), sd->shift_dev.p, sd->n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        // moments: every region in PBN_MOMENT_SUPERBLOCKS super-blocks (boundaries: multiples of 64 rows, a function of the
        // region's length only); rank r of `world` computes segments [S r / world, S (r + 1) / world) and leaves the others zero;
        // the caller adds the ranks' buffers (exact: every segment is non-zero on one rank only) and installs them
        // (pbn_scoredata_moments), and regions_from_segments adds a region's segments in segment order: the totals are the same
        // bit for bit for every world size.
        // (16 super-blocks up to 128 columns; wider tables fewer - a segment is n + n^2 doubles on every rank, 176 of them at 10 folds +
        //  hold-out: 350 MB per rank at n = 500 - down to 2 from 512 columns: the count is part of the summation order, so it is a
        //  function of n alone)
        static const int SB_env = std::min(1024, std::max(0, PBN_TUNE(MOMENT_SUPERBLOCKS, 0)));
        const int SB = SB_env > 0 ? SB_env : (sd->n <= 128 ? 16 : (sd->n <= 256 ? 8 : (sd->n <= 512 ? 4 : 2)));
        auto add_region = [&](int region, int64_t r0, int64_t len) {
            int64_t prev = 0;
            for (int i = 1; i <= SB; ++i) {
                int64_t b = i == SB ? len : (len * i / SB + 63) / 64 * 64;
                b = std::min(b, len);
                sd->seg_region.push_back(region);
                sd->seg_r0.push_back(r0 + prev);
                sd->seg_r1.push_back(r0 + std::max(prev, b));
                prev = std::max(prev, b);
            }
        };
        int region = 0;
        if (sd->k > 0) for (int f = 0; f < sd->k; ++f) add_region(region++, sd->limits[f], sd->limits[f + 1] - sd->limits[f]);
        else add_region(region++, 0, sd->n_cv);
        if (sd->n_hold > 0) add_region(region++, sd->n_cv, sd->n_hold);
        const size_t S = sd->seg_region.size();
        sd->seg.resize(S);
        for (auto& st : sd->seg) st.zero(sd->n);
        compute_stats_segments(sd.get(), sd->seg_r0, sd->seg_r1, S * (size_t)rank / (size_t)world, S * (size_t)(rank + 1) / (size_t)world, sd->seg);
        regions_from_segments(sd.get());
        sd->partial = world > 1;
        *out = sd.release();
    });
}

int pbn_scoredata_create(pbn_ctx* ctx, const pbn_table* table, int split, int k, uint32_t seed, double test_ratio,
                         pbn_scoredata** out) {
    return scoredata_create_impl(ctx, table, split, k, seed, test_ratio, 0, 1, out);
}

int pbn_scoredata_create_sharded(pbn_ctx* ctx, const pbn_table* table, int split, int k, uint32_t seed, double test_ratio,
                                 int rank, int world, pbn_scoredata** out) {
    return scoredata_create_impl(ctx, table, split, k, seed, test_ratio, rank, world, out);
}

// Serialised moments: for each SEGMENT (scoredata_create_impl: the regions' super-blocks, in order) S[n] then G[n*n].
// set == 0 copies them out (segments this rank did not compute are zero), set != 0 installs all of them, rebuilds the
// regions' totals in segment order and clears the partial flag.
int pbn_scoredata_moments(pbn_scoredata* sd, double* buf, int64_t* len, int set) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_moments: null argument");
        const size_t per = (size_t)sd->n + (size_t)sd->n * sd->n;
        if (len) *len = (int64_t)(per * sd->seg.size());
        if (!buf) return;
        double* p = buf;
        for (Stats& st : sd->seg) {
            if (set) {
                std::memcpy(st.S.data(), p, sd->n * sizeof(double));
                std::memcpy(st.G.data(), p + sd->n, (size_t)sd->n * sd->n * sizeof(double));
            } else {
                std::memcpy(p, st.S.data(), sd->n * sizeof(double));
                std::memcpy(p + sd->n, st.G.data(), (size_t)sd->n * sd->n * sizeof(double));
            }
            p += per;
        }
        if (!set) return;
        regions_from_segments(sd);
        sd->partial = false;
        sd->kde_cache.clear();
        sd->term_total.clear();
        sd->score_memo.clear();
    });
}

void pbn_scoredata_destroy(pbn_scoredata* sd) {
    if (!sd) return;
    pbn::ctx_pin pin_(sd->ctx);
    std::lock_guard<std::recursive_mutex> lock_(mu_of(sd));
    if (sd->perm_table) pbn_table_destroy(sd->perm_table);
    delete sd;
}

// Attach the dictionary-encoded (discrete) columns: codes[j][r] = dictionary index of SOURCE row r.
int pbn_scoredata_set_discrete(pbn_scoredata* sd, int n_disc, const int32_t* const* codes, const int* cardinality) {
    return guarded(mu_of(sd), [&] {
        if (!sd || (n_disc > 0 && (!codes || !cardinality))) throw invalid_error("pbn_scoredata_set_discrete: null argument");
        const int64_t rows = (int64_t)sd->perm.size();
        // everything derived from the previous codes (device row lists of the groupings, per-configuration KDE sums, memoised
        // local scores) is stale: let the work in flight finish, then drop it
        HIP_CHECK(hipStreamSynchronize(sd->ctx->stream));
        sd->ctx->sync_lanes(pbn_ctx::MAX_PARKED);
        sd->groupings.clear();
        sd->kde_cache.clear();
        sd->term_total.clear();
        sd->score_memo.clear();
        sd->n_disc = n_disc;
        sd->codes.assign(n_disc, std::vector<int32_t>((size_t)rows));
        sd->card.assign(cardinality, cardinality + n_disc);
        for (int j = 0; j < n_disc; ++j)
            for (int64_t r = 0; r < rows; ++r) {
                const int32_t v = codes[j][sd->perm[r]];
                if (v < 0 || v >= cardinality[j]) throw invalid_error("pbn_scoredata_set_discrete: code out of range");
                sd->codes[j][r] = v;
            }
    });
}

// Validity of the continuous columns for BIC / BGe on tables with nulls (bic.cpp:12-27 uses
// valid_rows(variable, parents); bge.hpp:52-72 drops its cache when any column has nulls): masks[c] is a byte
// array (1 = valid) in source row order, or NULL when column c has no nulls.  Likelihood scores never see nulls:
// CrossValidation / HoldOut keep only rows valid in every column (crossvalidation_adaptator.hpp:24-37).
int pbn_scoredata_set_validity(pbn_scoredata* sd, const uint8_t* const* masks) {
    return guarded(mu_of(sd), [&] {
        if (!sd || !masks) throw invalid_error("pbn_scoredata_set_validity: null argument");
        if (sd->split != PBN_SPLIT_NONE) throw invalid_error("pbn_scoredata_set_validity: only for BIC / BGe score data");
        const int64_t rows = (int64_t)sd->perm.size();
        sd->valid.assign(sd->n, {});
        sd->has_nulls = false;
        for (int c = 0; c < sd->n; ++c)
            if (masks[c]) {
                sd->valid[c].assign(masks[c], masks[c] + rows);
                sd->has_nulls = true;
            }
    });
}

// Bandwidth selector used for every CKDE the engine fits while scoring (CKDEType::new_factor with construction
// arguments, cv_likelihood.hpp:19-27; selectors kde/NormalReferenceRule.hpp, kde/ScottsBandwidth.hpp).
int pbn_scoredata_set_selector(pbn_scoredata* sd, int selector) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_set_selector: null argument");
        if (selector != PBN_SEL_NORMAL_REFERENCE && selector != PBN_SEL_SCOTT) throw invalid_error("pbn_scoredata_set_selector: unknown selector");
        sd->selector = selector;
        sd->kde_cache.clear();
        sd->term_total.clear();
        sd->score_memo.clear();
    });
}

// Set-function cache of the CKDE likelihood scores: entries held, sweeps launched so far.
int pbn_scoredata_cache_stats(const pbn_scoredata* sd, int64_t* entries, int64_t* sweeps) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_cache_stats: null argument");
        if (entries) *entries = (int64_t)sd->kde_cache.size();
        if (sweeps) *sweeps = sd->kde_sweeps;
    });
}

// The same layout without a table or a device (CrossValidation / HoldOut as stand-alone objects): perm has n_rows
// entries, limits k + 1 (nullable when k <= 1).
int pbn_split_layout(int64_t n_rows, int split, int k, uint32_t seed, double test_ratio, int32_t* perm, int32_t* limits,
                     int64_t* n_cv, int64_t* n_hold) {
    return guarded([&] {
        if (n_rows < 0 || !perm) throw invalid_error("pbn_split_layout: bad argument");
        SplitLayout lay = split_layout(n_rows, split, k, seed, test_ratio);
        std::memcpy(perm, lay.idx.data(), lay.idx.size() * sizeof(int32_t));
        if (limits && !lay.limits.empty()) std::memcpy(limits, lay.limits.data(), lay.limits.size() * sizeof(int32_t));
        if (n_cv) *n_cv = lay.n_cv;
        if (n_hold) *n_hold = lay.n_hold;
    });
}

int pbn_scoredata_layout(const pbn_scoredata* sd, int32_t* perm, int32_t* limits, int64_t* n_cv, int64_t* n_hold) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_layout: null argument");
        if (perm) std::memcpy(perm, sd->perm.data(), sd->perm.size() * sizeof(int32_t));
        if (limits && sd->k > 0) std::memcpy(limits, sd->limits.data(), sd->limits.size() * sizeof(int32_t));
        if (n_cv) *n_cv = sd->n_cv;
        if (n_hold) *n_hold = sd->n_hold;
    });
}

// MLE<LinearGaussianCPD>::estimate over the CV/training region (all rows for PBN_SPLIT_NONE).
int pbn_lg_fit(const pbn_scoredata* sd, int var, const int* parents, int p, double* beta, double* variance) {
    return guarded(mu_of(sd), [&] {
        if (!sd || !beta || !variance) throw invalid_error("pbn_lg_fit: null argument");
        if (sd->partial) throw invalid_error("pbn_lg_fit: row-sharded score data needs pbn_scoredata_moments(set) first");
        std::vector<int> cols(p + 1);
        cols[0] = var;
        for (int i = 0; i < p; ++i) cols[i + 1] = parents[i];
        for (int c : cols)
            if (c < 0 || c >= sd->n) throw invalid_error("pbn_lg_fit: column out of range");
        std::vector<double> mu(p + 1), sse((size_t)(p + 1) * (p + 1));
        subset_moments(sd, sd->all, cols.data(), p + 1, mu.data(), sse.data());
        bool suspect = false;
        *variance = lg_fit(sd->all.N, p, mu.data(), sse.data(), beta, &suspect);
        if (suspect && lg_guard_on() && p + 1 <= 16 && !sd->has_nulls)
            *variance = lg_fit_accurate(sd->table(), cols.data(), p + 1, 0, sd->n_cv, 0, sd->n_cv, nullptr, beta);
    });
}

// MLE<LinearGaussianCPD>::estimate on a table row range (one Gram pass + host solve).
int pbn_lg_fit_table(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* beta, double* variance) {
    return guarded(mu_of(t), [&] {
        if (!beta || !variance) throw invalid_error("pbn_lg_fit_table: null output");
        check_cols(t, cols, d, "pbn_lg_fit_table");
        check_range(t, row0, n, "pbn_lg_fit_table");
        if (d < 1 || d > 64) throw invalid_error("pbn_lg_fit_table: between 1 and 64 columns (variable + evidence) are supported");
        HIP_CHECK(hipSetDevice(t->ctx->device));
        std::vector<double> mu(d), sse((size_t)d * d);
        int rc = pbn_table_sse(t, cols, d, row0, n, mu.data(), sse.data());
        if (rc != PBN_OK) throw device_error(pbn_last_error());
        bool suspect = false;
        *variance = lg_fit(n, d - 1, mu.data(), sse.data(), beta, &suspect);
        if (suspect && lg_guard_on() && d <= 16) *variance = lg_fit_accurate(t, cols, d, row0, n, 0, n, nullptr, beta);
    });
}

// LinearGaussianCPD::logl / slogl (factors/continuous/LinearGaussianCPD.cpp:92-149,251-292).
static int lg_eval(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const double* beta, double variance,
                   double* out_logl, double* out_slogl, int want_cdf) {
    return guarded(mu_of(t), [&] {
        check_cols(t, cols, d, "pbn_lg_logl");
        check_range(t, row0, n, "pbn_lg_logl");
        if (!beta) throw invalid_error("pbn_lg_logl: null beta");
        if (d < 1 || d > 64) throw invalid_error("pbn_lg_logl: between 1 and 64 columns are supported");
        pbn_ctx* ctx = t->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        if (n == 0) { if (out_slogl) *out_slogl = 0.0; return; }
        const int64_t nblocks = ceil_div(n, 256);
        dev_buf<double> dlogl;
        if (out_logl) dlogl.alloc((size_t)n);
        ctx->scratch_misc.reserve((size_t)(nblocks + 1) * sizeof(double));
        double* bs = (double*)ctx->scratch_misc.p;
        LgArgs a{};
        a.base = t->data; a.ld = t->ld; a.p = d - 1; a.row0 = row0; a.n = n;
        for (int i = 0; i < d; ++i) { a.gc.cols[i] = cols[i]; a.beta[i] = beta[i]; }
        a.inv_std = 1.0 / std::sqrt(variance);
        a.cte = -0.5 * std::log(variance) - 0.5 * LOG_2PI;
        a.logl = dlogl.p; a.block_sums = bs; a.want_cdf = want_cdf;
        launch_lg_logl(a, t->dtype, ctx->stream);
        if (out_logl) HIP_CHECK(hipMemcpyAsync(out_logl, dlogl.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        std::vector<double> hb((size_t)nblocks);
        HIP_CHECK(hipMemcpyAsync(hb.data(), bs, (size_t)nblocks * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        if (out_slogl) {
            double s = 0.0;
            for (double v : hb) s += v;  // fixed order
            *out_slogl = s;
        }
    });
}

int pbn_lg_logl(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const double* beta, double variance,
                double* out_logl, double* out_slogl) {
    return lg_eval(t, cols, d, row0, n, beta, variance, out_logl, out_slogl, 0);
}

// LinearGaussianCPD::cdf (factors/continuous/LinearGaussianCPD.cpp:171-249): Phi((y - beta.x) / sigma) per row.
int pbn_lg_cdf(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const double* beta, double variance,
               double* out) {
    if (!out && n > 0) { set_last_error("pbn_lg_cdf: null output"); return PBN_ERR_INVALID; }
    return lg_eval(t, cols, d, row0, n, beta, variance, out, nullptr, 1);
}

// `want` (nullable, per candidate): 0 = the local score; 1 / 2 = only the joint / only the marginal CKDE term of the candidate, summed
// over the regions (pbn_score_terms)
// `only_region` (nullable, per candidate; pbn_score_term_regions): >= 0 = that region's contribution alone (a CV fold; 0 for the hold-out score)
// `parts_rank` / `parts_world` / `parts_out` (pbn_score_batch_parts): hybrid CKDE candidates evaluated on one rank's parts only, per-part sums to
// parts_out[c * PBN_HYBRID_PARTS ...]
static int score_batch_impl(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off,
                            const int* parents, const double* params, int n_params, double* out, const int* want,
                            int parts_rank = 0, int parts_world = 0, double* parts_out = nullptr, const int* only_region = nullptr) {
    return guarded(mu_of(sd), [&] {
        if (!sd || !var || !par_off || !out) throw invalid_error("pbn_score_batch: null argument");
        pbn_ctx* ctx = sd->ctx;
        if (sd->partial) throw invalid_error("pbn_score_batch: row-sharded score data needs pbn_scoredata_moments(set) first");
        HIP_CHECK(hipSetDevice(ctx->device));
        if ((kind == PBN_SCORE_CVLIK) && sd->k <= 0) throw invalid_error("pbn_score_batch: score data has no CV folds");
        if ((kind == PBN_SCORE_HOLDOUT) && sd->n_hold <= 0) throw invalid_error("pbn_score_batch: score data has no hold-out split");
        const pbn_table* t = sd->table();
        // The set-function cache and the local-score memo grow by one entry per (region, configuration, term) and per hybrid
        // candidate (~150 B a node): a long search over hybrid candidates would take gigabytes.  Both are pure accelerators -
        // beyond PBN_SCORE_CACHE_ENTRIES entries (default 2^21, ~300 MB) they start over, at a batch boundary (no entry of
        // the batch being assembled is lost).
        static const size_t cache_budget = (size_t)std::max(1ll, knob_ll("PBN_SCORE_CACHE_ENTRIES", 1ll << 21));
        if (sd->kde_cache.size() > cache_budget) { sd->kde_cache.clear(); ++sd->cache_resets; }
        // (term_total is trimmed in pbn_score_terms_put only: every rank of a job reaches that call with the same state, whereas a rank
        //  whose dealt list is empty skips the evaluation call - trimming here would let the ranks' "missing" lists drift apart)
        if (sd->score_memo.size() > cache_budget) { sd->score_memo.clear(); ++sd->cache_resets; }
        // BGe parameters
        double iss_mu = 1, iss_w = sd->n + 2;
        int total_nodes = sd->n;
        const double* nu = nullptr;
        if (kind == PBN_SCORE_BGE) {
            if (n_params >= 1) iss_mu = params[0];
            if (n_params >= 2) iss_w = params[1];
            if (n_params >= 3) total_nodes = (int)params[2];
            if (n_params >= 3 + sd->n) nu = params + 3;
        }
        struct Pending { int cand; int unit0; int units; };
        std::vector<Pending> pending;
        // hybrid CKDE candidates of this call go into one HybridBatch (hybrid.hip): enqueued as they come, finished together after the loop
        const bool hybrid_batched = knob_int("PBN_HYBRID_BATCH", 1) != 0;   // (per call: the tests switch it)
        std::unique_ptr<HybridBatch, void (*)(HybridBatch*) noexcept> hbatch(nullptr, hybrid_batch_end);
        auto hybrid_batch = [&]() -> HybridBatch* {
            if (!hybrid_batched) return nullptr;
            if (!hbatch) hbatch.reset(hybrid_batch_begin(sd));
            return hbatch.get();
        };
        int n_units = 0;
        Stats train;
        std::vector<int> cols;
        std::vector<double> mu, sse, beta, H;
        for (int c = 0; c < n_cand; ++c) {
            const int p = par_off[c + 1] - par_off[c];
            const int d = p + 1;
            cols.resize(d);
            cols[0] = var[c];
            for (int i = 0; i < p; ++i) cols[i + 1] = parents[par_off[c] + i];
            bool hybrid = false;
            for (int cc : cols) {
                if (cc < 0 || cc >= sd->n + sd->n_disc) throw invalid_error("pbn_score_batch: column out of range");
                if (cc >= sd->n) hybrid = true;
            }
            const int nt = node_type ? node_type[c] : PBN_NODE_LG;
            if (hybrid) {
                // Likelihood scores of DiscreteAdaptator factors cost k x configurations sweeps / Gram passes each, and a
                // hill-climb asks for the same (variable, type, parent SET) again and again: every RemoveArc cell is the
                // local score the node had before that arc was added, every FlipArc cell's source side likewise.  They
                // are remembered by parent set (score_hybrid evaluates the continuous parents in ascending order and the discrete
                // ones in canonical order: the value does not depend on the order they were given in).  BIC stays out: bic_clg ties.
                if (parts_out) {   // a share of the candidate: never memoised
                    HybridParts hp{parts_rank, parts_world, parts_out + (size_t)c * PBN_HYBRID_PARTS};
                    HybridSink sink{out + c, {}};
                    out[c] = score_hybrid(sd, kind, cols[0], nt, cols.data() + 1, p, &hp, hybrid_batch(), &sink);
                    continue;
                }
                const bool memo = score_memo_on() && !sd->force_precise && (kind == PBN_SCORE_CVLIK || kind == PBN_SCORE_HOLDOUT);   // (discrete factors too: a count over all rows on the host each)
                std::vector<int> key;
                if (memo) {
                    key.assign(cols.begin() + 1, cols.end());
                    std::sort(key.begin(), key.end());
                    key.insert(key.begin(), {kind, nt, cols[0]});
                    auto it = sd->score_memo.find(key);
                    if (it != sd->score_memo.end()) { out[c] = it->second; ++sd->memo_hits; continue; }
                }
                HybridSink sink{out + c, key};   // (key empty without the memo)
                bool deferred = false;
                out[c] = score_hybrid(sd, kind, cols[0], nt, cols.data() + 1, p, nullptr, hybrid_batch(), &sink, &deferred);
                if (memo && !deferred) sd->score_memo[key] = out[c];
                continue;
            }
            mu.resize(d); sse.resize((size_t)d * d); beta.resize(d);
            const Stats* full = &sd->all;
            Stats gathered;
            if (sd->has_nulls && (kind == PBN_SCORE_BIC || kind == PBN_SCORE_BGE)) {
                bool involved = false;
                for (int cc : cols) involved = involved || !sd->valid[cc].empty();
                if (involved) {
                    // rows valid in every involved column (DataFrame::combined_bitmap, dataset.cpp:208-235)
                    const int64_t rows = (int64_t)sd->perm.size();
                    std::vector<int32_t> keep;
                    keep.reserve(rows);
                    for (int64_t r = 0; r < rows; ++r) {
                        bool ok = true;
                        for (int cc : cols) ok = ok && (sd->valid[cc].empty() || sd->valid[cc][r]);
                        if (ok) keep.push_back((int32_t)r);
                    }
                    if (d > 64) throw invalid_error("pbn_score_batch: more than 64 columns in one candidate");
                    sd->rows_dev.reserve(keep.size() + 16);
                    if (!keep.empty())
                        HIP_CHECK(hipMemcpyAsync(sd->rows_dev.p, keep.data(), keep.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
                    std::vector<double> S(d), G((size_t)d * d);
                    if (!keep.empty())
                        gram_raw(t, cols.data(), d, 0, (int64_t)keep.size(), sd->rows_dev.p, sd->shift_dev.p, S.data(), G.data());
                    gathered.zero(sd->n);
                    gathered.N = (int64_t)keep.size();
                    for (int i = 0; i < d; ++i) {
                        gathered.S[cols[i]] = S[i];
                        for (int j = 0; j < d; ++j) gathered.G[cols[i] + (size_t)cols[j] * sd->n] = G[i + (size_t)j * d];
                    }
                    full = &gathered;
                }
            }
            if (kind == PBN_SCORE_BIC) {
                if (nt != PBN_NODE_LG) throw invalid_error("BIC: only LinearGaussianCPD node types are implemented on device");
                subset_moments(sd, *full, cols.data(), d, mu.data(), sse.data());
                bool suspect = false;
                double v = lg_fit(full->N, p, mu.data(), sse.data(), beta.data(), &suspect);
                if (suspect && lg_guard_on() && d <= 16) {   // nearly collinear parents / nearly exact fit: double-double refit
                    if (full == &gathered) v = lg_fit_accurate(t, cols.data(), d, 0, full->N, 0, full->N, sd->rows_dev.p, beta.data());
                    else v = lg_fit_accurate(t, cols.data(), d, 0, sd->n_cv, 0, sd->n_cv, nullptr, beta.data());
                }
                out[c] = bic_lg(full->N, p, v);
                continue;
            }
            if (kind == PBN_SCORE_BGE) {
                out[c] = bge_score(sd, *full, cols.data(), p, iss_mu, iss_w, total_nodes, nu);
                continue;
            }
            const bool cv = kind == PBN_SCORE_CVLIK;
            const int units = cv ? sd->k : 1;
            if (nt == PBN_NODE_LG) {
                double acc = 0.0;
                for (int f = 0; f < units; ++f) {
                    const Stats* tr = &sd->all;
                    const Stats* te = &sd->hold;
                    if (cv) { stats_minus(sd->all, sd->fold[f], train); tr = &train; te = &sd->fold[f]; }
                    subset_moments(sd, *tr, cols.data(), d, mu.data(), sse.data());
                    bool suspect = false;
                    double v = lg_fit(tr->N, p, mu.data(), sse.data(), beta.data(), &suspect);
                    if (suspect && lg_guard_on() && d <= 16) {
                        if (cv) v = lg_fit_accurate(t, cols.data(), d, 0, sd->limits[f], sd->limits[f + 1], tr->N, nullptr, beta.data());
                        else v = lg_fit_accurate(t, cols.data(), d, 0, sd->n_cv, 0, sd->n_cv, nullptr, beta.data());
                        // and the test fold's log-likelihood from its rows, as the reference evaluates it
                        const int64_t te0 = cv ? sd->limits[f] : sd->n_cv;
                        acc += lg_slogl_from_rows(t, cols.data(), d, te0, te->N, nullptr, beta.data(), v);
                        continue;
                    }
                    acc += lg_slogl_from_moments(sd, *te, cols.data(), p, beta.data(), v);
                }
                out[c] = acc;
                continue;
            }
            if (nt != PBN_NODE_CKDE) throw invalid_error("pbn_score_batch: unknown node type");
            // CKDE: enqueue fit + slogl per unit from scratch arenas
            ctx->scratch_misc.reserve(1);  // touched by kde_eval_enqueue
            pending.push_back({c, n_units, units});
            n_units += units;
        }
        hybrid_batch_flush(hbatch.get());   // (before the plain CKDE units below take the context's result slots)
        hbatch.reset();
        if (!pending.empty()) {
            // ---- CKDE likelihood units through the set-function cache ------------------------------------------------
            // slogl of a CKDE on a test region = A(vars, d) - A(parents, d), with A(S, m) = sum_q log KDE(S) under the
            // bandwidth rule evaluated for m dimensions on the region's training rows: the marginal of the reference
            // (CKDE.hpp:186-199) is the KDE of the parents with H[1:,1:], and H = k(N, d) cov, so both terms are
            // functions of a variable SET (and m, and the region) only.  A({s,t}, 2) serves s -> t and t -> s,
            // A({s}, 2) serves every child of s: each is swept once and remembered for the life of the score data.
            // Neither term known: one fused sweep yields both; one known: a plain sweep of the other.
            struct Term { double value = 0; int slot = -1; };           // cached value, or slot in the device sums
            std::map<std::vector<int>, int> scheduled;                    // key -> slot (this batch)
            std::vector<std::vector<int>> slot_key;
            auto key_of = [&](int region, int m, const int* v, int nv) {
                std::vector<int> k(v, v + nv);
                std::sort(k.begin(), k.end());
                k.insert(k.begin(), {region, m});
                return k;
            };
            const bool precise_all = sd->force_precise && sd->dtype == PBN_F64;
            auto lookup = [&](const std::vector<int>& key, Term& t) {
                auto it = sd->kde_cache.find(key);
                if (!precise_all && it != sd->kde_cache.end()) { t.value = it->second; return true; }
                auto is = scheduled.find(key);
                if (is != scheduled.end()) { t.slot = is->second; return true; }
                return false;
            };
            auto new_slot = [&](const std::vector<int>& key) {
                const int slot = (int)slot_key.size();
                slot_key.push_back(key);
                scheduled[key] = slot;
                return slot;
            };
            struct Unit { int cand; Term joint, marg; bool has_marg; int want; };
            // a term whose total over the regions was installed (pbn_score_terms_put): the total stands for region 0, the others add 0
            auto lookup_total = [&](int m, const int* v, int nv, int region_index, Term& t) {
                if (sd->term_total.empty() || only_region || precise_all) return false;   // (one region of a term: never its total)
                std::vector<int> k(v, v + nv);
                std::sort(k.begin(), k.end());
                k.insert(k.begin(), {kind, m});   // a validated score asks one handle for both kinds
                auto it = sd->term_total.find(k);
                if (it == sd->term_total.end()) return false;
                t.value = region_index == 0 ? it->second : 0.0;
                return true;
            };
            std::vector<Unit> units_v;
            struct Work { int cand, f, mode, slot_j, slot_m; };          // mode 1 joint term, 2 marginal term
            std::vector<Work> work;
            const bool cv = kind == PBN_SCORE_CVLIK;
            // Neither term known: two plain sweeps (joint, marginal), each pruned on its own box.  (The fused joint + marginal sweep costs
            // exactly two sweeps' worth of exponentials anyway - they, not the MFMAs, are the cost - and under tile pruning it can only
            // prune on the marginal box: C3 16-21 ms fused against 2 x 7 ms plain per fold.  The engine dropped it in round 2; the
            // stand-alone CKDE handles keep the fused kernel.)
            for (const Pending& pd : pending) {
                const int c = pd.cand;
                const int p = par_off[c + 1] - par_off[c], d = p + 1;
                cols.resize(d);
                cols[0] = var[c];
                for (int i = 0; i < p; ++i) cols[i + 1] = parents[par_off[c] + i];
                for (int f = 0; f < pd.units; ++f) {
                    if (only_region && only_region[c] >= 0 && f != only_region[c]) continue;
                    const int region = cv ? f : sd->k;                    // fold index, or the hold-out region
                    const int wnt = want ? want[c] : 0;
                    Unit u{c, {}, {}, wnt == 3 || (p > 0 && wnt != 1), wnt};
                    const std::vector<int> kj = key_of(region, d, cols.data(), d);
                    const bool have_j = wnt >= 2 || lookup_total(d, cols.data(), d, f, u.joint) || lookup(kj, u.joint);
                    bool have_m = true;
                    std::vector<int> km;
                    if (wnt == 3) {   // a marginal TERM (pbn_score_terms): all d columns of the pseudo-candidate under the rule for d + 1 dimensions, no child
                        km = key_of(region, d + 1, cols.data(), d);
                        have_m = lookup_total(d + 1, cols.data(), d, f, u.marg) || lookup(km, u.marg);
                    } else if (u.has_marg) { km = key_of(region, d, cols.data() + 1, p); have_m = lookup_total(d, cols.data() + 1, p, f, u.marg) || lookup(km, u.marg); }
                    if (!have_j && !have_m) {
                        u.joint.slot = new_slot(kj); u.marg.slot = new_slot(km);
                        work.push_back({c, f, 1, u.joint.slot, -1});
                        work.push_back({c, f, 2, -1, u.marg.slot});
                    } else if (!have_j) {
                        u.joint.slot = new_slot(kj);
                        work.push_back({c, f, 1, u.joint.slot, -1});
                    } else if (!have_m) {
                        u.marg.slot = new_slot(km);
                        work.push_back({c, f, 2, -1, u.marg.slot});
                    }
                    units_v.push_back(u);
                }
            }
            const size_t nslots = slot_key.size();
            // fp32 tables: behind the sums, one slot per sum for |z|^2 of the evaluation's farthest whitened training row (the pack
            // kernels report it): an evaluation that kde_wants_widening() flags is redone on fp64 fragments (KdeModel::widen) before its
            // value is used.  The choice is a function of that evaluation alone - nothing is remembered per variable set, so the double
            // a (term, region) gets does not depend on what this process evaluated before or on how a job dealt its terms
            static const bool check_after = PBN_TUNE(F32_CHECK, 1) != 0;   // 0: measurement only
            const bool f32 = sd->dtype == PBN_F32 && check_after;
            ctx->scratch_sums.reserve(std::max<size_t>(1, 2 * nslots));   // (grow-only: no hipMalloc / hipFree per batch)
            struct { double* p; } dsums{ctx->scratch_sums.p};
            HIP_CHECK(hipMemsetAsync(dsums.p, 0, std::max<size_t>(1, 2 * nslots) * sizeof(double), ctx->stream));
            double* const dmax = f32 ? dsums.p + nslots : nullptr;
            auto align = [](size_t x) { return (x + 255) / 256 * 256; };
            // host side of one evaluation: columns, training moments of the region, bandwidth, whitening (KdeModel without packs)
            struct Prep { KdeModel m; std::vector<int> use; int64_t row0, n0, row1, te0, te_n, ntrain; };
            auto prepare = [&](const Work& w, Prep& pr) {
                const int c = w.cand;
                const int p = par_off[c + 1] - par_off[c], d = p + 1;
                cols.resize(d);
                cols[0] = var[c];
                for (int i = 0; i < p; ++i) cols[i + 1] = parents[par_off[c] + i];
                mu.resize(d); sse.resize((size_t)d * d); H.resize((size_t)d * d);
                const Stats* tr = &sd->all;
                pr.row0 = 0; pr.n0 = sd->n_cv; pr.row1 = 0; pr.te0 = sd->n_cv; pr.te_n = sd->n_hold;
                if (cv) {
                    stats_minus(sd->all, sd->fold[w.f], train);
                    tr = &train;
                    pr.n0 = sd->limits[w.f]; pr.row1 = sd->limits[w.f + 1];
                    pr.te0 = sd->limits[w.f]; pr.te_n = sd->limits[w.f + 1] - sd->limits[w.f];
                }
                pr.ntrain = tr->N;
                // The plain terms are evaluated over their columns in ASCENDING order, whatever the candidate's orientation and the order
                // of its parents: A(S, m) is then a function of the set alone down to the last bit - the value a term gets does not
                // depend on which candidate asked for it first, and a job that evaluates the terms on other ranks (pbn_score_terms)
                // computes the very same doubles.  (H = k(N, d) cov: the bandwidth of a subset is the sub-block of the set's.)
                const int var0 = cols[0];
                std::sort(cols.begin(), cols.end());
                subset_moments(sd, *tr, cols.data(), d, mu.data(), sse.data());
                const double inv = 1.0 / (double)(tr->N - 1);
                for (auto& x : sse) x *= inv;  // covariance
                if (w.mode == 2 && want && want[c] == 3) {
                    // a marginal term evaluated for itself (a job that deals the terms to its ranks): KDE of these d columns with
                    // H = k(N, d + 1) cov - the block the joint bandwidth of ANY child over them would have (H = k(N, dims) cov for the
                    // library selectors), bit for bit, without a child whose own degeneracy (a constant column ...) has nothing to do
                    // with the term.  The pre-checks of the selector stand on the term's own columns; the real candidate's set is
                    // checked by its joint term.
                    bandwidth_full_block(sd->selector, sse.data(), d, d + 1, tr->N, sd->dtype, H.data());
                    kde_prepare(pr.m, sd->dtype, d, tr->N, H.data(), PBN_BW_FULL, false, mu.data());
                    pr.use = cols;
                    return;
                }
                bandwidth_from_cov(sd->selector, PBN_BW_FULL, sse.data(), d, tr->N, sd->dtype, H.data());
                if (w.mode == 2) {            // KDE of the parents with the bandwidth block of the parents
                    const int vp = (int)(std::find(cols.begin(), cols.end(), var0) - cols.begin());
                    std::vector<double> Hm((size_t)p * p), mum((size_t)p);
                    pr.use.clear();
                    for (int j = 0, jj = 0; j < d; ++j) {
                        if (j == vp) continue;
                        for (int i = 0, ii = 0; i < d; ++i) {
                            if (i == vp) continue;
                            Hm[ii + (size_t)jj * p] = H[i + (size_t)j * d];
                            ++ii;
                        }
                        mum[jj] = mu[j];
                        pr.use.push_back(cols[j]);
                        ++jj;
                    }
                    kde_prepare(pr.m, sd->dtype, p, tr->N, Hm.data(), PBN_BW_FULL, false, mum.data());
                } else {
                    kde_prepare(pr.m, sd->dtype, d, tr->N, H.data(), PBN_BW_FULL, false, mu.data());
                    pr.use = cols;
                }
            };
            // ---- grouped evaluation (kde_group.hip): the plain terms whose shape qualifies are collected into pools - one per
            // variable set, its units the regions asked for - and evaluated by ONE launch chain per batch of pools ---------------
            struct Batch { GroupBatch gb; std::vector<std::vector<GUnit>> pool_units; std::map<std::vector<int>, int> pool_of; };   // pool_of: [m, sorted columns...] -> pool
            Batch bt[1];
            std::vector<char> grouped(work.size(), 0);
            const int R = cv ? sd->k : 2;
            const int64_t min_train = cv ? sd->n_cv - (sd->limits[1] - sd->limits[0]) : sd->n_cv;
            for (size_t wi = 0; wi < work.size(); ++wi) {
                const Work& w = work[wi];
                const int p = par_off[w.cand + 1] - par_off[w.cand];
                const bool own_term = w.mode == 2 && want && want[w.cand] == 3;   // marginal term over all p + 1 columns, rule for p + 2
                const int dims = own_term ? p + 1 : (w.mode == 2 ? p : p + 1);
                if (precise_all || !kde_group_applies(sd->dtype, dims, min_train, R)) continue;   // (precise: one chain per term, per-row accuracy)
                Prep pr;
                prepare(w, pr);
                std::vector<int> key(pr.use);
                std::sort(key.begin(), key.end());
                Batch& B = bt[0];
                GroupBatch& gb = B.gb;
                std::vector<std::vector<GUnit>>& pool_units = B.pool_units;
                key.insert(key.begin(), own_term ? p + 2 : p + 1);
                auto it = B.pool_of.find(key);
                int pi;
                if (it == B.pool_of.end()) {
                    pi = (int)gb.pools.size();
                    B.pool_of[key] = pi;
                    GPool P{};
                    P.rows = nullptr; P.row_base = 0;
                    P.n = (int32_t)(cv ? sd->n_cv : sd->n_cv + sd->n_hold);
                    P.R = R; P.d = dims; P.kd = std::min(dims, 4);
                    if (cv) for (int f = 0; f <= sd->k; ++f) P.rb[f] = sd->limits[f];
                    else { P.rb[0] = 0; P.rb[1] = (int32_t)sd->n_cv; P.rb[2] = (int32_t)(sd->n_cv + sd->n_hold); }
                    for (int r = 0; r < PBN_GROUP_MAX_R; ++r) P.test_unit[r] = -1;
                    for (int i = 0; i < dims; ++i) P.cols[i] = pr.use[i];
                    // pool-level standardisation for the Morton keys: L^-1 of the covariance of the whole CV / training region
                    std::vector<double> gm(dims), gs((size_t)dims * dims), L((size_t)dims * dims), Li((size_t)dims * dims);
                    subset_moments(sd, sd->all, pr.use.data(), dims, gm.data(), gs.data());
                    for (auto& x : gs) x /= (double)std::max<int64_t>(1, sd->all.N - 1);
                    if (!hm::cholesky(gs.data(), dims, L.data())) {   // the keys only order the rows: fall back to the diagonal
                        std::fill(L.begin(), L.end(), 0.0);
                        for (int i = 0; i < dims; ++i) L[i + (size_t)i * dims] = std::sqrt(std::max(gs[i + (size_t)i * dims], 1e-300));
                    }
                    hm::lower_inverse(L.data(), dims, Li.data());
                    for (int i = 0; i < dims; ++i) {
                        P.mug[i] = gm[i];
                        for (int j = 0; j < dims; ++j) P.Wg[i * dims + j] = j <= i ? Li[i + (size_t)j * dims] : 0.0;
                    }
                    gb.pools.push_back(P);
                    pool_units.emplace_back();
                } else {
                    pi = it->second;
                }
                GUnit U{};
                U.test_region = cv ? w.f : 1;
                U.train_mask = cv ? (((R >= 64) ? ~0ull : ((1ull << R) - 1ull)) & ~(1ull << w.f)) : 1ull;
                U.N = (int32_t)pr.ntrain; U.nq = (int32_t)pr.te_n;
                U.lognorm = pr.m.lognorm;
                for (int i = 0; i < dims * dims; ++i) U.W[i] = pr.m.W[i];
                for (int i = 0; i < dims; ++i) U.mu[i] = pr.m.mu[i];
                U.sum_slot = w.mode == 2 ? w.slot_m : w.slot_j;
                gb.pools[pi].test_unit[U.test_region] = (int32_t)pool_units[pi].size();
                pool_units[pi].push_back(U);
                grouped[wi] = 1;
            }
            for (Batch& B : bt)
                for (size_t pi = 0; pi < B.gb.pools.size(); ++pi) {
                    B.gb.pools[pi].unit0 = (int32_t)B.gb.units.size();
                    B.gb.pools[pi].nunits = (int32_t)B.pool_units[pi].size();
                    B.gb.units.insert(B.gb.units.end(), B.pool_units[pi].begin(), B.pool_units[pi].end());
                }
            size_t n_legacy = 0;
            for (char gflag : grouped) n_legacy += gflag ? 0 : 1;
            // independent evaluations alternate between the context's two issue lanes (common.hpp)
            const int lanes = (n_legacy > 1 && !ctx->profiling) ? score_lanes(t->n_rows) : 1;
            if (lanes > 1) { ctx->ensure_lanes(lanes - 1); ctx->lanes_wait_for_stream(lanes - 1); }
            if (!bt[0].gb.pools.empty()) kde_group_run(ctx, t, bt[0].gb, dsums.p, dmax, false);
            // one evaluation through its own launch chain (shapes the grouped path does not take; the redo of a flagged evaluation)
            auto run_single = [&](const Work& w, bool force64, bool precise = false) {
                Prep pr;
                prepare(w, pr);
                KdeModel& m = pr.m;
                const int* use_cols = pr.use.data();
                const int slot = w.mode == 2 ? w.slot_m : w.slot_j;
                if (f32 && force64) kde_widen(m);
                const KdePackBytes pb = kde_pack_bytes(m.fdtype(), m.dm, m.cond, m.N);
                ctx->scratch_train.reserve(pb.apack + pb.nxpack + pb.axpack + 768);
                char* base = ctx->scratch_train.p;
                m.Apack = base;
                m.nxpack = base + align(pb.apack);
                m.Axpack = m.cond ? base + align(pb.apack) + align(pb.nxpack) : nullptr;
                kde_pack_train(ctx, m, t, use_cols, pr.row0, pr.n0, pr.row1, nullptr, /*prune=*/true, dmax ? dmax + slot : nullptr);
                kde_eval_enqueue(ctx, m, t, use_cols, pr.te0, pr.te_n, nullptr, dsums.p + slot, nullptr, nullptr, precise);
            };
            size_t wi = 0, li = 0;
            for (const Work& w : work) {
                if (grouped[wi++]) continue;
                LaneSwitch lane(ctx, (int)(li++ % (size_t)lanes));
                run_single(w, false, precise_all);
            }
            std::vector<double> hs(std::max<size_t>(1, 2 * nslots));
            if (lanes > 1) ctx->sync_lanes(lanes - 1);
            if (nslots) HIP_CHECK(hipMemcpyAsync(hs.data(), dsums.p, (f32 ? 2 : 1) * nslots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            if (f32 && nslots) {
                // check-after: evaluations whose training rows reach beyond what fp32 fragments hold (2^-24 max|z|^2 above the threshold of
                // kde_wants_widening) are redone on fp64 fragments
                std::vector<const Work*> redo;
                for (const Work& w : work) {
                    const int slot = w.mode == 2 ? w.slot_m : w.slot_j;
                    // the criterion follows the fragments' layout, i.e. the number of variables of the term: p + 1 for a joint term, p or (a
                    // marginal TERM of pbn_score_terms) p + 1 for a marginal one - the stricter of the two there
                    const int pw = par_off[w.cand + 1] - par_off[w.cand];
                    const double far2 = hs[nslots + (size_t)slot];
                    if (kde_wants_widening(far2, pw + 1) || (w.mode == 2 && kde_wants_widening(far2, pw))) redo.push_back(&w);
                }
                if (!redo.empty()) {
                    for (const Work* w : redo) {
                        Prep pr;
                        prepare(*w, pr);
                        const int slot = w->mode == 2 ? w->slot_m : w->slot_j;
                        HIP_CHECK(hipMemsetAsync(dsums.p + slot, 0, sizeof(double), ctx->stream));
                        run_single(*w, true);
                    }
                    HIP_CHECK(hipMemcpyAsync(hs.data(), dsums.p, nslots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    sd->kde_sweeps += (int64_t)redo.size();
                }
            }
            if (sd->dtype == PBN_F64 && nslots) {
                // fp64 tables: a term whose sum over its test rows is a cancellation to ~0 (kde_sum_needs_precision: the sum-only sweeps'
                // absolute error budget would exceed 5e-7 of it) is evaluated once more at the accuracy of the per-row path
                std::vector<const Work*> redo;
                for (const Work& w : work) {
                    const int slot = w.mode == 2 ? w.slot_m : w.slot_j;
                    const int64_t nq = cv ? sd->limits[w.f + 1] - sd->limits[w.f] : sd->n_hold;
                    if (kde_sum_needs_precision(hs[(size_t)slot], nq)) redo.push_back(&w);
                }
                if (!redo.empty()) {
                    for (const Work* w : redo) {
                        const int slot = w->mode == 2 ? w->slot_m : w->slot_j;
                        HIP_CHECK(hipMemsetAsync(dsums.p + slot, 0, sizeof(double), ctx->stream));
                        run_single(*w, false, /*precise=*/true);
                    }
                    HIP_CHECK(hipMemcpyAsync(hs.data(), dsums.p, nslots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
                    HIP_CHECK(hipStreamSynchronize(ctx->stream));
                    sd->kde_sweeps += (int64_t)redo.size();
                    sd->precise_redos += (int64_t)redo.size();
                }
            }
            ctx->drop_staged();
            for (size_t i = 0; i < nslots; ++i) sd->kde_cache[slot_key[i]] = hs[i];
            sd->kde_sweeps += (int64_t)work.size();
            // a candidate = (its joint terms added in region order) - (its marginal terms added in region order): the two sums are
            // what pbn_score_terms hands out, so a job that computes the terms on different ranks assembles the same doubles
            std::vector<double> jsum((size_t)n_cand, 0.0), msum((size_t)n_cand, 0.0);
            for (const Unit& u : units_v) {
                if (u.want < 2) jsum[u.cand] += u.joint.slot >= 0 ? hs[u.joint.slot] : u.joint.value;
                if (u.has_marg) msum[u.cand] += u.marg.slot >= 0 ? hs[u.marg.slot] : u.marg.value;
            }
            for (const Pending& pd : pending) {
                const int wnt = want ? want[pd.cand] : 0;
                out[pd.cand] = wnt >= 2 ? msum[pd.cand] : (wnt == 1 ? jsum[pd.cand] : jsum[pd.cand] - msum[pd.cand]);
            }
        }
    });
}

int pbn_scoredata_set_precise(pbn_scoredata* sd, int on) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_set_precise: null handle");
        sd->force_precise = on != 0;
    });
}

int pbn_score_batch(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off,
                    const int* parents, const double* params, int n_params, double* out) {
    if (sd && sd->has_comm) {   // one process per GPU (pbn_scoredata_set_comm): this rank's share + one all-gather (shard.hip)
        // (no lock held across the host's collective: the evaluations inside take the handle's lock themselves)
        try {
            pbn::score::score_batch_sharded(sd, kind, n_cand, var, node_type, par_off, parents, params, n_params, out);
            return PBN_OK;
        } catch (const invalid_error& e) { set_last_error(e.what()); return PBN_ERR_INVALID;
        } catch (const singular_error& e) { set_last_error(e.what()); return PBN_ERR_SINGULAR;
        } catch (const std::exception& e) { set_last_error(e.what()); return PBN_ERR_DEVICE; }
    }
    return score_batch_impl(sd, kind, n_cand, var, node_type, par_off, parents, params, n_params, out, nullptr);
}

extern "C++" {
namespace pbn {
namespace score {
int score_batch_local(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents,
                      const double* params, int n_params, double* out) {
    return score_batch_impl(sd, kind, n_cand, var, node_type, par_off, parents, params, n_params, out, nullptr);
}
}  // namespace score
}  // namespace pbn
}  // extern "C++"

extern "C++" {
namespace {
// term i = columns vars[off[i] .. off[i + 1]) under the bandwidth rule for m[i] dimensions: m = the number of columns for a joint term,
// one more for the marginal term of a candidate with those parents
void check_terms(const pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, const char* who) {
    if (!sd || (n_terms > 0 && (!off || !vars || !m))) throw invalid_error(std::string(who) + ": null argument");
    if (kind != PBN_SCORE_CVLIK && kind != PBN_SCORE_HOLDOUT) throw invalid_error(std::string(who) + ": likelihood scores only");
    for (int i = 0; i < n_terms; ++i) {
        const int nv = off[i + 1] - off[i];
        if (nv < 1 || (m[i] != nv && m[i] != nv + 1)) throw invalid_error(std::string(who) + ": a term has m = its columns (joint) or one more (marginal)");
        for (int j = off[i]; j < off[i + 1]; ++j)
            if (vars[j] < 0 || vars[j] >= sd->n) throw invalid_error(std::string(who) + ": continuous columns only");
    }
}
std::vector<int> term_key(int kind, const int* v, int nv, int m) {
    std::vector<int> k(v, v + nv);
    std::sort(k.begin(), k.end());
    k.insert(k.begin(), {kind, m});
    return k;
}
}  // namespace
}  // extern "C++"

int pbn_score_terms(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, double* out) {
    return guarded(mu_of(sd), [&] {
        check_terms(sd, kind, n_terms, off, vars, m, "pbn_score_terms");
        if (n_terms > 0 && !out) throw invalid_error("pbn_score_terms: null output");
        // every term as a pseudo-candidate of the batch engine over the term's OWN columns (first column | the others): a joint term
        // asks for the joint half only (want 1), a marginal term for "the KDE of all my columns under the rule for one more dimension"
        // (want 3) - no pseudo-child: the bandwidth block of the parents does not depend on the child (H = k(N, d) cov for the library
        // selectors), and a degenerate unrelated column must not fail a term no real candidate pairs with it
        std::vector<int> var((size_t)n_terms), nt((size_t)n_terms, PBN_NODE_CKDE), po{0}, par, want((size_t)n_terms);
        for (int i = 0; i < n_terms; ++i) {
            const int nv = off[i + 1] - off[i];
            const int* v = vars + off[i];
            var[i] = v[0];
            want[i] = m[i] == nv ? 1 : 3;
            par.insert(par.end(), v + 1, v + nv);
            po.push_back((int)par.size());
        }
        if (n_terms > 0) {
            const int rc = score_batch_impl(sd, kind, n_terms, var.data(), nt.data(), po.data(), par.data(), nullptr, 0, out, want.data());
            // the inner status class goes out unchanged: a SingularCovarianceData of a term is one on every rank, not a device error
            if (rc == PBN_ERR_SINGULAR) throw singular_error(pbn_last_error());
            if (rc == PBN_ERR_INVALID) throw invalid_error(pbn_last_error());
            if (rc != PBN_OK) throw device_error(pbn_last_error());
        }
    });
}

int pbn_score_term_regions(pbn_scoredata* sd, int kind, int n_items, const int* off, const int* vars, const int* m, const int* region, double* out) {
    return guarded(mu_of(sd), [&] {
        check_terms(sd, kind, n_items, off, vars, m, "pbn_score_term_regions");
        if (n_items > 0 && (!out || !region)) throw invalid_error("pbn_score_term_regions: null argument");
        const int regions = kind == PBN_SCORE_CVLIK ? sd->k : 1;
        std::vector<int> var((size_t)n_items), nt((size_t)n_items, PBN_NODE_CKDE), po{0}, par, want((size_t)n_items);
        for (int i = 0; i < n_items; ++i) {
            if (region[i] < 0 || region[i] >= regions) throw invalid_error("pbn_score_term_regions: region out of range (CV folds; 0 for the hold-out score)");
            const int nv = off[i + 1] - off[i];
            const int* v = vars + off[i];
            var[i] = v[0];
            want[i] = m[i] == nv ? 1 : 3;   // as in pbn_score_terms
            par.insert(par.end(), v + 1, v + nv);
            po.push_back((int)par.size());
        }
        if (n_items > 0) {
            const int rc = score_batch_impl(sd, kind, n_items, var.data(), nt.data(), po.data(), par.data(), nullptr, 0, out, want.data(), 0, 0, nullptr, region);
            if (rc == PBN_ERR_SINGULAR) throw singular_error(pbn_last_error());
            if (rc == PBN_ERR_INVALID) throw invalid_error(pbn_last_error());
            if (rc != PBN_OK) throw device_error(pbn_last_error());
        }
    });
}

int pbn_score_terms_put(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, const double* values) {
    return guarded(mu_of(sd), [&] {
        check_terms(sd, kind, n_terms, off, vars, m, "pbn_score_terms_put");
        if (n_terms > 0 && !values) throw invalid_error("pbn_score_terms_put: null values");
        static const size_t budget = (size_t)std::max(1ll, knob_ll("PBN_SCORE_CACHE_ENTRIES", 1ll << 21));
        if (sd->term_total.size() > budget) { sd->term_total.clear(); ++sd->cache_resets; }   // before the new terms go in: the batch being assembled keeps its own
        for (int i = 0; i < n_terms; ++i) sd->term_total[term_key(kind, vars + off[i], off[i + 1] - off[i], m[i])] = values[i];
    });
}

int pbn_score_batch_parts(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off,
                          const int* parents, int part, int n_parts, double* out) {
    return guarded(mu_of(sd), [&] {
        if (!sd || !var || !node_type || !par_off || !out) throw invalid_error("pbn_score_batch_parts: null argument");
        if (n_parts < 1 || n_parts > PBN_HYBRID_PARTS || part < 0 || part >= n_parts) throw invalid_error("pbn_score_batch_parts: part / n_parts out of range (at most 64 parts)");
        for (int c = 0; c < n_cand; ++c) {
            bool disc = false;
            for (int j = par_off[c]; j < par_off[c + 1]; ++j) disc = disc || parents[j] >= sd->n;
            if (!disc || node_type[c] != PBN_NODE_CKDE) throw invalid_error("pbn_score_batch_parts: CKDE candidates with discrete parents only");
        }
        std::vector<double> whole((size_t)std::max(n_cand, 1));
        std::fill(out, out + (size_t)n_cand * PBN_HYBRID_PARTS, 0.0);
        if (n_cand > 0) {
            const int rc = score_batch_impl(sd, kind, n_cand, var, node_type, par_off, parents, nullptr, 0, whole.data(), nullptr, part, n_parts, out);
            if (rc == PBN_ERR_SINGULAR) throw singular_error(pbn_last_error());
            if (rc == PBN_ERR_INVALID) throw invalid_error(pbn_last_error());
            if (rc != PBN_OK) throw device_error(pbn_last_error());
        }
    });
}

int pbn_score_terms_missing(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, int* missing) {
    return guarded(mu_of(sd), [&] {
        check_terms(sd, kind, n_terms, off, vars, m, "pbn_score_terms_missing");
        if (n_terms > 0 && !missing) throw invalid_error("pbn_score_terms_missing: null output");
        for (int i = 0; i < n_terms; ++i)
            missing[i] = (sd->force_precise && sd->dtype == PBN_F64) ||   // (precise mode: every term is evaluated again, by the rank it is dealt to)
                         sd->term_total.find(term_key(kind, vars + off[i], off[i + 1] - off[i], m[i])) == sd->term_total.end() ? 1 : 0;
    });
}

}  // extern "C"
