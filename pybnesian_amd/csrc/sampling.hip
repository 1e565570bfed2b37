// Sampling from fitted factors: CKDE::sample (factors/continuous/CKDE.hpp:289-508 + the normalize_accum_sum_mat_cols /
// find_random_indices kernels, kde/opencl_kernels/KDE.cl.src:351-374), LinearGaussianCPD::sample
// (factors/continuous/LinearGaussianCPD.cpp:317-380) and DiscreteFactor::sample_indices
// (factors/discrete/DiscreteFactor.hpp:144-205).
//
// Random numbers are drawn on the host exactly as the reference draws them - std::mt19937{seed} through libstdc++'s
// uniform_real / uniform_int / normal distributions, in the same call order - so a given seed names the same stream.
// The device work of CKDE::sample is the instance selection: the reference materialises exp(logl_mat) (N x n), an
// inclusive prefix sum down every column, a division by the column total and a search for the bracket holding the
// uniform number.  Here: one weights-only sweep leaves (offset, sum w) per (training split, query); a per-query
// scan over the splits finds the split holding the target mass; a second per-query pass walks that split's rows
// in order.  No N x n matrix exists.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <random>

#include "common.hpp"
#include "kde_handle.hpp"
#include "stats_kernels.hpp"

using namespace pbn;

namespace {

// per query: the split whose cumulative weight first exceeds rn * total, and the mass still to cover inside it
__global__ __launch_bounds__(256) void pick_locate_kernel(const double* __restrict__ part, int nsplit, int64_t nqtiles, int64_t nq,
                                                          const double* __restrict__ rn, int32_t* __restrict__ split_out,
                                                          double* __restrict__ resid_out, double* __restrict__ m_out) {
    const int64_t q = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (q >= nq) return;
    const double* p = part + q * 4;
    const int64_t stride = nqtiles * 16 * 4;
    double M = p[0];
    for (int s = 1; s < nsplit; ++s) M = fmax(M, p[s * stride]);
    double total = 0.0;
    for (int s = 0; s < nsplit; ++s) total += p[s * stride + 1] * exp2(p[s * stride] - M);
    const double target = rn[q] * total;
    double cum = 0.0;
    int sel = nsplit - 1;
    for (int s = 0; s < nsplit; ++s) {
        const double c = p[s * stride + 1] * exp2(p[s * stride] - M);
        if (cum + c > target) { sel = s; break; }
        if (s + 1 < nsplit) cum += c;
    }
    split_out[q] = sel;
    resid_out[q] = target - cum;
    m_out[q] = M;
}

// per query: first row j of the chosen split with cum_j > resid (rows in training order); also log2 of the
// un-normalised weight of training row 0 (the reference leaves row 0 of its prefix sums un-normalised).
template <typename T>
__global__ __launch_bounds__(64) void pick_scan_kernel(const T* __restrict__ Ap, const T* __restrict__ Np, const T* __restrict__ Bp,
                                                       const T* __restrict__ NYp, int KS, int dm, int64_t N, int64_t ntiles,
                                                       int64_t tiles_per_split, int64_t nq, const int32_t* __restrict__ split,
                                                       const double* __restrict__ resid, const double* __restrict__ Moff,
                                                       int32_t* __restrict__ j_out, double* __restrict__ log2w0) {
    const int64_t q = (int64_t)blockIdx.x * 64 + threadIdx.x;
    if (q >= nq) return;
    const int64_t qt = q >> 4;
    const int qi = (int)(q & 15);
    double zq[16];   // the query's whitened coordinates, cached up to 16 dimensions (beyond: read from the fragments at every term)
    for (int i = 0; i < 16; ++i) zq[i] = i < dm ? (double)Bp[(qt * KS + (i >> 2)) * 64 + (i & 3) * 16 + qi] : 0.0;
    const double ny = (double)NYp[qt * 16 + qi];
    auto s2 = [&](int64_t r) {
        const int64_t t = r >> 4;
        const int idx = (int)(r & 15);
        int lg, i;
        if (sizeof(T) == 8) { lg = idx & 3; i = idx >> 2; } else { lg = idx >> 2; i = idx & 3; }
        double acc = (double)Np[t * 16 + lg * 4 + i] + ny;
        if (dm <= 16) {
            for (int k = 0; k < dm; ++k) acc = __builtin_fma((double)Ap[(t * KS + (k >> 2)) * 64 + (k & 3) * 16 + idx], zq[k], acc);
        } else {
            for (int k = 0; k < dm; ++k)
                acc = __builtin_fma((double)Ap[(t * KS + (k >> 2)) * 64 + (k & 3) * 16 + idx], (double)Bp[(qt * KS + (k >> 2)) * 64 + (k & 3) * 16 + qi], acc);
        }
        return acc;
    };
    log2w0[q] = s2(0);
    const int64_t t0 = (int64_t)split[q] * tiles_per_split;
    const int64_t t1 = (t0 + tiles_per_split < ntiles) ? t0 + tiles_per_split : ntiles;
    const int64_t r0 = t0 * 16, r1 = (t1 * 16 < N) ? t1 * 16 : N;
    const double M = Moff[q], target = resid[q];
    double cum = 0.0;
    int64_t j = r1 - 1;
    for (int64_t r = r0; r < r1; ++r) {
        cum += exp2(s2(r) - M);
        if (cum > target) { j = r; break; }
    }
    j_out[q] = (int32_t)j;
}

template <typename T>
void read_column(const pbn_table* t, int col, int64_t row0, int64_t n, T* host, hipStream_t st) {
    HIP_CHECK(hipMemcpyAsync(host, (const T*)t->col(col) + row0, (size_t)n * sizeof(T), hipMemcpyDeviceToHost, st));
}

template <typename T>
void ckde_sample_t(pbn_kde* k, int64_t n, int64_t stream_n, const pbn_table* ev, const int* ev_cols, uint32_t seed, T* out) {
    pbn_ctx* ctx = k->ctx;
    const KdeModel& m = k->m;
    const int d = m.d, p = d - 1;
    const int64_t N = m.N;
    const pbn_table* tr = k->train;
    std::mt19937 rng{seed};
    if (p == 0) {  // CKDE.hpp:294-316
        std::uniform_int_distribution<> uniform(0, (int)N - 1);
        // bandwidth(0,0) from the whitening: W = sqrt(log2 e) / sqrt(H00)
        const double h00 = 1.4426950408889634073599246810019 / (m.W[0] * m.W[0]);
        std::normal_distribution<T> normal(0, (T)std::sqrt(h00));
        std::vector<T> train((size_t)N);
        read_column<T>(tr, k->cols_fit[0], k->train_row0, N, train.data(), ctx->stream);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < n; ++i) {
            const int index = uniform(rng);
            out[i] = train[index] + normal(rng);
        }
        return;
    }
    // ---- CKDE.hpp:319-385 _sample_multivariate ---------------------------------------------------------
    if (!ev || !ev_cols) throw invalid_error("Evidence values not present for sampling.");
    check_cols(ev, ev_cols, p, "pbn_ckde_sample");
    if (ev->n_rows < n) throw invalid_error("pbn_ckde_sample: evidence table has fewer than n rows");
    if (ev->dtype != m.dtype) throw invalid_error("Data type of training and test datasets is different.");
    std::vector<double> rn((size_t)n);
    {
        std::uniform_real_distribution<T> uniform(0, 1);
        for (int64_t i = 0; i < n; ++i) rn[i] = (double)uniform(rng);
        for (int64_t i = n; i < stream_n; ++i) (void)uniform(rng);  // the reference draws all of them before the normals
    }
    // query fragments of the evidence rows: whitening order = caller's evidence order, W = leading p x p block
    const bool wide = k->cdf_wide;               // more than 16 evidence variables: fp64 fragments, runtime-sized kernels
    const size_t es = wide ? sizeof(double) : sizeof(T);
    const int KS = k->cdf_KS;
    const int64_t tps = std::min<int64_t>(m.ntiles, 64);
    const int64_t nsplit = ceil_div(m.ntiles, tps);
    int64_t chunk = std::max<int64_t>(16, (((int64_t)1 << 24) / nsplit) & ~(int64_t)15);
    chunk = std::min<int64_t>(chunk, ceil_div(n, 16) * 16);
    const int64_t cq_tiles = chunk / 16;
    ctx->scratch_q.reserve((size_t)cq_tiles * KS * 64 * es + (size_t)cq_tiles * 16 * es + 256);
    ctx->scratch_part.reserve((size_t)nsplit * chunk * 4 * sizeof(double));
    dev_buf<double> d_rn((size_t)n), d_resid((size_t)chunk), d_m((size_t)chunk), d_w0((size_t)n);
    dev_buf<int32_t> d_split((size_t)chunk), d_j((size_t)n);
    HIP_CHECK(hipMemcpyAsync(d_rn.p, rn.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    for (int64_t q0 = 0; q0 < n; q0 += chunk) {
        const int64_t nq = std::min<int64_t>(chunk, n - q0), nqtiles = ceil_div(nq, 16);
        char* qb = ctx->scratch_q.p;
        struct { void* pack; void* npack; } pa{qb, qb + (size_t)nqtiles * KS * 64 * es};
        if (wide) {
            std::vector<int> hc((size_t)p);
            for (int i = 0; i < p; ++i) hc[i] = ev_cols[m.perm[i] - 1];
            pbn::WidePackArgs wq{};
            // the evidence's whitening = the leading p x p block of the joint matrix (rows of stride d)
            pbn::kde_wide_pack_args(ctx, wq, ev, hc.data(), p, p, m.W.data(), d, m.mu.data(), nullptr);
            wq.rows = nullptr; wq.row0 = q0; wq.n0 = nq; wq.row1 = 0; wq.n = nq; wq.ntiles = nqtiles; wq.is_query = 1;
            wq.pack = (double*)pa.pack; wq.npack = (double*)pa.npack; wq.upack = nullptr;
            KernelTimer kt(ctx, PBN_K_PACK);
            pbn::launch_pack_wide(wq, ctx->stream);
        } else {
            PackArgs pq{};
            pq.base = ev->data; pq.ld = ev->ld; pq.d = p; pq.dm = p; pq.KS = KS;
            for (int i = 0; i < p; ++i) {
                pq.cols[i] = ev_cols[m.perm[i] - 1];
                pq.mu[i] = m.mu[i];
                for (int j = 0; j < p; ++j) pq.W[i * p + j] = m.W[(size_t)i * d + j];
            }
            pq.row0 = q0; pq.n0 = nq; pq.row1 = 0; pq.n = nq; pq.ntiles = nqtiles; pq.is_query = 1;
            pq.pack = pa.pack; pq.npack = pa.npack;
            KernelTimer kt(ctx, PBN_K_PACK);
            launch_pack_classic(pq, m.dtype, ctx->stream);
        }
        CdfArgs ca{};
        ca.Apack = k->cA.p; ca.nxpack = k->cN.p; ca.utrain = nullptr;
        ca.Bpack = pa.pack; ca.nypack = pa.npack; ca.uquery = nullptr;
        ca.ntiles = m.ntiles; ca.nqtiles = nqtiles; ca.tiles_per_split = tps;
        ca.part = (double*)ctx->scratch_part.p;
        { KernelTimer kt(ctx, PBN_K_SWEEP); launch_cdf(ca, wide ? PBN_F64 : m.dtype, KS, (int)nsplit, ctx->stream); }
        KernelTimer kt(ctx, PBN_K_FINISH);
        hipLaunchKernelGGL(pick_locate_kernel, dim3((unsigned)ceil_div(nq, 256)), dim3(256), 0, ctx->stream, ca.part, (int)nsplit,
                           nqtiles, nq, d_rn.p + q0, d_split.p, d_resid.p, d_m.p);
        if (wide)
            hipLaunchKernelGGL(pick_scan_kernel<double>, dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, ctx->stream, (const double*)k->cA.p,
                               (const double*)k->cN.p, (const double*)pa.pack, (const double*)pa.npack, KS, p, N, m.ntiles, tps, nq, d_split.p,
                               d_resid.p, d_m.p, d_j.p + q0, d_w0.p + q0);
        else
            hipLaunchKernelGGL(pick_scan_kernel<T>, dim3((unsigned)ceil_div(nq, 64)), dim3(64), 0, ctx->stream, (const T*)k->cA.p,
                               (const T*)k->cN.p, (const T*)pa.pack, (const T*)pa.npack, KS, p, N, m.ntiles, tps, nq, d_split.p,
                               d_resid.p, d_m.p, d_j.p + q0, d_w0.p + q0);
        HIP_CHECK(hipGetLastError());
    }
    std::vector<int32_t> jsel((size_t)n);
    std::vector<double> w0((size_t)n);
    HIP_CHECK(hipMemcpyAsync(jsel.data(), d_j.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(w0.data(), d_w0.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // The reference's bracket search (KDE.cl.src:351-374): prefix sums c[0..N-1], rows 1.. divided by the total, row 0
    // left as exp(logl) itself; index r is taken when c[r] <= rn < c[r+1]; no bracket leaves the default N-1.  With
    // j = first row whose normalised prefix sum exceeds rn that is r = j - 1, and r = 0 additionally needs the raw
    // c[0] <= rn.
    const double lognorm_marg = m.lognorm_marg;
    std::vector<int32_t> idx((size_t)n);
    for (int64_t i = 0; i < n; ++i) {
        const int64_t j = jsel[i];
        if (j >= 2) idx[i] = (int32_t)(j - 1);
        else idx[i] = ((double)(T)std::exp(w0[i] * 0.6931471805599453094 + lognorm_marg) <= rn[i]) ? 0 : (int32_t)(N - 1);
    }
    // gather the sampled training rows (all d columns, caller order) and read the evidence rows back
    dev_buf<int32_t> d_idx((size_t)n);
    dev_buf<T> d_rows((size_t)n * d);
    std::vector<int32_t> src((size_t)n);
    for (int64_t i = 0; i < n; ++i) src[i] = (int32_t)(k->train_row0 + idx[i]);
    HIP_CHECK(hipMemcpyAsync(d_idx.p, src.data(), (size_t)n * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    for (int c = 0; c < d; ++c)
        launch_take(tr->col(k->cols_fit[c]), 0, d_rows.p + (size_t)c * n, 0, d_idx.p, n, 1, m.dtype, ctx->stream);
    std::vector<T> rows((size_t)n * d), evh((size_t)n * p);
    HIP_CHECK(hipMemcpyAsync(rows.data(), d_rows.p, rows.size() * sizeof(T), hipMemcpyDeviceToHost, ctx->stream));
    for (int j = 0; j < p; ++j) read_column<T>(ev, ev_cols[j], 0, n, evh.data() + (size_t)j * n, ctx->stream);
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    // transform = H12 H22^-1 and cond_var from the last whitening row (see pbn_ckde_cdf): row = (-b, 1) / sigma_c * sqrt(log2 e)
    const double sc = std::sqrt(1.4426950408889634073599246810019);
    const double wlast = m.W[(size_t)(d - 1) * d + (d - 1)];
    const double sigma_c = sc / wlast;
    std::vector<T> transform(p);  // caller's evidence order
    for (int i = 0; i < p; ++i) transform[m.perm[i] - 1] = (T)(-m.W[(size_t)(d - 1) * d + i] / wlast);
    std::normal_distribution<T> normal(0, (T)sigma_c);
    for (int64_t i = 0; i < n; ++i) {
        T cm = 0;
        for (int j = 0; j < p; ++j) cm += (evh[(size_t)j * n + i] - rows[(size_t)(j + 1) * n + i]) * transform[j];
        cm += rows[i] + normal(rng);
        out[i] = cm;
    }
}

}  // namespace

extern "C" {

int pbn_ckde_sample(pbn_kde* k, int64_t n, int64_t stream_n, const pbn_table* evidence, const int* ev_cols, uint32_t seed,
                    void* out) {
    return guarded(mu_of(k), [&] {
        if (!k) throw invalid_error("CKDE factor not fitted.");
        if (!k->ckde || !k->train) throw invalid_error("pbn_ckde_sample: the handle was not created by pbn_ckde_fit");
        if (n < 0) throw invalid_error("n should be a non-negative number");
        if (n == 0) return;
        if (!out) throw invalid_error("pbn_ckde_sample: null output");
        if (n > INT32_MAX || k->m.N > INT32_MAX) throw invalid_error("pbn_ckde_sample: more than 2^31 rows");
        HIP_CHECK(hipSetDevice(k->ctx->device));
        if (k->m.dtype == PBN_F64) ckde_sample_t<double>(k, n, std::max(n, stream_n), evidence, ev_cols, seed, (double*)out);
        else ckde_sample_t<float>(k, n, std::max(n, stream_n), evidence, ev_cols, seed, (float*)out);
    });
}

// LinearGaussianCPD::sample (LinearGaussianCPD.cpp:317-380): host only.  evidence[j]: n values of ev_dtype (PBN_F64 /
// PBN_F32) for beta[j + 1]; out: n doubles.
int pbn_lg_sample(int64_t n, const double* beta, int p, double variance, uint32_t seed, const void* const* evidence,
                  int ev_dtype, double* out) {
    return guarded([&] {
        if (n < 0) throw invalid_error("n should be a non-negative number");
        if (!beta || (n > 0 && !out) || (p > 0 && !evidence)) throw invalid_error("pbn_lg_sample: null argument");
        std::mt19937 rng{seed};
        std::normal_distribution<> normal(beta[0], std::sqrt(variance));
        for (int64_t i = 0; i < n; ++i) out[i] = normal(rng);
        for (int j = 0; j < p; ++j) {
            if (!evidence[j]) throw invalid_error("Evidence values not present for sampling.");
            if (ev_dtype == PBN_F64) {
                const double* e = (const double*)evidence[j];
                for (int64_t i = 0; i < n; ++i) out[i] += beta[j + 1] * e[i];
            } else {
                const float* e = (const float*)evidence[j];
                for (int64_t i = 0; i < n; ++i) out[i] += beta[j + 1] * e[i];
            }
        }
    });
}

// DiscreteFactor::sample_indices (DiscreteFactor.hpp:144-205): logprob is the CPT in the factor's layout (variable
// fastest, card values per parent configuration), parent_offset[i] = offset of row i's configuration (NULL without
// evidence).  out: n category indices.
int pbn_discrete_sample(int64_t n, const double* logprob, int card, int64_t n_entries, const int32_t* parent_offset,
                        uint32_t seed, int32_t* out) {
    return guarded([&] {
        if (n < 0) throw invalid_error("n should be a non-negative number");
        if (!logprob || card < 1 || (n > 0 && !out)) throw invalid_error("pbn_discrete_sample: bad argument");
        std::vector<double> accum((size_t)n_entries, 0.0);
        for (int64_t off = 0; off + card <= n_entries; off += card) {
            accum[off] = std::exp(logprob[off]);
            for (int j = 1; j < card - 1; ++j) accum[off + j] = accum[off + j - 1] + std::exp(logprob[off + j]);
        }
        std::mt19937 rng{seed};
        std::uniform_real_distribution<> uniform(0, 1);
        for (int64_t i = 0; i < n; ++i) {
            const double r = uniform(rng);
            const int64_t off = parent_offset ? parent_offset[i] : 0;
            if (off < 0 || off + card > n_entries) throw invalid_error("pbn_discrete_sample: configuration out of range");
            int index = card - 1;
            for (int j = 0; j < card - 1; ++j)
                if (r < accum[off + j]) { index = j; break; }
            out[i] = index;
        }
    });
}

}  // extern "C"
