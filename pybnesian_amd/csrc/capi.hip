// C-ABI glue of libpbn_hip.so: contexts, device tables, column statistics, bandwidth selectors and the
// KDE / ProductKDE / CKDE entry points declared in include/pbn_hip.h.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <numeric>

#include "common.hpp"
#include "hostmath.hpp"
#include "kde_kernels.hpp"
#include "stats_kernels.hpp"

namespace pbn {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
}  // namespace pbn

using namespace pbn;

struct pbn_kde {
    pbn_ctx* ctx = nullptr;
    int dtype = PBN_F64;
    int d = 0;        // number of variables
    int dm = 0;       // dimensions in the main MFMA contraction (d, or d-1 for CKDE)
    int KS = 0;       // ceil(dm / 4)
    bool cond = false;
    int64_t N = 0;
    int64_t ntiles = 0;
    double lognorm = 0.0, lognorm_marg = 0.0;
    std::vector<int> perm;  // position in the caller's column list -> whitening order
    dev_buf<char> Apack, nxpack, Axpack;
    std::vector<double> W, mu;  // whitening matrix (d x d row-major lower) and centring offsets, whitening order
};

extern "C" {

const char* pbn_last_error(void) { return g_last_error.c_str(); }
const char* pbn_version(void) { return "pbn_hip 0.1 (gfx950)"; }

// ---------------------------------------------------------------------------------------------------
int pbn_ctx_create(int device, pbn_ctx** out) {
    return guarded([&] {
        if (!out) throw invalid_error("pbn_ctx_create: out is null");
        int count = 0;
        HIP_CHECK(hipGetDeviceCount(&count));
        if (device < 0 || device >= count) throw device_error("pbn_ctx_create: no such HIP device");
        HIP_CHECK(hipSetDevice(device));
        auto ctx = std::make_unique<pbn_ctx>();
        ctx->device = device;
        HIP_CHECK(hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking));
        hipDeviceProp_t prop;
        HIP_CHECK(hipGetDeviceProperties(&prop, device));
        ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
        *out = ctx.release();
    });
}

void pbn_ctx_destroy(pbn_ctx* ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->scratch_part.release();
    ctx->scratch_q.release();
    ctx->scratch_misc.release();
    ctx->scratch_red.release();
    (void)hipStreamDestroy(ctx->stream);
    delete ctx;
}

int pbn_ctx_sync(pbn_ctx* ctx) {
    return guarded([&] {
        HIP_CHECK(hipSetDevice(ctx->device));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

void* pbn_ctx_stream(pbn_ctx* ctx) { return (void*)ctx->stream; }

static void drain_timers(pbn_ctx* ctx) {
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (auto& t : ctx->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, t.e0, t.e1) == hipSuccess) {
            ctx->kernel_ms[t.which] += ms;
            ctx->kernel_launches[t.which] += 1;
        }
        (void)hipEventDestroy(t.e0);
        (void)hipEventDestroy(t.e1);
    }
    ctx->pending.clear();
}

int pbn_ctx_set_profiling(pbn_ctx* ctx, int on) {
    return guarded([&] {
        if (!ctx) throw invalid_error("pbn_ctx_set_profiling: null context");
        HIP_CHECK(hipSetDevice(ctx->device));
        drain_timers(ctx);
        ctx->profiling = on != 0;
        for (int i = 0; i < PBN_NUM_KERNEL_CLASSES; ++i) { ctx->kernel_ms[i] = 0; ctx->kernel_launches[i] = 0; }
    });
}

int pbn_ctx_kernel_time(pbn_ctx* ctx, int kernel_class, double* total_ms, int64_t* launches) {
    return guarded([&] {
        if (!ctx || kernel_class < 0 || kernel_class >= PBN_NUM_KERNEL_CLASSES) throw invalid_error("pbn_ctx_kernel_time: bad argument");
        HIP_CHECK(hipSetDevice(ctx->device));
        drain_timers(ctx);
        if (total_ms) *total_ms = ctx->kernel_ms[kernel_class];
        if (launches) *launches = ctx->kernel_launches[kernel_class];
    });
}

// ---------------------------------------------------------------------------------------------------
static inline bool bit_set(const uint8_t* bm, int64_t i) { return (bm[i >> 3] >> (i & 7)) & 1; }

int pbn_table_create(pbn_ctx* ctx, const void* const* cols, int n_cols, int64_t n_rows, int dtype,
                     const uint8_t* valid, int64_t valid_offset, pbn_table** out) {
    return guarded([&] {
        if (!ctx || !out || (n_cols > 0 && !cols)) throw invalid_error("pbn_table_create: null argument");
        if (dtype != PBN_F64 && dtype != PBN_F32) throw invalid_error("Wrong data type. [double] or [float] data is expected.");
        if (n_cols < 0 || n_rows < 0) throw invalid_error("pbn_table_create: negative size");
        HIP_CHECK(hipSetDevice(ctx->device));
        const size_t es = dtype_size(dtype);
        int64_t n_valid = n_rows;
        std::vector<int64_t> keep;
        if (valid) {
            keep.reserve(n_rows);
            for (int64_t i = 0; i < n_rows; ++i)
                if (bit_set(valid, valid_offset + i)) keep.push_back(i);
            n_valid = (int64_t)keep.size();
        }
        auto t = std::make_unique<pbn_table>();
        t->ctx = ctx; t->dtype = dtype; t->n_cols = n_cols; t->n_rows = n_valid;
        t->ld = std::max<int64_t>(64, (n_valid + 63) / 64 * 64);
        t->owns = true;
        if (n_cols > 0) HIP_CHECK(hipMalloc(&t->data, (size_t)t->ld * n_cols * es));
        std::vector<char> tmp;
        if (valid) tmp.resize((size_t)n_valid * es);
        for (int c = 0; c < n_cols; ++c) {
            const char* src = (const char*)cols[c];
            if (valid) {  // compact the null rows away, order preserved (dataset.hpp:92-106)
                if (es == 8) {
                    const double* s = (const double*)src; double* o = (double*)tmp.data();
                    for (int64_t i = 0; i < n_valid; ++i) o[i] = s[keep[i]];
                } else {
                    const float* s = (const float*)src; float* o = (float*)tmp.data();
                    for (int64_t i = 0; i < n_valid; ++i) o[i] = s[keep[i]];
                }
                src = tmp.data();
            }
            if (n_valid > 0)
                HIP_CHECK(hipMemcpy((char*)t->data + (size_t)c * t->ld * es, src, (size_t)n_valid * es, hipMemcpyHostToDevice));
        }
        *out = t.release();
    });
}

int pbn_table_from_device(pbn_ctx* ctx, void* dev_base, int64_t ld, int n_cols, int64_t n_rows, int dtype,
                          pbn_table** out) {
    return guarded([&] {
        if (!ctx || !out || !dev_base) throw invalid_error("pbn_table_from_device: null argument");
        if (dtype != PBN_F64 && dtype != PBN_F32) throw invalid_error("Wrong data type. [double] or [float] data is expected.");
        if (ld < n_rows) throw invalid_error("pbn_table_from_device: ld < n_rows");
        auto t = std::make_unique<pbn_table>();
        t->ctx = ctx; t->dtype = dtype; t->n_cols = n_cols; t->n_rows = n_rows; t->ld = ld;
        t->data = dev_base; t->owns = false;
        *out = t.release();
    });
}

void pbn_table_destroy(pbn_table* t) {
    if (!t) return;
    if (t->owns && t->data) {
        (void)hipSetDevice(t->ctx->device);
        (void)hipStreamSynchronize(t->ctx->stream);
        (void)hipFree(t->data);
    }
    delete t;
}

int64_t pbn_table_rows(const pbn_table* t) { return t ? t->n_rows : 0; }
int pbn_table_cols(const pbn_table* t) { return t ? t->n_cols : 0; }

int pbn_table_take(const pbn_table* t, const int32_t* rows, int64_t n, pbn_table** out) {
    return guarded([&] {
        if (!t || !out || (n > 0 && !rows)) throw invalid_error("pbn_table_take: null argument");
        for (int64_t i = 0; i < n; ++i)
            if (rows[i] < 0 || rows[i] >= t->n_rows) throw invalid_error("pbn_table_take: row index out of range");
        pbn_ctx* ctx = t->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        auto o = std::make_unique<pbn_table>();
        o->ctx = ctx; o->dtype = t->dtype; o->n_cols = t->n_cols; o->n_rows = n;
        o->ld = std::max<int64_t>(64, (n + 63) / 64 * 64);
        o->owns = true;
        const size_t es = dtype_size(t->dtype);
        if (t->n_cols > 0) HIP_CHECK(hipMalloc(&o->data, (size_t)o->ld * o->n_cols * es));
        dev_buf<int32_t> drows((size_t)std::max<int64_t>(n, 1));
        if (n > 0) HIP_CHECK(hipMemcpyAsync(drows.p, rows, (size_t)n * 4, hipMemcpyHostToDevice, ctx->stream));
        launch_take(t->data, t->ld, o->data, o->ld, drows.p, n, t->n_cols, t->dtype, ctx->stream);
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = o.release();
    });
}

int pbn_table_read(const pbn_table* t, const int* cols, int n_sel, void* outp) {
    return guarded([&] {
        if (!t || !cols || !outp) throw invalid_error("pbn_table_read: null argument");
        HIP_CHECK(hipSetDevice(t->ctx->device));
        HIP_CHECK(hipStreamSynchronize(t->ctx->stream));
        const size_t es = dtype_size(t->dtype);
        for (int c = 0; c < n_sel; ++c) {
            if (cols[c] < 0 || cols[c] >= t->n_cols) throw invalid_error("pbn_table_read: column out of range");
            if (t->n_rows > 0)
                HIP_CHECK(hipMemcpy((char*)outp + (size_t)c * t->n_rows * es, t->col(cols[c]), (size_t)t->n_rows * es,
                                    hipMemcpyDeviceToHost));
        }
    });
}

// ---------------------------------------------------------------------------------------------------
static void check_cols(const pbn_table* t, const int* cols, int d, const char* who) {
    if (!t || !cols) throw invalid_error(std::string(who) + ": null argument");
    for (int i = 0; i < d; ++i)
        if (cols[i] < 0 || cols[i] >= t->n_cols) throw invalid_error(std::string(who) + ": column index out of range");
}
static void check_range(const pbn_table* t, int64_t row0, int64_t n, const char* who) {
    if (row0 < 0 || n < 0 || row0 + n > t->n_rows) throw invalid_error(std::string(who) + ": row range out of bounds");
}

// Shifted Gram of up to 64 columns -> host means (d) and centred SSE (d x d, col-major).
static void sse_block(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* means, double* sse) {
    pbn_ctx* ctx = t->ctx;
    const int nct = (d + 15) / 16;
    const int WS = gram_ws(nct);
    int nblocks = (int)std::min<int64_t>(2 * ctx->num_cus, std::max<int64_t>(1, ceil_div(n, 256)));
    int64_t rpb = ceil_div(std::max<int64_t>(n, 1), nblocks);
    rpb = (rpb + 63) / 64 * 64;
    nblocks = (int)std::max<int64_t>(1, ceil_div(n, rpb));
    ctx->scratch_red.reserve((size_t)nblocks * WS + WS + 64);
    double* partial = ctx->scratch_red.p;
    double* total = partial + (size_t)nblocks * WS;
    double* shift = total + WS;
    GramArgs a{};
    a.base = t->data; a.ld = t->ld; a.n_cols = d; a.row0 = row0; a.rows = nullptr; a.n = n;
    for (int i = 0; i < d; ++i) a.gc.cols[i] = cols[i];
    a.rows_per_block = rpb; a.shift = shift; a.partial = partial;
    launch_pilot(t->data, t->ld, a.gc, d, row0, nullptr, n, t->dtype, shift, ctx->stream);
    { KernelTimer kt(ctx, PBN_K_GRAM); launch_gram(a, t->dtype, nblocks, total, ctx->stream); }
    std::vector<double> h((size_t)WS + 64);
    HIP_CHECK(hipMemcpyAsync(h.data(), total, ((size_t)WS + 64) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double* hs = h.data() + WS;            // pilot shifts
    const double* S = h.data() + (WS - nct * 16);  // shifted column sums
    const double N = (double)n;
    for (int i = 0; i < d; ++i) means[i] = hs[i] + (n > 0 ? S[i] / N : 0.0);
    int p = 0;
    for (int I = 0; I < nct; ++I)
        for (int J = I; J < nct; ++J, ++p) {
            const double* tile = h.data() + (size_t)p * 256;
            for (int e = 0; e < 256; ++e) {
                const int reg = e >> 6, lane = e & 63;
                const int r = I * 16 + (lane >> 4) + 4 * reg, c = J * 16 + (lane & 15);
                if (r >= d || c >= d) continue;
                const double v = tile[e] - (n > 0 ? S[r] * S[c] / N : 0.0);
                sse[r + (size_t)c * d] = v;
                if (I != J) sse[c + (size_t)r * d] = v;
            }
        }
    // diagonal tiles hold both triangles; symmetrise from the upper one for exact symmetry
    for (int c = 0; c < d; ++c)
        for (int r = 0; r < c; ++r) sse[c + (size_t)r * d] = sse[r + (size_t)c * d];
}

int pbn_table_sse(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* means, double* sse) {
    return guarded([&] {
        check_cols(t, cols, d, "pbn_table_sse");
        check_range(t, row0, n, "pbn_table_sse");
        if (!means || !sse) throw invalid_error("pbn_table_sse: null output");
        if (d <= 0) return;
        if (d > 64) throw invalid_error("pbn_table_sse: more than 64 columns per call is not supported yet");
        HIP_CHECK(hipSetDevice(t->ctx->device));
        sse_block(t, cols, d, row0, n, means, sse);
    });
}

// ---------------------------------------------------------------------------------------------------
int pbn_bandwidth(int selector, int kind, const double* cov, int d, int64_t n, int dtype, double* out) {
    return guarded([&] {
        if (!cov || !out || d <= 0) throw invalid_error("pbn_bandwidth: bad argument");
        const bool f32 = dtype == PBN_F32;
        const double N = (double)n, D = (double)d;
        auto not_enough = [&](const char* what) {
            throw singular_error(std::string(what) + " of " + std::to_string(d) + " variables cannot be estimated with " +
                                 std::to_string(n) + " instances");
        };
        if (selector == PBN_SEL_SCOTT) {
            // kde/ScottsBandwidth.hpp:66-117
            if (kind == PBN_BW_DIAG) {
                if (n <= 1) not_enough("Diagonal bandwidth matrix");
                const double k = std::pow(N, -2.0 / (D + 4.0));
                for (int i = 0; i < d; ++i) out[i] = k * cov[i + (size_t)i * d];
            } else {
                if (n <= d) not_enough("Bandwidth matrix");
                if (!hm::is_psd(cov, d, f32)) throw singular_error("Covariance matrix is not positive-definite.");
                const double k = std::pow(N, -2.0 / (D + 4.0));
                for (int i = 0; i < d * d; ++i) out[i] = k * cov[i];
            }
            return;
        }
        if (selector != PBN_SEL_NORMAL_REFERENCE) throw invalid_error("pbn_bandwidth: unknown selector");
        // kde/NormalReferenceRule.hpp:12-59 (pre-checks), :72-106 (diag), :109-134 (full)
        if (n <= d) not_enough(kind == PBN_BW_DIAG ? "Diagonal bandwidth matrix" : "Bandwidth matrix");
        if (!hm::is_psd(cov, d, f32)) throw singular_error("Covariance matrix is not positive-definite.");
        if (kind == PBN_BW_FULL) {
            const double k = std::pow(4.0 / (N * (D + 2.0)), 2.0 / (D + 4.0));
            for (int i = 0; i < d * d; ++i) out[i] = k * cov[i];
            return;
        }
        // Chacon & Duong (2018) eq. 3.4: delta = diag(cov)^-1 cov
        std::vector<double> delta((size_t)d * d), dinv((size_t)d * d), dd((size_t)d * d);
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) delta[i + (size_t)j * d] = cov[i + (size_t)j * d] / cov[i + (size_t)i * d];
        if (!hm::inverse(delta.data(), d, dinv.data())) throw singular_error("Covariance matrix is not positive-definite.");
        double tr = 0.0, tr2 = 0.0;
        for (int i = 0; i < d; ++i) tr += dinv[i + (size_t)i * d];
        for (int i = 0; i < d; ++i)
            for (int k2 = 0; k2 < d; ++k2) tr2 += dinv[i + (size_t)k2 * d] * dinv[k2 + (size_t)i * d];
        const double k = 4.0 * D * std::sqrt(hm::determinant(delta.data(), d)) / (2.0 * tr2 + tr * tr);
        const double f = std::pow(k / N, 2.0 / (D + 4.0));
        for (int i = 0; i < d; ++i) out[i] = f * cov[i + (size_t)i * d];
    });
}

// ---------------------------------------------------------------------------------------------------
static const double LOG2E = 1.4426950408889634073599246810019;
static const double LOG_2PI = 1.8378770664093454835606594728112;

static void fill_pack_common(PackArgs& pa, const pbn_table* t, const int* cols, const std::vector<int>& perm, int d,
                             int dm, int KS, int64_t row0, int64_t n, const pbn_kde* k) {
    pa.base = t->data; pa.ld = t->ld; pa.d = d; pa.dm = dm; pa.KS = KS;
    for (int i = 0; i < d; ++i) pa.cols[i] = cols[perm[i]];
    pa.row0 = row0; pa.n0 = n; pa.row1 = 0; pa.rows = nullptr; pa.n = n; pa.ntiles = ceil_div(n, 16);
    for (int i = 0; i < d * d; ++i) pa.W[i] = k->W[i];
    for (int i = 0; i < d; ++i) pa.mu[i] = k->mu[i];
}

static void kde_fit_impl(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                         const double* bw, int kind, bool cond, const double* center, pbn_kde** out) {
    if (!ctx || !out || !bw) throw invalid_error("pbn_kde_fit: null argument");
    check_cols(train, cols, d, "pbn_kde_fit");
    check_range(train, row0, n, "pbn_kde_fit");
    if (d <= 0) throw invalid_error("pbn_kde_fit: no variables");
    if (n <= 0) throw invalid_error("pbn_kde_fit: no training instances");
    if (cond && d < 2) cond = false;  // CKDE without evidence is a plain KDE (CKDE.hpp:232-241)
    const int dm = cond ? d - 1 : d;
    if (dm > 16) throw invalid_error("KDE with more than 16 (+1 conditional) variables is not supported");
    HIP_CHECK(hipSetDevice(ctx->device));

    auto k = std::make_unique<pbn_kde>();
    k->ctx = ctx; k->dtype = train->dtype; k->d = d; k->dm = dm; k->KS = (dm + 3) / 4; k->cond = cond;
    k->N = n; k->ntiles = ceil_div(n, 16);
    k->perm.resize(d);
    if (cond) {  // evidence first, variable last
        for (int i = 0; i < d - 1; ++i) k->perm[i] = i + 1;
        k->perm[d - 1] = 0;
    } else {
        std::iota(k->perm.begin(), k->perm.end(), 0);
    }

    // whitening matrix W (row-major lower) = sqrt(log2 e) * L^-1, with L = chol(P H P^T)
    std::vector<double> W((size_t)d * d, 0.0);
    const double sc = std::sqrt(LOG2E);
    double logdet_half = 0.0, logdet_half_marg = 0.0;  // sum log L_ii
    if (kind == PBN_BW_DIAG) {
        for (int i = 0; i < d; ++i) {
            const double h = bw[k->perm[i]];
            if (!(h > 0.0) || !std::isfinite(h)) throw singular_error("ProductKDE: bandwidth must be positive");
            W[(size_t)i * d + i] = sc / std::sqrt(h);
            logdet_half += 0.5 * std::log(h);
            if (i < dm) logdet_half_marg += 0.5 * std::log(h);
        }
    } else {
        std::vector<double> H((size_t)d * d), L((size_t)d * d), Li((size_t)d * d);
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) H[i + (size_t)j * d] = bw[k->perm[i] + (size_t)k->perm[j] * d];
        if (!hm::cholesky(H.data(), d, L.data())) throw singular_error("KDE: bandwidth matrix is not positive-definite");
        hm::lower_inverse(L.data(), d, Li.data());
        for (int i = 0; i < d; ++i) {
            for (int j = 0; j <= i; ++j) W[(size_t)i * d + j] = sc * Li[i + (size_t)j * d];
            logdet_half += std::log(L[i + (size_t)i * d]);
            if (i < dm) logdet_half_marg += std::log(L[i + (size_t)i * d]);
        }
    }
    // KDE.hpp:476-477 / ProductKDE.hpp:188-189
    k->lognorm = -logdet_half - 0.5 * d * LOG_2PI - std::log((double)n);
    k->lognorm_marg = -logdet_half_marg - 0.5 * dm * LOG_2PI - std::log((double)n);

    k->W = W;
    k->mu.assign(d, 0.0);
    if (center) {
        for (int i = 0; i < d; ++i) k->mu[i] = center[k->perm[i]];
    } else {
        // centring offsets = pilot means (any offset is exact in the distances; it only keeps |z| small)
        GramCols gc{};
        for (int i = 0; i < d; ++i) gc.cols[i] = cols[k->perm[i]];
        ctx->scratch_red.reserve(64);
        launch_pilot(train->data, train->ld, gc, d, row0, nullptr, n, train->dtype, ctx->scratch_red.p, ctx->stream);
        HIP_CHECK(hipMemcpyAsync(k->mu.data(), ctx->scratch_red.p, d * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    }

    const size_t es = dtype_size(train->dtype);
    k->Apack.alloc((size_t)k->ntiles * k->KS * 64 * es);
    k->nxpack.alloc((size_t)k->ntiles * 16 * es);
    if (cond) k->Axpack.alloc((size_t)k->ntiles * 64 * es);
    PackArgs pa{};
    fill_pack_common(pa, train, cols, k->perm, d, dm, k->KS, row0, n, k.get());
    pa.is_query = 0;
    pa.pack = k->Apack.p; pa.npack = k->nxpack.p; pa.xpack = cond ? k->Axpack.p : nullptr;
    { KernelTimer kt(ctx, PBN_K_PACK); launch_pack(pa, train->dtype, ctx->stream); }
    *out = k.release();
}

int pbn_kde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                const double* bandwidth, int kind, const double* center, pbn_kde** out) {
    return guarded([&] {
        if (kind != PBN_BW_FULL && kind != PBN_BW_DIAG) throw invalid_error("pbn_kde_fit: unknown bandwidth kind");
        kde_fit_impl(ctx, train, cols, d, row0, n, bandwidth, kind, false, center, out);
    });
}

int pbn_ckde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                 const double* H, const double* center, pbn_kde** out) {
    return guarded([&] { kde_fit_impl(ctx, train, cols, d, row0, n, H, PBN_BW_FULL, true, center, out); });
}

void pbn_kde_destroy(pbn_kde* k) {
    if (!k) return;
    (void)hipSetDevice(k->ctx->device);
    (void)hipStreamSynchronize(k->ctx->stream);
    delete k;
}

int64_t pbn_kde_num_instances(const pbn_kde* k) { return k ? k->N : 0; }
double pbn_kde_lognorm(const pbn_kde* k, int which) { return which ? k->lognorm_marg : k->lognorm; }

static int env_int(const char* name, int dflt) {
    const char* s = std::getenv(name);
    return s && *s ? std::atoi(s) : dflt;
}

// Enqueue pack(queries) -> sweep -> finish on the context stream.  dev_logl / dev_sum are nullable.
static void kde_eval_enqueue(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                             double* dev_logl, double* dev_sum) {
    if (!k) throw invalid_error("KDE factor not fitted.");
    check_cols(test, cols, k->d, "pbn_kde_logl");
    check_range(test, row0, n, "pbn_kde_logl");
    if (test->dtype != k->dtype) throw invalid_error("Data type of training and test datasets is different.");
    pbn_ctx* ctx = k->ctx;
    if (test->ctx->device != ctx->device) throw invalid_error("pbn_kde_logl: test table lives on another device");
    HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {
        if (dev_sum) HIP_CHECK(hipMemsetAsync(dev_sum, 0, sizeof(double), ctx->stream));
        return;
    }
    const size_t es = dtype_size(k->dtype);
    const int64_t nqtiles = ceil_div(n, 16);
    // query fragments in scratch: Bpack | nypack | Bxpack
    const size_t bpack_b = (size_t)nqtiles * k->KS * 64 * es, ny_b = (size_t)nqtiles * 16 * es,
                 bx_b = k->cond ? (size_t)nqtiles * 64 * es : 0;
    ctx->scratch_q.reserve(bpack_b + ny_b + bx_b + 256);
    char* q = ctx->scratch_q.p;
    PackArgs pa{};
    fill_pack_common(pa, test, cols, k->perm, k->d, k->dm, k->KS, row0, n, k);
    pa.is_query = 1;
    pa.pack = q; pa.npack = q + bpack_b; pa.xpack = k->cond ? q + bpack_b + ny_b : nullptr;
    { KernelTimer kt(ctx, PBN_K_PACK); launch_pack(pa, k->dtype, ctx->stream); }

    // split the training tiles so that the grid is a few waves deep on every CU
    const int64_t qblocks = ceil_div(nqtiles, 4 * sweep_qg(k->dtype, k->cond));
    const int64_t target = (int64_t)ctx->num_cus * env_int("PBN_SWEEP_BLOCKS_PER_CU", 24);
    int64_t nsplit = std::max<int64_t>(1, ceil_div(target, qblocks));
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, k->ntiles / env_int("PBN_SWEEP_MIN_TILES", 32)));
    nsplit = std::min<int64_t>(nsplit, 4096);
    const int64_t tps = ceil_div(k->ntiles, nsplit);
    nsplit = ceil_div(k->ntiles, tps);
    const int P = k->cond ? 4 : 2;
    ctx->scratch_part.reserve((size_t)nsplit * nqtiles * 16 * P * sizeof(double));
    SweepArgs sa{};
    sa.Apack = k->Apack.p; sa.nxpack = k->nxpack.p; sa.Axpack = k->Axpack.p;
    sa.Bpack = pa.pack; sa.nypack = pa.npack; sa.Bxpack = pa.xpack;
    sa.ntiles = k->ntiles; sa.nqtiles = nqtiles; sa.tiles_per_split = tps;
    sa.part = (double*)ctx->scratch_part.p;
    { KernelTimer kt(ctx, PBN_K_SWEEP); launch_sweep(sa, k->dtype, k->KS, k->cond, (int)nsplit, ctx->stream); }

    const int64_t nblocks = ceil_div(n, 256);
    ctx->scratch_misc.reserve((size_t)nblocks * sizeof(double));
    FinishArgs fa{};
    fa.part = sa.part; fa.nsplit = (int)nsplit; fa.nqtiles = nqtiles; fa.nq = n;
    fa.lognorm = k->lognorm; fa.lognorm_marg = k->lognorm_marg;
    fa.logl = dev_logl; fa.block_sums = dev_sum ? (double*)ctx->scratch_misc.p : nullptr;
    { KernelTimer kt(ctx, PBN_K_FINISH); launch_finish(fa, k->cond, dev_sum, ctx->stream); }
}

int pbn_kde_logl_dev(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* dev_out) {
    return guarded([&] {
        if (!dev_out && n > 0) throw invalid_error("pbn_kde_logl_dev: null output");
        kde_eval_enqueue(k, test, cols, row0, n, dev_out, nullptr);
    });
}

int pbn_kde_logl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out) {
    return guarded([&] {
        if (!out && n > 0) throw invalid_error("pbn_kde_logl: null output");
        if (!k) throw invalid_error("KDE factor not fitted.");
        dev_buf<double> tmp((size_t)std::max<int64_t>(n, 1));
        kde_eval_enqueue(k, test, cols, row0, n, tmp.p, nullptr);
        if (n > 0) HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, k->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
    });
}

int pbn_kde_slogl_async(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* dev_out) {
    return guarded([&] {
        if (!dev_out) throw invalid_error("pbn_kde_slogl_async: null output");
        kde_eval_enqueue(k, test, cols, row0, n, nullptr, dev_out);
    });
}

int pbn_kde_slogl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out) {
    return guarded([&] {
        if (!out) throw invalid_error("pbn_kde_slogl: null output");
        if (!k) throw invalid_error("KDE factor not fitted.");
        k->ctx->scratch_red.reserve(8);
        double* dsum = k->ctx->scratch_red.p;
        kde_eval_enqueue(k, test, cols, row0, n, nullptr, dsum);
        HIP_CHECK(hipMemcpyAsync(out, dsum, sizeof(double), hipMemcpyDeviceToHost, k->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
    });
}

}  // extern "C"
