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
#include "kde_model.hpp"
#include "kde_handle.hpp"
#include "stats_kernels.hpp"

namespace pbn {
static thread_local std::string g_last_error;
void set_last_error(const std::string& s) { g_last_error = s; }
std::recursive_mutex& api_mutex() {
    static std::recursive_mutex m;
    return m;
}
}  // namespace pbn

using namespace pbn;

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

}  // extern "C"

// the last reference is gone: nobody can be inside the context any more
void pbn::ctx_release(pbn_ctx* ctx) {
    if (ctx->refs.fetch_sub(1, std::memory_order_acq_rel) != 1) return;
    PBN_API_LOCK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    ctx->scratch_part.release();
    ctx->scratch_q.release();
    ctx->scratch_misc.release();
    ctx->scratch_red.release();
    (void)hipStreamDestroy(ctx->stream);
    for (auto& ln : ctx->parked)
        if (ln.stream) { (void)hipStreamSynchronize(ln.stream); (void)hipStreamDestroy(ln.stream); (void)hipEventDestroy(ln.fence); }
    delete ctx;
}

extern "C" {

// Drops the creator's reference: the context goes now if no handle created on it is alive, with the last of them otherwise
// (common.hpp, "Lifetime").
void pbn_ctx_destroy(pbn_ctx* ctx) {
    if (ctx) pbn::ctx_release(ctx);
}

int pbn_ctx_sync(pbn_ctx* ctx) {
    return guarded(mu_of(ctx), [&] {
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
    return guarded(mu_of(ctx), [&] {
        if (!ctx) throw invalid_error("pbn_ctx_set_profiling: null context");
        HIP_CHECK(hipSetDevice(ctx->device));
        drain_timers(ctx);
        ctx->profiling = on == 1;   // 1: events + the score engine on ONE stream (per-kernel attribution under a profiler)
        ctx->timing = on == 2;      // 2: events only - issue lanes, batching and every other behaviour unchanged
        for (int i = 0; i < PBN_NUM_KERNEL_CLASSES; ++i) { ctx->kernel_ms[i] = 0; ctx->kernel_launches[i] = 0; }
    });
}

int pbn_ctx_kernel_time(pbn_ctx* ctx, int kernel_class, double* total_ms, int64_t* launches) {
    return guarded(mu_of(ctx), [&] {
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
    return guarded(mu_of(ctx), [&] {
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
    return guarded(mu_of(ctx), [&] {
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
    pbn::ctx_pin pin_(t->ctx);
    std::lock_guard<std::recursive_mutex> lock_(mu_of(t));
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
    return guarded(mu_of(t), [&] {
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
    return guarded(mu_of(t), [&] {
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
// Shifted Gram of up to 64 columns -> host means (d) and centred SSE (d x d, col-major).
static void sse_block(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* means, double* sse) {
    pbn_ctx* ctx = t->ctx;
    const int nct = (d + 15) / 16;
    const int WS = gram_ws(nct);
    int nblocks = (int)std::min<int64_t>(4 * ctx->num_cus, std::max<int64_t>(1, ceil_div(n, 256)));
    int64_t rpb = ceil_div(std::max<int64_t>(n, 1), nblocks);
    rpb = (rpb + 63) / 64 * 64;
    nblocks = (int)std::max<int64_t>(1, ceil_div(n, rpb));
    const size_t nsh = (size_t)t->n_cols;
    ctx->scratch_red.reserve((size_t)nblocks * WS + WS + nsh);
    double* partial = ctx->scratch_red.p;
    double* total = partial + (size_t)nblocks * WS;
    double* shift = total + WS;
    GramArgs a{};
    a.base = t->data; a.ld = t->ld; a.n_cols = d; a.row0 = row0; a.rows = nullptr; a.n = n;
    for (int i = 0; i < d; ++i) a.gc.cols[i] = cols[i];
    a.rows_per_block = rpb; a.shift = shift; a.partial = partial; a.num_cus = ctx->num_cus;
    launch_pilot(t->data, t->ld, a.gc, d, row0, nullptr, n, t->dtype, shift, ctx->stream);
    { KernelTimer kt(ctx, PBN_K_GRAM); launch_gram(a, t->dtype, nblocks, total, ctx->stream); }
    std::vector<double> h((size_t)WS + nsh);
    HIP_CHECK(hipMemcpyAsync(h.data(), total, ((size_t)WS + nsh) * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const double* hs = h.data() + WS;            // pilot shifts, indexed by table column
    const double* S = h.data() + (WS - nct * 16);  // shifted column sums
    const double N = (double)n;
    for (int i = 0; i < d; ++i) means[i] = hs[cols[i]] + (n > 0 ? S[i] / N : 0.0);
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
    return guarded(mu_of(t), [&] {
        check_cols(t, cols, d, "pbn_table_sse");
        check_range(t, row0, n, "pbn_table_sse");
        if (!means || !sse) throw invalid_error("pbn_table_sse: null output");
        if (d <= 0) return;
        HIP_CHECK(hipSetDevice(t->ctx->device));
        if (d <= 64) { sse_block(t, cols, d, row0, n, means, sse); return; }
        // more than 64 columns: the Gram kernels take 64 at a time - every pair of 32-column blocks (I <= J) is one call over I u J,
        // whose cross block (and, for I == J, diagonal block) lands in the result; an entry is the centred dot product of its two
        // columns whatever else shares the launch, up to the pilot shift's rounding
        const int B = 32, nb = (d + B - 1) / B;
        std::vector<int> sel;
        std::vector<double> mu2(2 * B), s2((size_t)4 * B * B);
        for (int I = 0; I < nb; ++I)
            for (int J = I; J < nb; ++J) {
                const int i0 = I * B, i1 = std::min(d, i0 + B), j0 = J * B, j1 = std::min(d, j0 + B);
                sel.assign(cols + i0, cols + i1);
                if (J != I) sel.insert(sel.end(), cols + j0, cols + j1);
                const int dd = (int)sel.size(), ni = i1 - i0;
                sse_block(t, sel.data(), dd, row0, n, mu2.data(), s2.data());
                if (J == I) {
                    for (int a = 0; a < ni; ++a) means[i0 + a] = mu2[a];   // a block's means come from ONE launch, its own (the pilot shift of a column is
                                                                          // a function of the column and the row range alone: every launch centres it alike)
                    for (int a = 0; a < ni; ++a)
                        for (int b = 0; b < ni; ++b) sse[(i0 + a) + (size_t)(i0 + b) * d] = s2[a + (size_t)b * dd];
                } else {
                    for (int a = 0; a < ni; ++a)
                        for (int b = 0; b < j1 - j0; ++b) {
                            const double v = s2[a + (size_t)(ni + b) * dd];
                            sse[(i0 + a) + (size_t)(j0 + b) * d] = v;
                            sse[(j0 + b) + (size_t)(i0 + a) * d] = v;
                        }
                }
            }
    });
}

// ---------------------------------------------------------------------------------------------------
int pbn_bandwidth(int selector, int kind, const double* cov, int d, int64_t n, int dtype, double* out) {
    return guarded([&] { bandwidth_from_cov(selector, kind, cov, d, n, dtype, out); });
}

// ---------------------------------------------------------------------------------------------------
static void kde_fit_impl(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                         const double* bw, int kind, bool cond, const double* center, pbn_kde** out, bool ckde = false) {
    if (!ctx || !out) throw invalid_error("pbn_kde_fit: null argument");
    check_cols(train, cols, d, "pbn_kde_fit");
    check_range(train, row0, n, "pbn_kde_fit");
    HIP_CHECK(hipSetDevice(ctx->device));
    auto k = std::make_unique<pbn_kde>();
    k->ctx = ctx;
    std::vector<double> pilot;
    if (!center && d > 0 && n > 0) {
        // centring offsets = pilot means (any offset is exact in the distances; it only keeps |z| small); 64 columns per launch
        ctx->scratch_red.reserve((size_t)train->n_cols + 8);
        for (int c0 = 0; c0 < d; c0 += 64) {
            GramCols sel{};
            const int dc = std::min(64, d - c0);
            for (int i = 0; i < dc; ++i) sel.cols[i] = cols[c0 + i];
            launch_pilot(train->data, train->ld, sel, dc, row0, nullptr, n, train->dtype, ctx->scratch_red.p, ctx->stream);
        }
        std::vector<double> all((size_t)train->n_cols);
        HIP_CHECK(hipMemcpyAsync(all.data(), ctx->scratch_red.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        pilot.resize(d);
        for (int i = 0; i < d; ++i) pilot[i] = all[cols[i]];
        center = pilot.data();
    }
    kde_prepare(k->m, train->dtype, d, n, bw, kind, cond, center);
    // fp32 table whose whitened rows reach too far from the centre for the fp32 Gram form (tiny bandwidths on spread-out data:
    // diagonal bandwidths of nearly collinear columns, user-set bandwidths): fp64 fragments + fp64 sweep on the float columns
    // (wide models - more than 32 dimensions - are packed into doubles anyway)
    if (train->dtype == PBN_F32 && n > 0 && !k->m.wide && kde_wants_widening(kde_max_norm2(ctx, k->m, train, cols, row0, n, 0), k->m.dm)) kde_widen(k->m);
    const int fdt = k->m.fdtype();
    // Low-dimensional CKDE on a large training set: two pruned plain sweeps (joint over [variable, evidence], marginal over
    // the evidence with H[1:, 1:] - CKDE.hpp:186-199) beat the fused sweep, whose pruning can only use the marginal box
    // (tools/prune_handles_timing.py; fp64 up to 3 variables - 4 is a tie, 5 goes to the fused sweep -, fp32 up to 4).  PBN_CKDE_SPLIT=0 keeps the fused sweep, =1 splits
    // whenever the marginal qualifies for pruning.
    static const int split_mode = PBN_TUNE(CKDE_SPLIT, -1);
    const bool split = ckde && k->m.cond && ((split_mode != 0 && kde_prune_applies(fdt, d - 1, n) && (split_mode > 0 || d <= (fdt == PBN_F64 ? 3 : 4))) ||
                                             k->m.wide);   // more than 32 evidence variables: no fused form - joint minus marginal
    if (split) {
        std::vector<double> Hm((size_t)(d - 1) * (d - 1));
        for (int j = 1; j < d; ++j)
            for (int i = 1; i < d; ++i) Hm[(i - 1) + (size_t)(j - 1) * (d - 1)] = bw[i + (size_t)j * d];
        pbn_kde* kj = nullptr;
        pbn_kde* km = nullptr;
        kde_fit_impl(ctx, train, cols, d, row0, n, bw, PBN_BW_FULL, false, center, &kj);
        k->split_joint.reset(kj);
        kde_fit_impl(ctx, train, cols + 1, d - 1, row0, n, Hm.data(), PBN_BW_FULL, false, center ? center + 1 : nullptr, &km);
        k->split_marg.reset(km);
    } else {
        const KdePackBytes pb = kde_pack_bytes(fdt, k->m.dm, k->m.cond, n);
        k->Apack.alloc(pb.apack);
        k->nxpack.alloc(pb.nxpack);
        if (k->m.cond) k->Axpack.alloc(pb.axpack);
        k->m.Apack = k->Apack.p; k->m.nxpack = k->nxpack.p; k->m.Axpack = k->Axpack.p;
        // low-dimensional, large models: rows packed in Morton order with per-tile boxes, so that logl / slogl skip the
        // tile pairs that cannot contribute (same rule as the score engine's sweeps)
        // (which dimensions: kde_prune_applies - up to 6 marginal dimensions; beyond, boxes over 4 of the dimensions prune too little)
        static const int max_dm = PBN_TUNE(HANDLE_PRUNE_DIMS, 8);
        const bool prune = k->m.dm <= max_dm;
        kde_pack_train(ctx, k->m, train, cols, row0, n, 0, nullptr, prune);
        kde_prune_persist(ctx, k->m, k->prune_store);
    }
    if (ckde) {
        // The joint Cholesky factor with the variable last has the Schur complement on its corner: the last
        // whitening row IS (x - H12 H22^-1 e) / sigma_c (CKDE.hpp:538-555 "transform" and "cond_var").
        const KdeModel& m = k->m;
        k->ckde = true;
        k->cols_fit.assign(cols, cols + d);
        k->train = train;
        k->train_row0 = row0;
        const double sc = std::sqrt(2.0 * 1.4426950408889634073599246810019);
        k->wu.assign((size_t)m.d, 0.0);
        for (int j = 0; j < m.d; ++j) k->wu[j] = m.W[(size_t)(m.d - 1) * m.d + j] / sc;
        if (m.d <= PBN_W_INLINE_D) {   // cdf / sample fragments: up to 16 evidence variables (logl / slogl go to 32 in fp64)
            k->cdf_KS = std::max(1, (m.d - 1 + 3) / 4);
            const size_t es = dtype_size(m.dtype);
            k->cA.alloc((size_t)m.ntiles * k->cdf_KS * 64 * es);
            k->cN.alloc((size_t)m.ntiles * 16 * es);
            k->cU.alloc((size_t)m.ntiles * 16 * es);
            PackArgs pa{};
            fill_cdf_pack(pa, *k, train, cols);
            pa.row0 = row0; pa.n0 = n; pa.row1 = 0; pa.n = n; pa.ntiles = m.ntiles; pa.is_query = 0;
            pa.pack = k->cA.p; pa.npack = k->cN.p; pa.upack = k->cU.p;
            KernelTimer kt(ctx, PBN_K_PACK);
            launch_pack_classic(pa, m.dtype, ctx->stream);
        } else {
            // more than 16 evidence variables (the reference's cdf / sample loop over any number: CKDE.hpp:289-735): fp64 fragments in
            // the classic order through the generic pack, runtime-sized kernels (kde_cdf_kernel<double, 0, ...>)
            k->cdf_wide = true;
            k->cdf_KS = (m.d - 1 + 3) / 4;
            k->cA.alloc((size_t)m.ntiles * k->cdf_KS * 64 * sizeof(double));
            k->cN.alloc((size_t)m.ntiles * 16 * sizeof(double));
            k->cU.alloc((size_t)m.ntiles * 16 * sizeof(double));
            pbn::WidePackArgs wa{};
            fill_cdf_pack_wide(ctx, wa, *k, train, cols);
            wa.rows = nullptr; wa.row0 = row0; wa.n0 = n; wa.row1 = 0; wa.n = n; wa.ntiles = m.ntiles; wa.is_query = 0;
            wa.pack = (double*)k->cA.p; wa.npack = (double*)k->cN.p; wa.upack = (double*)k->cU.p;
            KernelTimer kt(ctx, PBN_K_PACK);
            launch_pack_wide(wa, ctx->stream);
        }
    }
    *out = k.release();
}

int pbn_kde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                const double* bandwidth, int kind, const double* center, pbn_kde** out) {
    return guarded(mu_of(ctx), [&] {
        if (kind != PBN_BW_FULL && kind != PBN_BW_DIAG) throw invalid_error("pbn_kde_fit: unknown bandwidth kind");
        kde_fit_impl(ctx, train, cols, d, row0, n, bandwidth, kind, false, center, out);
    });
}

int pbn_ckde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                 const double* H, const double* center, pbn_kde** out) {
    return guarded(mu_of(ctx), [&] { kde_fit_impl(ctx, train, cols, d, row0, n, H, PBN_BW_FULL, true, center, out, true); });
}

void pbn_kde_destroy(pbn_kde* k) {
    if (!k) return;
    pbn::ctx_pin pin_(k->ctx);
    std::lock_guard<std::recursive_mutex> lock_(mu_of(k));
    (void)hipSetDevice(k->ctx->device);
    (void)hipStreamSynchronize(k->ctx->stream);
    delete k;
}

int64_t pbn_kde_num_instances(const pbn_kde* k) { return k ? k->m.N : 0; }
double pbn_kde_lognorm(const pbn_kde* k, int which) { return which ? k->m.lognorm_marg : k->m.lognorm; }

static void kde_eval_enqueue(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                             double* dev_logl, double* dev_sum, bool precise = false) {
    if (!k) throw invalid_error("KDE factor not fitted.");
    if (k->split_joint) {   // CKDE as joint - marginal from two plain (pruned) sweeps
        pbn_ctx* ctx = k->ctx;
        check_cols(test, cols, k->m.d, "pbn_kde_logl");
        const size_t per = dev_logl ? (size_t)std::max<int64_t>(n, 1) : 0;
        ctx->scratch_split.reserve(2 * per + 2);
        double* lj = dev_logl ? ctx->scratch_split.p : nullptr;
        double* lm = dev_logl ? ctx->scratch_split.p + per : nullptr;
        double* sj = dev_sum ? ctx->scratch_split.p + 2 * per : nullptr;
        double* sm = dev_sum ? sj + 1 : nullptr;
        // the two sweeps are independent: the marginal one goes to the second issue lane, so that the joint sweep's tail (the
        // last workgroups of a pruned sweep drain for 2-3 ms) runs under it; the difference waits for both on the device
        const bool lanes = pbn::score_lanes() > 1 && !ctx->profiling;
        if (lanes) { ctx->ensure_lanes(1); ctx->lanes_wait_for_stream(1); }
        pbn::kde_eval_enqueue(ctx, k->split_joint->m, test, cols, row0, n, lj, sj, nullptr, nullptr, precise);
        {
            pbn::LaneSwitch lane(ctx, lanes ? 1 : 0);
            pbn::kde_eval_enqueue(ctx, k->split_marg->m, test, cols + 1, row0, n, lm, sm, nullptr, nullptr, precise);
        }
        if (lanes) ctx->stream_waits_for_lane(0);
        if (dev_logl) launch_diff(dev_logl, lj, lm, n, ctx->stream);
        if (dev_sum) launch_diff(dev_sum, sj, sm, 1, ctx->stream);
        return;
    }
    pbn::kde_eval_enqueue(k->ctx, k->m, test, cols, row0, n, dev_logl, dev_sum, nullptr, nullptr, precise);
}

int pbn_kde_logl_dev(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* dev_out) {
    return guarded(mu_of(k), [&] {
        if (!dev_out && n > 0) throw invalid_error("pbn_kde_logl_dev: null output");
        kde_eval_enqueue(k, test, cols, row0, n, dev_out, nullptr);
    });
}

int pbn_kde_logl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out) {
    return guarded(mu_of(k), [&] {
        if (!out && n > 0) throw invalid_error("pbn_kde_logl: null output");
        if (!k) throw invalid_error("KDE factor not fitted.");
        dev_buf<double> tmp((size_t)std::max<int64_t>(n, 1));
        kde_eval_enqueue(k, test, cols, row0, n, tmp.p, nullptr);
        if (n > 0) HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, k->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
    });
}

int pbn_ckde_cdf(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out) {
    return guarded(mu_of(k), [&] {
        if (!k) throw invalid_error("CKDE factor not fitted.");
        if (!k->ckde) throw invalid_error("pbn_ckde_cdf: the handle was not created by pbn_ckde_fit");
        if (!out && n > 0) throw invalid_error("pbn_ckde_cdf: null output");
        pbn_ctx* ctx = k->ctx;
        const KdeModel& m = k->m;
        check_cols(test, cols, m.d, "pbn_ckde_cdf");
        check_range(test, row0, n, "pbn_ckde_cdf");
        if (test->dtype != m.dtype) throw invalid_error("Data type of training and test datasets is different.");
        if (test->ctx->device != ctx->device) throw invalid_error("pbn_ckde_cdf: test table lives on another device");
        HIP_CHECK(hipSetDevice(ctx->device));
        if (n == 0) return;
        const int fdt = k->cdf_wide ? PBN_F64 : m.dtype;   // type of the cdf fragments
        const size_t es = dtype_size(fdt);
        const int64_t nqtiles = ceil_div(n, 16);
        const size_t b_b = (size_t)nqtiles * k->cdf_KS * 64 * es, n_b = (size_t)nqtiles * 16 * es;
        ctx->scratch_q.reserve(b_b + 2 * n_b + 256);
        char* q = ctx->scratch_q.p;
        struct { void* pack; void* npack; void* upack; } pa{q, q + b_b, q + b_b + n_b};
        if (k->cdf_wide) {
            pbn::WidePackArgs wq{};
            fill_cdf_pack_wide(ctx, wq, *k, test, cols);
            wq.rows = nullptr; wq.row0 = row0; wq.n0 = n; wq.row1 = 0; wq.n = n; wq.ntiles = nqtiles; wq.is_query = 1;
            wq.pack = (double*)pa.pack; wq.npack = (double*)pa.npack; wq.upack = (double*)pa.upack;
            KernelTimer kt(ctx, PBN_K_PACK);
            launch_pack_wide(wq, ctx->stream);
        } else {
            PackArgs pq{};
            fill_cdf_pack(pq, *k, test, cols);
            pq.row0 = row0; pq.n0 = n; pq.row1 = 0; pq.n = n; pq.ntiles = nqtiles; pq.is_query = 1;
            pq.pack = pa.pack; pq.npack = pa.npack; pq.upack = pa.upack;
            KernelTimer kt(ctx, PBN_K_PACK);
            launch_pack_classic(pq, m.dtype, ctx->stream);
        }
        const int64_t qblocks = ceil_div(nqtiles, 8);
        int64_t nsplit = std::max<int64_t>(1, ceil_div((int64_t)ctx->num_cus * 16, qblocks));
        nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, m.ntiles / 16));
        const int64_t tps = ceil_div(m.ntiles, nsplit);
        nsplit = ceil_div(m.ntiles, tps);
        ctx->scratch_part.reserve((size_t)nsplit * nqtiles * 16 * 4 * sizeof(double));
        CdfArgs ca{};
        ca.Apack = k->cA.p; ca.nxpack = k->cN.p; ca.utrain = k->cU.p;
        ca.Bpack = pa.pack; ca.nypack = pa.npack; ca.uquery = pa.upack;
        ca.ntiles = m.ntiles; ca.nqtiles = nqtiles; ca.tiles_per_split = tps;
        ca.part = (double*)ctx->scratch_part.p;
        { KernelTimer kt(ctx, PBN_K_SWEEP); launch_cdf(ca, fdt, k->cdf_KS, (int)nsplit, ctx->stream); }
        dev_buf<double> tmp((size_t)n);
        { KernelTimer kt(ctx, PBN_K_FINISH); launch_cdf_finish(ca.part, (int)nsplit, nqtiles, n, tmp.p, ctx->stream); }
        HIP_CHECK(hipMemcpyAsync(out, tmp.p, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
    });
}

int pbn_kde_slogl_async(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* dev_out) {
    return guarded(mu_of(k), [&] {
        if (!dev_out) throw invalid_error("pbn_kde_slogl_async: null output");
        kde_eval_enqueue(k, test, cols, row0, n, nullptr, dev_out);
    });
}

int pbn_kde_slogl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out) {
    return guarded(mu_of(k), [&] {
        if (!out) throw invalid_error("pbn_kde_slogl: null output");
        if (!k) throw invalid_error("KDE factor not fitted.");
        k->ctx->scratch_red.reserve(8);
        double* dsum = k->ctx->scratch_red.p;
        kde_eval_enqueue(k, test, cols, row0, n, nullptr, dsum);
        HIP_CHECK(hipMemcpyAsync(out, dsum, sizeof(double), hipMemcpyDeviceToHost, k->ctx->stream));
        HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
        // a sum that is a cancellation to ~0: once more at the accuracy of the per-row path (kde_sum_needs_precision; fp64 fragments only -
        // the fp32 sweeps have their own, wider tolerance)
        if (k->m.fdtype() == PBN_F64 && kde_sum_needs_precision(*out, n)) {
            kde_eval_enqueue(k, test, cols, row0, n, nullptr, dsum, /*precise=*/true);
            HIP_CHECK(hipMemcpyAsync(out, dsum, sizeof(double), hipMemcpyDeviceToHost, k->ctx->stream));
            HIP_CHECK(hipStreamSynchronize(k->ctx->stream));
        }
    });
}

}  // extern "C"
