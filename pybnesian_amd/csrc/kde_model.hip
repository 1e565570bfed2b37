// Host side of KDE / ProductKDE / CKDE fitting and evaluation, shared by the public handles and the score
// engine.  See kde_kernels.hip for the device side.
#include <cstdio>
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <numeric>

#include "hostmath.hpp"
#include "kde_kernels.hpp"
#include "kde_model.hpp"
#include "stats_kernels.hpp"

namespace pbn {

static const double LOG2E = 1.4426950408889634073599246810019;
static const double LOG_2PI = 1.8378770664093454835606594728112;

void check_cols(const pbn_table* t, const int* cols, int d, const char* who) {
    if (!t || !cols) throw invalid_error(std::string(who) + ": null argument");
    for (int i = 0; i < d; ++i)
        if (cols[i] < 0 || cols[i] >= t->n_cols) throw invalid_error(std::string(who) + ": column index out of range");
}
void check_range(const pbn_table* t, int64_t row0, int64_t n, const char* who) {
    if (row0 < 0 || n < 0 || row0 + n > t->n_rows) throw invalid_error(std::string(who) + ": row range out of bounds");
}

void bandwidth_from_cov(int selector, int kind, const double* cov, int d, int64_t n, int dtype, double* out) {
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
    std::vector<double> delta((size_t)d * d), dinv((size_t)d * d);
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
}

// The d x d block a FULL bandwidth matrix of `rule_d` >= d variables has over d of them, from THEIR covariance alone: both library
// selectors are H = k(N, rule_d) cov (kde/ScottsBandwidth.hpp:100-117, kde/NormalReferenceRule.hpp:109-134), so the block is
// k(N, rule_d) cov_sub - the very doubles of the sub-block of bandwidth_from_cov on the whole set.  Pre-checks on the block's own columns.
void bandwidth_full_block(int selector, const double* cov, int d, int rule_d, int64_t n, int dtype, double* out) {
    if (!cov || !out || d <= 0 || rule_d < d) throw invalid_error("pbn_bandwidth: bad argument");
    if (selector != PBN_SEL_SCOTT && selector != PBN_SEL_NORMAL_REFERENCE) throw invalid_error("pbn_bandwidth: unknown selector");
    if (n <= d)
        throw singular_error("Bandwidth matrix of " + std::to_string(d) + " variables cannot be estimated with " + std::to_string(n) + " instances");
    if (!hm::is_psd(cov, d, dtype == PBN_F32)) throw singular_error("Covariance matrix is not positive-definite.");
    const double N = (double)n, D = (double)rule_d;
    const double k = selector == PBN_SEL_SCOTT ? std::pow(N, -2.0 / (D + 4.0)) : std::pow(4.0 / (N * (D + 2.0)), 2.0 / (D + 4.0));
    for (int i = 0; i < d * d; ++i) out[i] = k * cov[i];
}

KdePackBytes kde_pack_bytes(int dtype, int dm, bool cond, int64_t n) {
    const size_t es = dtype_size(dtype);
    const int64_t ntiles = ceil_div(n, 16);
    if (use_f16x2(dtype))  // [ntiles][NB][64][8 f16]; the training norm lives inside the fragments
        return {(size_t)ntiles * f16x2_mfmas(dm) * 64 * 16, 64, cond ? (size_t)ntiles * 64 * 16 : 0};
    const int KS = (dm + 3) / 4;
    // norms [ntiles][16], then the weights 2^norm [ntiles][16] of the WMUL sweep
    // ... then one double per tile: sqrt(max -norm) of its rows (PackArgs::write_r)
    return {(size_t)ntiles * KS * 64 * es, (size_t)ntiles * 16 * es * 2 + (size_t)ntiles * 8, cond ? (size_t)ntiles * 64 * es : 0};
}

void kde_prepare(KdeModel& m, int dtype, int d, int64_t n, const double* bw, int kind, bool cond, const double* center) {
    if (!bw) throw invalid_error("pbn_kde_fit: null bandwidth");
    if (d <= 0) throw invalid_error("pbn_kde_fit: no variables");
    if (n <= 0) throw invalid_error("pbn_kde_fit: no training instances");
    if (cond && d < 2) cond = false;  // CKDE without evidence is a plain KDE (CKDE.hpp:232-241)
    const int dm = cond ? d - 1 : d;
    // up to 32 main dimensions: fp64 KS <= 8 MFMAs per tile pair, fp32 (f16x2 fragments, 6 slots per dimension + 3) <= 7; the
    // classic fp32 fragments (PBN_F32_F16X2=0, a measurement switch) stay at 16
    const int max_dm = (dtype == PBN_F64 || use_f16x2(dtype)) ? 32 : 16;
    m.dtype = dtype; m.widen = false; m.d = d; m.dm = dm; m.KS = use_f16x2(dtype) ? f16x2_mfmas(dm) : (dm + 3) / 4; m.cond = cond;
    // beyond the templated shapes: the generic runtime-sized pack / sweep in fp64 fragments (kde_kernels.hpp "wide"); a conditional
    // model of that size has no fused sweep - its owner evaluates joint - marginal (capi.hip: split handles; the score engine's terms
    // are plain anyway) and kde_pack_train refuses it
    m.wide = dm > max_dm;
    if (m.wide) { m.widen = dtype == PBN_F32; m.KS = (dm + 3) / 4; }
    m.perm.assign((size_t)d, 0);
    m.N = n; m.ntiles = ceil_div(n, 16);
    if (cond) {  // evidence first, variable last
        for (int i = 0; i < d - 1; ++i) m.perm[i] = i + 1;
        m.perm[d - 1] = 0;
    } else {
        for (int i = 0; i < d; ++i) m.perm[i] = i;
    }
    // whitening matrix W (row-major lower) = sqrt(log2 e) * L^-1, with L = chol(P H P^T)
    m.W.assign((size_t)d * d, 0.0);
    const double sc = std::sqrt(LOG2E);
    double logdet_half = 0.0, logdet_half_marg = 0.0;  // sum log L_ii
    if (kind == PBN_BW_DIAG) {
        for (int i = 0; i < d; ++i) {
            const double h = bw[m.perm[i]];
            if (!(h > 0.0) || !std::isfinite(h)) throw singular_error("ProductKDE: bandwidth must be positive");
            m.W[(size_t)i * d + i] = sc / std::sqrt(h);
            logdet_half += 0.5 * std::log(h);
            if (i < dm) logdet_half_marg += 0.5 * std::log(h);
        }
    } else {
        std::vector<double> H((size_t)d * d), L((size_t)d * d), Li((size_t)d * d);
        for (int j = 0; j < d; ++j)
            for (int i = 0; i < d; ++i) H[i + (size_t)j * d] = bw[m.perm[i] + (size_t)m.perm[j] * d];
        if (!hm::cholesky(H.data(), d, L.data())) throw singular_error("KDE: bandwidth matrix is not positive-definite");
        hm::lower_inverse(L.data(), d, Li.data());
        for (int i = 0; i < d; ++i) {
            for (int j = 0; j <= i; ++j) m.W[(size_t)i * d + j] = sc * Li[i + (size_t)j * d];
            logdet_half += std::log(L[i + (size_t)i * d]);
            if (i < dm) logdet_half_marg += std::log(L[i + (size_t)i * d]);
        }
    }
    // KDE.hpp:476-477 / ProductKDE.hpp:188-189
    m.lognorm = -logdet_half - 0.5 * d * LOG_2PI - std::log((double)n);
    m.lognorm_marg = -logdet_half_marg - 0.5 * dm * LOG_2PI - std::log((double)n);
    m.mu.assign(d, 0.0);
    if (center)
        for (int i = 0; i < d; ++i) m.mu[i] = center[m.perm[i]];
}

static void fill_pack_common(pbn_ctx* ctx, PackArgs& pa, const pbn_table* t, const int* cols, const KdeModel& m) {
    if (m.d > PBN_MAX_D) throw invalid_error("internal: a wide model reached the fixed-size pack arguments");
    pa.base = t->data; pa.ld = t->ld; pa.d = m.d; pa.dm = m.dm; pa.KS = m.KS;
    pa.src_f32 = (m.widen && m.dtype == PBN_F32) ? 1 : 0;
    for (int i = 0; i < m.d; ++i) pa.cols[i] = cols[m.perm[i]];
    pa.rows = nullptr;
    pa.Wdev = nullptr;
    if (m.d <= PBN_W_INLINE_D) {
        for (int i = 0; i < m.d * m.d; ++i) pa.W[i] = m.W[i];
    } else {   // too large for the kernel arguments: through the context's (lane's) scratch, in stream order behind its last reader
        ctx->scratch_w.reserve((size_t)PBN_MAX_D * PBN_MAX_D);
        HIP_CHECK(hipMemcpyAsync(ctx->scratch_w.p, m.W.data(), (size_t)m.d * m.d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        pa.Wdev = ctx->scratch_w.p;
    }
    for (int i = 0; i < m.d; ++i) pa.mu[i] = m.mu[i];
}


double kde_max_norm2(pbn_ctx* ctx, const KdeModel& m, const pbn_table* t, const int* cols, int64_t row0, int64_t n0, int64_t row1,
                     const int32_t* dev_rows) {
    PackArgs pa{};
    fill_pack_common(ctx, pa, t, cols, m);
    pa.rows = dev_rows;
    pa.row0 = row0; pa.n0 = n0; pa.row1 = row1; pa.n = m.N; pa.ntiles = m.ntiles;
    ctx->scratch_misc.reserve(sizeof(double));
    double* slot = (double*)ctx->scratch_misc.p;
    HIP_CHECK(hipMemsetAsync(slot, 0, sizeof(double), ctx->stream));
    launch_max_norm2(pa, t->dtype, slot, ctx->stream);
    double v = 0.0;
    HIP_CHECK(hipMemcpyAsync(&v, slot, sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return v;
}

bool kde_wants_widening(double max_norm2, int dm) {
    // (read per call: tests move it inside one process; inf switches the widening off)
    // four-product fragments: the f32 accumulation of the Gram form, 2^-24 |z|^2 on an exponent, against 5e-4 (the reference tests' fp32 tolerance per
    // logl); three-product fragments (8, 9, 16...20, ... dimensions): the dropped products a2 b2, <= 2^-22 |z|^2 in the worst alignment of the
    // rounding residuals and a tenth of that typically, against twice that - both limits scale with PBN_F32_WIDEN_AT
    const double at = knob_double("PBN_F32_WIDEN_AT", 5e-4);
    if (f16x2_spd(dm) == 4) return !(max_norm2 * 5.9604644775390625e-08 <= at);
    return !(max_norm2 * 2.384185791015625e-07 <= 2.0 * at);
}

void kde_widen(KdeModel& m) {
    if (m.dtype != PBN_F32 || m.widen) return;
    m.widen = true;
    m.KS = (m.dm + 3) / 4;   // classic fp64 fragments: 4 coordinates per MFMA
}

// Morton order of the logical rows described by `pa` (already filled): whitened rows -> keys -> stable sort.  Returns the
// carved buffers inside `arena`.
struct PruneSide {
    double* zrow;        // [n][zd] in logical order
    uint32_t* keys;      // [n] sorted
    int32_t* perm;       // [n] sorted position -> logical row
    char* rest;          // first free byte behind them (256-aligned)
};
static PruneSide prune_sort_side(pbn_ctx* ctx, dev_buf<char>& arena, const PackArgs& pa, int dtype, int zd, int kd, size_t extra_bytes) {
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t n = (size_t)pa.n;
    const size_t zb = al(n * zd * sizeof(double)), kb = al(n * sizeof(uint32_t)), ib = al(n * sizeof(int32_t));
    arena.reserve(zb + 2 * kb + 2 * ib + extra_bytes + 256);
    char* p = arena.p;
    PruneSide s{};
    s.zrow = (double*)p; p += zb;
    uint32_t* keys_unsorted = (uint32_t*)p; p += kb;
    s.keys = (uint32_t*)p; p += kb;
    int32_t* iota = (int32_t*)p; p += ib;
    s.perm = (int32_t*)p; p += ib;
    s.rest = p;
    launch_prune_keys(pa, dtype, zd, kd, s.zrow, keys_unsorted, iota, ctx->stream);
    sort_keys(ctx->scratch_sort, keys_unsorted, s.keys, iota, s.perm, pa.n, prune_key_bits(kd) * kd, ctx->stream);
    return s;
}

// Up to 6 marginal dimensions (PBN_PRUNE_MAX_DIMS overrides): at 1e6 x 1e5 rows (tools/prune_dims67.py, boxes over 4 dimensions)
// d = 6 gains 10 % in fp64 and 9-13 % in fp32 on correlated and on independent normal data and is even on heavy-tailed data;
// d = 7 is even at best (heavy-tailed: +3...5 %), d = 8 loses 10 %.
bool kde_prune_applies(int dtype, int dm, int64_t n) {
    const int max_dims = PBN_TUNE(PRUNE_MAX_DIMS, 6);
    return knob_int("PBN_SWEEP_PRUNE", 1) && (dtype == PBN_F64 || use_f16x2(dtype)) && dm <= max_dims && n >= knob_int("PBN_PRUNE_MIN_ROWS", 32768);
}

// bytes of the subsample packs (kde_pack_bytes of nsub rows, each part 256-aligned)
struct SubBytes { size_t a, n, x, total; };
static SubBytes sub_bytes(const KdeModel& m, int64_t nsub) {
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const KdePackBytes pb = kde_pack_bytes(m.fdtype(), m.dm, m.cond, nsub);
    SubBytes b{al(pb.apack), al(pb.nxpack), al(pb.axpack), 0};
    b.total = b.a + b.n + b.x;
    return b;
}

// wide models: columns, centring offsets and whitening matrix through the context's (lane's) scratch, in stream order
void kde_wide_pack_args(pbn_ctx* ctx, WidePackArgs& wa, const pbn_table* t, const int* cols_whitened, int d_, int dm, const double* W, int ldw,
                        const double* mu, const double* wu) {
    const size_t d = (size_t)d_, wn = (size_t)d_ * (size_t)ldw;
    ctx->scratch_w.reserve(wn + 2 * d + d / 2 + 8);   // doubles: W | mu | wu | cols (ints)
    double* Wd = ctx->scratch_w.p;
    double* mud = Wd + wn;
    double* wud = mud + d;
    int* colsd = (int*)(wud + d);
    HIP_CHECK(hipMemcpyAsync(Wd, W, wn * sizeof(double), hipMemcpyHostToDevice, ctx->stream));   // pageable sources: staged before the call returns
    HIP_CHECK(hipMemcpyAsync(mud, mu, d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (wu) HIP_CHECK(hipMemcpyAsync(wud, wu, d * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(colsd, cols_whitened, d * sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    wa.base = t->data; wa.ld = t->ld; wa.cols = colsd; wa.mu = mud; wa.W = Wd; wa.wu = wu ? wud : nullptr;
    wa.d = d_; wa.dm = dm; wa.ldw = ldw; wa.KS = (dm + 3) / 4; wa.src_f32 = t->dtype == PBN_F32 ? 1 : 0;
}

static void fill_wide_pack(pbn_ctx* ctx, WidePackArgs& wa, const pbn_table* t, const int* cols, const KdeModel& m) {
    std::vector<int> hc((size_t)m.d);
    for (int i = 0; i < m.d; ++i) hc[i] = cols[m.perm[i]];
    kde_wide_pack_args(ctx, wa, t, hc.data(), m.d, m.d, m.W.data(), m.d, m.mu.data(), nullptr);
    wa.KS = m.KS;
}

void kde_pack_train(pbn_ctx* ctx, KdeModel& m, const pbn_table* t, const int* cols, int64_t row0, int64_t n0,
                    int64_t row1, const int32_t* dev_rows, bool prune, double* dev_max_norm2) {
    if (m.wide) {
        if (m.cond) throw invalid_error("KDE: a conditional model of more than 32 evidence variables is evaluated as joint - marginal");
        WidePackArgs wa{};
        fill_wide_pack(ctx, wa, t, cols, m);
        wa.rows = dev_rows; wa.row0 = row0; wa.n0 = n0; wa.row1 = row1; wa.n = m.N; wa.ntiles = m.ntiles; wa.is_query = 0;
        wa.pack = (double*)m.Apack; wa.npack = (double*)m.nxpack;
        m.prune = false;
        KernelTimer kt(ctx, PBN_K_PACK);
        launch_pack_wide(wa, ctx->stream);
        return;
    }
    PackArgs pa{};
    fill_pack_common(ctx, pa, t, cols, m);
    pa.rows = dev_rows;
    pa.row0 = row0; pa.n0 = n0; pa.row1 = row1; pa.n = m.N; pa.ntiles = m.ntiles;
    pa.is_query = 0;
    if (dev_max_norm2 && m.dtype == PBN_F32 && !m.widen) launch_max_norm2(pa, t->dtype, dev_max_norm2, ctx->stream);
    m.prune = false;
    if (prune && kde_prune_applies(m.fdtype(), m.dm, m.N)) {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        m.zdims = m.d;
        m.pdims = std::min(m.dm, std::min(PBN_TUNE(PRUNE_BOX_DIMS, 4), PBN_PRUNE_PD));   // dimensions of the Morton keys and the boxes (<= PBN_PRUNE_PD)
        const size_t box_b = al((size_t)m.ntiles * 2 * m.pdims * sizeof(double)), zs_b = al((size_t)m.N * m.zdims * sizeof(double));
        // more dimensions than the keys cover: a stratified subsample (every N / nsub-th row of the sorted order, <= 4096
        // rows, <= 1/64 of the set) is packed as well; the queries are swept against it first (kde_eval_enqueue)
        m.nsub = 0;
        if (m.d > m.pdims && PBN_TUNE(PRUNE_SUBSAMPLE, 1)) m.nsub = std::min<int64_t>(4096, m.N / 64) / 16 * 16;
        const SubBytes sb = sub_bytes(m, m.nsub);
        const PruneSide s = prune_sort_side(ctx, ctx->scratch_prune, pa, m.fdtype(), m.zdims, m.pdims, box_b + zs_b + (m.nsub ? sb.total : 0));
        double* box = (double*)s.rest;
        double* zsorted = (double*)(s.rest + box_b);
        launch_tile_boxes(s.zrow, s.perm, m.N, m.zdims, m.pdims, box, zsorted, ctx->stream);
        pa.perm = s.perm;
        m.prune = true;
        m.tile_box = box; m.zsorted = zsorted; m.keys_sorted = s.keys;
        if (m.nsub) {
            char* sp = s.rest + box_b + zs_b;
            m.ntiles_sub = m.nsub / 16;
            m.Asub = sp; m.nxsub = sp + sb.a; m.Axsub = m.cond ? sp + sb.a + sb.n : nullptr;
        }
    }
    pa.fold_norm = 1;   // harmless for consumers whose query pack leaves the slot 0
    pa.write_w = !use_f16x2(m.fdtype());
    pa.write_r = pa.write_w && m.fdtype() == PBN_F64;
    m.tile_r = pa.write_r != 0;
    pa.pack = m.Apack; pa.npack = m.nxpack; pa.xpack = m.cond ? m.Axpack : nullptr;
    KernelTimer kt(ctx, PBN_K_PACK);
    launch_pack(pa, m.fdtype(), ctx->stream);
    if (m.prune && m.nsub) {
        PackArgs ps = pa;
        ps.perm_stride = m.N / m.nsub;
        ps.n = m.nsub; ps.ntiles = m.ntiles_sub;
        ps.pack = m.Asub; ps.npack = m.nxsub; ps.xpack = m.Axsub;
        launch_pack(ps, m.fdtype(), ctx->stream);
    }
}

// A fitted handle outlives the context's arenas: move the pruning tables of a freshly packed model into `store`.
void kde_prune_persist(pbn_ctx* ctx, KdeModel& m, dev_buf<char>& store) {
    if (!m.prune) return;
    auto al = [](size_t x) { return (x + 255) / 256 * 256; };
    const size_t box_b = al((size_t)m.ntiles * 2 * m.pdims * sizeof(double)), zs_b = al((size_t)m.N * m.zdims * sizeof(double)),
                 key_b = al((size_t)m.N * sizeof(uint32_t));
    const SubBytes sb = sub_bytes(m, m.nsub);
    store.alloc(box_b + zs_b + key_b + (m.nsub ? sb.total : 0));
    HIP_CHECK(hipMemcpyAsync(store.p, m.tile_box, box_b, hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(store.p + box_b, m.zsorted, (size_t)m.N * m.zdims * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(store.p + box_b + zs_b, m.keys_sorted, (size_t)m.N * sizeof(uint32_t), hipMemcpyDeviceToDevice, ctx->stream));
    m.tile_box = (const double*)store.p;
    m.zsorted = (const double*)(store.p + box_b);
    m.keys_sorted = (const uint32_t*)(store.p + box_b + zs_b);
    if (m.nsub) {   // the three subsample packs are contiguous in the arena (kde_pack_train)
        char* sp = store.p + box_b + zs_b + key_b;
        HIP_CHECK(hipMemcpyAsync(sp, m.Asub, sb.total, hipMemcpyDeviceToDevice, ctx->stream));
        m.Asub = sp; m.nxsub = sp + sb.a; m.Axsub = m.cond ? sp + sb.a + sb.n : nullptr;
    }
}

void kde_eval_enqueue(pbn_ctx* ctx, const KdeModel& m, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                      double* dev_logl, double* dev_sum, const int32_t* dev_rows, double* dev_sum_marg, bool precise) {
    check_cols(test, cols, m.d, "pbn_kde_logl");
    const bool sum_only = dev_logl == nullptr && !precise;   // only sums leave this call, at the sum-only error budget
    if (!dev_rows) check_range(test, row0, n, "pbn_kde_logl");
    if (test->dtype != m.dtype) throw invalid_error("Data type of training and test datasets is different.");
    const int fdt = m.fdtype();   // type of the fragments and of the sweep (double for a widened fp32 model)
    if (m.wide) {   // more than 32 dimensions: generic pack + sweep, plain, fp64 fragments
        if (m.cond) throw invalid_error("KDE: a conditional model of more than 32 evidence variables is evaluated as joint - marginal");
        HIP_CHECK(hipSetDevice(ctx->device));
        if (n == 0) {
            if (dev_sum) HIP_CHECK(hipMemsetAsync(dev_sum, 0, sizeof(double), ctx->stream));
            return;
        }
        const int64_t nqt = ceil_div(n, 16);
        const size_t bp = (size_t)nqt * m.KS * 64 * sizeof(double), nyb = (size_t)nqt * 16 * sizeof(double);
        ctx->scratch_q.reserve(bp + nyb + 256);
        WidePackArgs wq{};
        fill_wide_pack(ctx, wq, test, cols, m);
        wq.rows = dev_rows; wq.row0 = row0; wq.n0 = n; wq.row1 = 0; wq.n = n; wq.ntiles = nqt; wq.is_query = 1;
        wq.pack = (double*)ctx->scratch_q.p; wq.npack = (double*)(ctx->scratch_q.p + bp);
        { KernelTimer kt(ctx, PBN_K_PACK); launch_pack_wide(wq, ctx->stream); }
        int64_t ns = std::max<int64_t>(1, ceil_div((int64_t)ctx->num_cus * 8, ceil_div(nqt, 4)));
        ns = std::min<int64_t>(ns, std::max<int64_t>(1, m.ntiles / 16));
        ns = std::min<int64_t>(ns, 1024);
        const int64_t tpsw = ceil_div(m.ntiles, ns);
        ns = ceil_div(m.ntiles, tpsw);
        ctx->scratch_part.reserve((size_t)ns * nqt * 16 * 2 * sizeof(double));
        SweepArgs sw{};
        sw.Apack = m.Apack; sw.nxpack = m.nxpack; sw.Bpack = wq.pack; sw.nypack = wq.npack;
        sw.ntiles = m.ntiles; sw.nqtiles = nqt; sw.tiles_per_split = tpsw; sw.part = (double*)ctx->scratch_part.p;
        { KernelTimer kt(ctx, PBN_K_SWEEP); launch_sweep_wide(sw, m.KS, (int)ns, ctx->stream); }
        const int64_t nb = ceil_div(n, 256);
        ctx->scratch_misc.reserve((size_t)nb * 2 * sizeof(double));
        FinishArgs fw{};
        fw.part = sw.part; fw.nsplit = (int)ns; fw.nqtiles = nqt; fw.nq = n;
        fw.lognorm = m.lognorm; fw.lognorm_marg = m.lognorm_marg;
        fw.logl = dev_logl; fw.scatter = nullptr; fw.block_sums = dev_sum ? (double*)ctx->scratch_misc.p : nullptr;
        { KernelTimer kt(ctx, PBN_K_FINISH); launch_finish(fw, false, dev_sum, ctx->stream, nullptr); }
        return;
    }
    if (test->ctx->device != ctx->device) throw invalid_error("pbn_kde_logl: test table lives on another device");
    HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) {
        if (dev_sum) HIP_CHECK(hipMemsetAsync(dev_sum, 0, sizeof(double), ctx->stream));
        if (dev_sum_marg) HIP_CHECK(hipMemsetAsync(dev_sum_marg, 0, sizeof(double), ctx->stream));
        return;
    }
    const size_t es = dtype_size(fdt);
    const int64_t nqtiles = ceil_div(n, 16);
    // query fragments in scratch: Bpack | nypack | Bxpack
    const bool b3 = use_f16x2(fdt);
    const size_t bpack_b = b3 ? (size_t)nqtiles * m.KS * 64 * 16 : (size_t)nqtiles * m.KS * 64 * es,
                 ny_b = (size_t)nqtiles * 16 * es,
                 bx_b = m.cond ? (b3 ? (size_t)nqtiles * 64 * 16 : (size_t)nqtiles * 64 * es) : 0,
                 xn_b = (m.cond && b3) ? (size_t)nqtiles * 16 * 4 : 0,
                 ff_b = b3 ? ((size_t)nqtiles * 16 + 255) / 256 * 256 : 0;   // f16x2 fragments: one flag per query row (beyond the f16 range)
    ctx->scratch_q.reserve(bpack_b + ny_b + bx_b + xn_b + ff_b + 512);
    char* q = ctx->scratch_q.p;
    PackArgs pa{};
    fill_pack_common(ctx, pa, test, cols, m);
    pa.rows = dev_rows;
    pa.row0 = row0; pa.n0 = n; pa.row1 = 0; pa.n = n; pa.ntiles = nqtiles;
    pa.is_query = 1;
    const bool fold = sweep_folds_norm(fdt, m.cond, m.KS, m.dm);
    pa.fold_norm = fold ? 1 : 0;
    const double* qbox = nullptr;
    const double* qthr = nullptr;
    const double* qlb = nullptr;
    const int32_t* qperm = nullptr;
    PruneSide qs{};
    if (m.prune) {
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        const size_t qbox_b = al((size_t)nqtiles * 2 * m.pdims * sizeof(double)), qthr_b = al((size_t)nqtiles * sizeof(double));
        const size_t qlb_b = al((size_t)nqtiles * 16 * sizeof(double));
        qs = prune_sort_side(ctx, ctx->scratch_pruneq, pa, fdt, m.zdims, m.pdims, qbox_b + qthr_b + qlb_b);
        pa.perm = qs.perm;
        qperm = qs.perm;
        qbox = (double*)qs.rest; qthr = (double*)(qs.rest + qbox_b);
        static const bool use_qlb = PBN_TUNE(SWEEP_QLB, 1) != 0;   // offsets of the pruned plain sweeps from the prepass bounds
        if (use_qlb) qlb = (double*)(qs.rest + qbox_b + qthr_b);
    }
    pa.pack = q; pa.npack = q + bpack_b; pa.xpack = m.cond ? q + bpack_b + ny_b : nullptr;
    pa.xnorm = xn_b ? q + bpack_b + ny_b + bx_b : nullptr;
    pa.far_flag = ff_b ? (unsigned char*)(q + ((bpack_b + ny_b + bx_b + xn_b + 255) / 256) * 256) : nullptr;
    { KernelTimer kt(ctx, PBN_K_PACK); launch_pack(pa, fdt, ctx->stream); }
    const int P = m.cond ? 4 : 2;
    const bool wmul = !fold && sweep_weights_norm(fdt, m.cond, m.KS, m.dm);
    if (m.prune) {
        // bounds of the queries' largest exponents: neighbours in Morton order, and (more dimensions than keys) one sweep over
        // the subsample of the training rows; then per query tile the smallest bound and the box
        const double* subpart = nullptr;
        if (m.nsub) {
            ctx->scratch_part.reserve((size_t)nqtiles * 16 * P * sizeof(double));
            SweepArgs ss{};
            ss.Apack = m.Asub; ss.nxpack = m.nxsub; ss.Axpack = m.Axsub;
            ss.Bpack = pa.pack; ss.nypack = pa.npack; ss.Bxpack = pa.xpack; ss.Bxnorm = pa.xnorm;
            ss.ntiles = m.ntiles_sub; ss.nqtiles = nqtiles; ss.tiles_per_split = m.ntiles_sub;
            ss.fold = fold ? 1 : 0; ss.wmul = wmul ? 1 : 0; ss.prune = 0;
            ss.part = (double*)ctx->scratch_part.p;
            { KernelTimer kt(ctx, PBN_K_PACK); launch_sweep(ss, fdt, m.KS, m.cond, 1, ctx->stream); }
            subpart = ss.part;
        }
        launch_query_prepass(qs.zrow, qs.perm, n, qs.keys, m.zsorted, m.keys_sorted, m.N, m.zdims, m.pdims, (double*)qbox, (double*)qthr, (double*)qlb,
                             ctx->stream, subpart, P, m.cond ? 2 : 0, subpart ? std::log2((double)m.nsub) : 0.0, m.cond ? nullptr : m.tile_box);
    }

    // split the training tiles so that the grid is a few waves deep on every CU
    const int64_t qblocks = ceil_div(nqtiles, 4 * sweep_qg(fdt, m.cond, m.KS, m.prune));
    const int64_t target = (int64_t)ctx->num_cus * PBN_TUNE(SWEEP_BLOCKS_PER_CU, 24);
    int64_t nsplit = std::max<int64_t>(1, ceil_div(target, qblocks));
    // with the XCD-aware block order (xcd_block) the blocks resident on one XCD share a split: keep a split's training
    // fragments within half of the 4 MB L2, and the number of splits a multiple of 8 so that the XCDs get equal shares
    const int64_t tile_bytes = b3 ? (int64_t)m.KS * 64 * 16 + (m.cond ? 64 * 16 : 0)
                                  : ((int64_t)m.KS * 64 + 16 + (m.cond ? 64 : 0)) * (int64_t)es;
    nsplit = std::max<int64_t>(nsplit, ceil_div(m.ntiles * tile_bytes, (int64_t)PBN_TUNE(SWEEP_SPLIT_KB, 2048) * 1024));
    // pruned sweeps: a workgroup of the queries' own neighbourhood visits most tiles of its split, and those long workgroups are
    // the sweep's tail - at most PBN_PRUNE_MAX_TILES tiles per split (1e6 x 1e5 handles: fp64 -5...9 %, fp32 -7 %; the score
    // engine's slices are below that anyway - splitting THEM four times finer costs C5 12 %)
    if (m.prune) nsplit = std::max<int64_t>(nsplit, ceil_div(m.ntiles, (int64_t)PBN_TUNE(PRUNE_MAX_TILES, 1024)));
    if (nsplit > 1) nsplit = ceil_div(nsplit, 8) * 8;
    nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, m.ntiles / PBN_TUNE(SWEEP_MIN_TILES, 64)));
    nsplit = std::min<int64_t>(nsplit, 4096);
    const int64_t tps = ceil_div(m.ntiles, nsplit);
    nsplit = ceil_div(m.ntiles, tps);
    // pruned fp64 plain sweeps with per-group masks walk their tiles in two levels (kde_sweep_body): the boxes of the 64-tile batches of every split,
    // behind the partials (they depend on the split size, which depends on the number of queries)
    static const int gmasks = PBN_TUNE(PRUNE_GROUP_MASKS, 1);
    const bool bboxes = m.prune && gmasks && !m.cond && fdt == PBN_F64 && PBN_TUNE(GROUP_BATCH_BOXES, 1) != 0;
    const size_t part_b = ((size_t)nsplit * nqtiles * 16 * P * sizeof(double) + 255) / 256 * 256;
    const size_t bbox_b = bboxes ? (size_t)nsplit * ceil_div(tps, 64) * 2 * m.pdims * sizeof(double) : 0;
    ctx->scratch_part.reserve(part_b + bbox_b);
    SweepArgs sa{};
    sa.Apack = m.Apack; sa.nxpack = m.nxpack; sa.Axpack = m.Axpack;
    sa.Bpack = pa.pack; sa.nypack = pa.npack; sa.Bxpack = pa.xpack; sa.Bxnorm = pa.xnorm;
    sa.ntiles = m.ntiles; sa.nqtiles = nqtiles; sa.tiles_per_split = tps;
    sa.fold = fold ? 1 : 0;
    sa.wmul = wmul ? 1 : 0;
    sa.w32 = (b3 && !m.cond && (m.prune ? f16x2_w32p(m.dm, m.KS) : f16x2_w32(m.dm, m.KS))) ? 1 : 0;
    sa.far_span = (sum_only && m.prune) ? (double)knob_int("PBN_FAR_SPAN", 17) : 0.0;   // sum-only pruned sweeps: fp32 tail for tiles 26+ bits below the sum bound
    const bool guard = knob_int("PBN_MAGIC_GUARD", 1) != 0;   // 0: exp2_magic keeps its clamp everywhere (the same bits, slower: tests/test_magic_exp2_gpu.py)
    sa.tile_r = (guard && m.tile_r && !m.prune && sum_only) ? (const double*)((const char*)m.nxpack + (size_t)m.ntiles * 16 * es * 2) : nullptr;
    sa.fast = sum_only ? 1 : 0;   // only sums leave this call: 2^f on the fp32 transcendental unit; per-row logl keeps the polynomial
    sa.count_redo = knob_int("PBN_SWEEP_COUNT_REDO", 0);
    sa.box_full = (guard && m.prune && m.pdims == m.dm) ? 1 : 0;
    sa.prune = m.prune ? 1 : 0; sa.pdims = m.pdims; sa.prune_margin = prune_margin(fdt, m.N, sum_only); sa.tile_box = m.tile_box; sa.qtile_box = qbox; sa.qtile_thr = qthr; sa.qlb = qlb;
    sa.part = (double*)ctx->scratch_part.p;
    sa.group_masks = gmasks;
    if (bboxes) {
        double* bb = (double*)((char*)ctx->scratch_part.p + part_b);
        launch_batch_boxes(m.tile_box, m.pdims, m.ntiles, tps, (int)nsplit, bb, ctx->stream);
        sa.batch_box = bb; sa.batches_per_split = (int)ceil_div(tps, 64);
    }
    static const bool log_sweeps = PBN_TUNE(SWEEP_LOG, 0) != 0;   // one line per sweep on stderr (tools/c5_sweeps.py)
    if (log_sweeps) std::fprintf(stderr, "pbn-sweep N=%lld n=%lld d=%d cond=%d prune=%d nsub=%lld nsplit=%lld\n", (long long)m.N, (long long)n, m.d, (int)m.cond, (int)m.prune, (long long)(m.prune ? m.nsub : 0), (long long)nsplit);
    { KernelTimer kt(ctx, PBN_K_SWEEP); launch_sweep(sa, fdt, m.KS, m.cond, (int)nsplit, ctx->stream); }
    // f16x2 fragments: queries beyond the f16 range were clamped by the pack - their partials are recomputed in fp64 (a flag test otherwise)
    if (b3) launch_far_fix(pa, m.Apack, m.Axpack, m.KS, m.N, m.ntiles, sa.part, (int)nsplit, nqtiles, m.cond, ctx->stream);

    const int64_t nblocks = ceil_div(n, 256);
    ctx->scratch_misc.reserve((size_t)nblocks * 2 * sizeof(double));
    FinishArgs fa{};
    fa.part = sa.part; fa.nsplit = (int)nsplit; fa.nqtiles = nqtiles; fa.nq = n;
    fa.lognorm = m.lognorm; fa.lognorm_marg = m.lognorm_marg;
    fa.logl = dev_logl; fa.scatter = qperm; fa.block_sums = dev_sum ? (double*)ctx->scratch_misc.p : nullptr;
    fa.block_sums_marg = (m.cond && dev_sum && dev_sum_marg) ? (double*)ctx->scratch_misc.p + nblocks : nullptr;
    { KernelTimer kt(ctx, PBN_K_FINISH); launch_finish(fa, m.cond, dev_sum, ctx->stream, dev_sum_marg); }
}

}  // namespace pbn
