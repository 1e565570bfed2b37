// Unbiased cross-validation bandwidth selector (kde/UCV.{hpp,cpp}): the objective's two pair sums on the device, the
// simplex search on the host.
//
// N * UCV(H) = exp(lognorm_2H) + 2/N sum_{i<j} K_2H(x_i - x_j) - 4/(N-1) sum_{i<j} K_H(x_i - x_j)      (UCV.cpp:226-298, 300-360)
// The reference enumerates the N(N-1)/2 pairs in chunks of 10^6 through triangular-index kernels (KDE.cl.src:470-574).
// Here the training rows are swept against themselves with the KDE machinery: w = exp(-1/2 d^2) comes out of the MFMA
// Gram form, exp(-1/4 d^2) is its square root, self pairs never overflow so the offset stays 0, and the ordered-pair
// totals minus the N diagonal terms, halved, are the two sums.
//
// The search is NLopt's LN_NELDERMEAD in the reference (UCV.cpp:469-481, 505-517; ftol_rel = xtol_rel = 1e-4).  NLopt
// is not in this image: the simplex method below restates its published algorithm (nldrmd.c: reflection 1,
// expansion 2, contraction 1/2, shrink 1/2, initial step x_i, the ftol / xtol tests of stop.c).  Objective values are
// checked in tests against an independent all-pairs restatement; the optimiser's trajectory is UNPINNED (no reference test covers UCV).
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>

#include "common.hpp"
#include "hostmath.hpp"
#include "kde_kernels.hpp"
#include "kde_model.hpp"
#include "stats_kernels.hpp"

using namespace pbn;

namespace {

constexpr double MACHINE_TOL = 1.4901161193847656e-08;

struct UcvScorer {
    pbn_ctx* ctx;
    const pbn_table* t;
    std::vector<int> cols;
    int d;
    int64_t row0, N;
    std::vector<double> center;
    int64_t evals = 0;

    // bw: H (d*d col-major) for PBN_BW_FULL, h (d variances) for PBN_BW_DIAG
    double score(const double* bw, int kind) {
        ++evals;
        KdeModel m;
        kde_prepare(m, t->dtype, d, N, bw, kind, false, center.data());   // throws singular_error when H is not PD
        const int KS = (d + 3) / 4;
        const bool wide = d > 16;                     // beyond the templated shapes: fp64 fragments through the generic pack, runtime-sized kernel
        const int fdt = wide ? PBN_F64 : t->dtype;
        const size_t es = dtype_size(fdt);
        const int64_t ntiles = ceil_div(N, 16);
        const size_t frag = (size_t)ntiles * KS * 64 * es, nrm = (size_t)ntiles * 16 * es;
        auto al = [](size_t x) { return (x + 255) / 256 * 256; };
        ctx->scratch_train.reserve(al(frag) + al(nrm) + 256);
        ctx->scratch_q.reserve(al(frag) + al(nrm) + 256);
        struct { void* pack; void* npack; } pa{ctx->scratch_train.p, ctx->scratch_train.p + al(frag)}, pq{ctx->scratch_q.p, ctx->scratch_q.p + al(frag)};
        if (wide) {
            WidePackArgs wa{};
            kde_wide_pack_args(ctx, wa, t, cols.data(), d, d, m.W.data(), d, m.mu.data(), nullptr);
            wa.rows = nullptr; wa.row0 = row0; wa.n0 = N; wa.row1 = 0; wa.n = N; wa.ntiles = ntiles; wa.upack = nullptr;
            wa.is_query = 0; wa.pack = (double*)pa.pack; wa.npack = (double*)pa.npack;
            { KernelTimer kt(ctx, PBN_K_PACK); launch_pack_wide(wa, ctx->stream); }
            wa.is_query = 1; wa.pack = (double*)pq.pack; wa.npack = (double*)pq.npack;
            { KernelTimer kt(ctx, PBN_K_PACK); launch_pack_wide(wa, ctx->stream); }
        } else {
            PackArgs pt{};
            pt.base = t->data; pt.ld = t->ld; pt.d = d; pt.dm = d; pt.KS = KS;
            for (int i = 0; i < d; ++i) { pt.cols[i] = cols[i]; pt.mu[i] = m.mu[i]; }
            for (int i = 0; i < d * d; ++i) pt.W[i] = m.W[i];
            pt.row0 = row0; pt.n0 = N; pt.row1 = 0; pt.n = N; pt.ntiles = ntiles;
            pt.is_query = 0; pt.pack = pa.pack; pt.npack = pa.npack;
            { KernelTimer kt(ctx, PBN_K_PACK); launch_pack_classic(pt, t->dtype, ctx->stream); }
            PackArgs pu = pt;
            pu.is_query = 1; pu.pack = pq.pack; pu.npack = pq.npack;
            { KernelTimer kt(ctx, PBN_K_PACK); launch_pack_classic(pu, t->dtype, ctx->stream); }
        }
        const int64_t qblocks = ceil_div(ntiles, 8);
        int64_t nsplit = std::max<int64_t>(1, ceil_div((int64_t)ctx->num_cus * 16, qblocks));
        nsplit = std::min<int64_t>(nsplit, std::max<int64_t>(1, ntiles / 16));
        const int64_t tps = ceil_div(ntiles, nsplit);
        nsplit = ceil_div(ntiles, tps);
        ctx->scratch_part.reserve((size_t)nsplit * ntiles * 16 * 4 * sizeof(double));
        const int64_t nblocks = ceil_div(N, 256);
        ctx->scratch_misc.reserve((size_t)(2 * nblocks + 2) * sizeof(double));
        CdfArgs ca{};
        ca.Apack = pa.pack; ca.nxpack = pa.npack; ca.utrain = nullptr;
        ca.Bpack = pq.pack; ca.nypack = pq.npack; ca.uquery = nullptr;
        ca.ntiles = ntiles; ca.nqtiles = ntiles; ca.tiles_per_split = tps;
        ca.part = (double*)ctx->scratch_part.p;
        double* blk = (double*)ctx->scratch_misc.p;
        double* dout = blk + 2 * nblocks;
        { KernelTimer kt(ctx, PBN_K_SWEEP); launch_ucv(ca, fdt, KS, (int)nsplit, N, blk, dout, ctx->stream); }
        double tot[2];
        HIP_CHECK(hipMemcpyAsync(tot, dout, 2 * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const double n = (double)N;
        const double lognorm_H = m.lognorm + std::log(n);                 // kde_prepare folds -log N in
        const double lognorm_2H = lognorm_H - 0.5 * d * std::log(2.0);
        const double sh = std::exp(lognorm_H) * 0.5 * (tot[0] - n);        // sum_{i<j} K_H
        const double s2h = std::exp(lognorm_2H) * 0.5 * (tot[1] - n);      // sum_{i<j} K_2H
        return std::exp(lognorm_2H) + 2 * s2h / n - 4 * sh / (n - 1);
    }
};

// ---- Nelder-Mead as NLopt's nldrmd.c states it ------------------------------------------------------------------------
bool relstop(double vold, double vnew, double reltol) {
    if (std::isinf(vold)) return false;
    return std::fabs(vnew - vold) < reltol * (std::fabs(vnew) + std::fabs(vold)) * 0.5 || (reltol > 0 && vnew == vold);
}

bool close_(double a, double b) { return std::fabs(a - b) <= 1e-13 * (std::fabs(a) + std::fabs(b)); }

// xnew = c + scale * (c - xold); false when the new point coincides with c or xold
bool reflectpt(int n, double* xnew, const double* c, double scale, const double* xold) {
    bool equalc = true, equalold = true;
    for (int i = 0; i < n; ++i) {
        const double v = c[i] + scale * (c[i] - xold[i]);
        equalc = equalc && close_(v, c[i]);
        equalold = equalold && close_(v, xold[i]);
        xnew[i] = v;
    }
    return !(equalc || equalold);
}

template <typename F>
double nelder_mead(int n, std::vector<double>& x, F&& f, double ftol_rel, double xtol_rel, int max_evals) {
    const double alpha = 1.0, beta = 0.5, gamm = 2.0, delta = 0.5;
    std::vector<std::vector<double>> pts(n + 1, std::vector<double>(n));
    std::vector<double> fv(n + 1);
    int evals = 0;
    pts[0] = x;
    fv[0] = f(pts[0].data()); ++evals;
    for (int i = 0; i < n; ++i) {
        pts[i + 1] = x;
        double step = x[i];                       // nlopt_set_default_initial_step without bounds
        if (step == 0.0 || std::isinf(step)) step = 1.0;
        pts[i + 1][i] += step;
        fv[i + 1] = f(pts[i + 1].data()); ++evals;
    }
    std::vector<double> c(n), xcur(n);
    while (true) {
        int lo = 0, hi = 0;
        for (int i = 1; i <= n; ++i) {
            if (fv[i] < fv[lo]) lo = i;
            if (fv[i] >= fv[hi]) hi = i;          // the last of equal maxima, as the tree's rightmost node
        }
        if (relstop(fv[hi], fv[lo], ftol_rel)) break;
        std::fill(c.begin(), c.end(), 0.0);
        for (int i = 0; i <= n; ++i)
            if (i != hi)
                for (int j = 0; j < n; ++j) c[j] += pts[i][j] / n;
        bool xstop = true;
        for (int j = 0; j < n; ++j) xstop = xstop && relstop(pts[hi][j], c[j], xtol_rel);
        if (xstop || evals >= max_evals) break;
        double second = -std::numeric_limits<double>::infinity();   // predecessor of the highest
        for (int i = 0; i <= n; ++i)
            if (i != hi) second = std::max(second, fv[i]);
        if (!reflectpt(n, xcur.data(), c.data(), alpha, pts[hi].data())) break;
        const double fr = f(xcur.data()); ++evals;
        if (fr < fv[lo]) {                        // new best: try to expand
            std::vector<double> xe(n);
            if (!reflectpt(n, xe.data(), c.data(), gamm, pts[hi].data())) break;
            const double fe = f(xe.data()); ++evals;
            if (fe >= fr) { pts[hi] = xcur; fv[hi] = fr; } else { pts[hi] = xe; fv[hi] = fe; }
        } else if (fr < second) {                 // accept the reflection
            pts[hi] = xcur; fv[hi] = fr;
        } else {                                   // contract
            if (!reflectpt(n, xcur.data(), c.data(), fv[hi] <= fr ? -beta : beta, pts[hi].data())) break;
            const double fc = f(xcur.data()); ++evals;
            if (fc < fr && fc < fv[hi]) { pts[hi] = xcur; fv[hi] = fc; }
            else {                                 // shrink towards the best point
                bool ok = true;
                for (int i = 0; i <= n && ok; ++i) {
                    if (i == lo) continue;
                    std::vector<double> xs(n);
                    ok = reflectpt(n, xs.data(), pts[lo].data(), -delta, pts[i].data());
                    if (!ok) break;
                    pts[i] = xs;
                    fv[i] = f(pts[i].data()); ++evals;
                }
                if (!ok) break;
            }
        }
    }
    int lo = 0;
    for (int i = 1; i <= n; ++i) if (fv[i] < fv[lo]) lo = i;
    x = pts[lo];
    return fv[lo];
}

UcvScorer make_scorer(pbn_ctx* ctx, const pbn_table* table, const int* cols, int d, int64_t row0, int64_t n) {
    if (!ctx || !table) throw invalid_error("pbn_ucv: null argument");
    check_cols(table, cols, d, "pbn_ucv");
    check_range(table, row0, n, "pbn_ucv");
    if (d < 1) throw invalid_error("UCV: at least one variable is needed");
    if (n < 2) throw invalid_error("UCV: at least two training instances are needed");
    HIP_CHECK(hipSetDevice(ctx->device));
    UcvScorer s{ctx, table, std::vector<int>(cols, cols + d), d, row0, n, std::vector<double>(d, 0.0)};
    // centre at the pilot means (any offset is exact in the differences)
    ctx->scratch_red.reserve((size_t)table->n_cols + 8);
    for (int c0 = 0; c0 < d; c0 += 64) {   // 64 columns per pilot launch
        GramCols sel{};
        const int dc = std::min(64, d - c0);
        for (int i = 0; i < dc; ++i) sel.cols[i] = cols[c0 + i];
        launch_pilot(table->data, table->ld, sel, dc, row0, nullptr, n, table->dtype, ctx->scratch_red.p, ctx->stream);
    }
    std::vector<double> all((size_t)table->n_cols);
    HIP_CHECK(hipMemcpyAsync(all.data(), ctx->scratch_red.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < d; ++i) s.center[i] = all[cols[i]];
    return s;
}

}  // namespace

extern "C" {

// UCVScorer::score_unconstrained / score_diagonal (kde/UCV.cpp:226-360): N * UCV of the given bandwidth.
int pbn_ucv_score(pbn_ctx* ctx, const pbn_table* table, const int* cols, int d, int64_t row0, int64_t n, const double* bandwidth,
                  int kind, double* out) {
    return guarded(mu_of(ctx), [&] {
        if (!bandwidth || !out) throw invalid_error("pbn_ucv_score: null argument");
        if (kind != PBN_BW_FULL && kind != PBN_BW_DIAG) throw invalid_error("pbn_ucv_score: unknown bandwidth kind");
        UcvScorer s = make_scorer(ctx, table, cols, d, row0, n);
        *out = s.score(bandwidth, kind);
    });
}

// UCV::bandwidth / UCV::diag_bandwidth (kde/UCV.cpp:452-526): simplex search from `start` (the normal reference
// bandwidth) over the Cholesky factor's vech (full) or the square roots of the variances (diagonal), with the
// reference's guards on the determinant and on the score (wrap_ucv_optim / wrap_ucv_diag_optim, :395-450).
int pbn_ucv_bandwidth(pbn_ctx* ctx, const pbn_table* table, const int* cols, int d, int64_t row0, int64_t n, int kind,
                      const double* start, double* out, int64_t* n_evals) {
    return guarded(mu_of(ctx), [&] {
        if (!start || !out) throw invalid_error("pbn_ucv_bandwidth: null argument");
        UcvScorer s = make_scorer(ctx, table, cols, d, row0, n);
        if (kind == PBN_BW_DIAG) {
            const double start_score = s.score(start, PBN_BW_DIAG);
            double start_det = 1.0;
            for (int i = 0; i < d; ++i) start_det *= start[i];
            std::vector<double> x(d), h(d);
            for (int i = 0; i < d; ++i) x[i] = std::sqrt(start[i]);
            auto obj = [&](const double* v) {
                double det_sqrt = 1.0;
                for (int i = 0; i < d; ++i) det_sqrt *= v[i];
                const double det = det_sqrt * det_sqrt;
                if (det <= MACHINE_TOL || det < 1e-3 * start_det || det > 1e3 * start_det) return start_score + 10e-8;
                for (int i = 0; i < d; ++i) h[i] = v[i] * v[i];
                const double sc = s.score(h.data(), PBN_BW_DIAG);
                if (std::fabs(sc) > 1e3 * std::fabs(start_score)) return start_score + 10e-8;
                return sc;
            };
            nelder_mead(d, x, obj, 1e-4, 1e-4, 100000);
            for (int i = 0; i < d; ++i) out[i] = x[i] * x[i];
        } else if (kind == PBN_BW_FULL) {
            const double start_score = s.score(start, PBN_BW_FULL);
            const double start_det = hm::determinant(start, d);
            std::vector<double> L((size_t)d * d);
            if (!hm::cholesky(start, d, L.data())) throw singular_error("UCV: the starting bandwidth is not positive-definite");
            const int nv = d * (d + 1) / 2;
            std::vector<double> x(nv), Lx((size_t)d * d), H((size_t)d * d);
            int pos = 0;
            for (int j = 0; j < d; ++j)            // vech: lower triangle, column by column (util/vech_ops.hpp)
                for (int i = j; i < d; ++i) x[pos++] = L[i + (size_t)j * d];
            auto unpack = [&](const double* v) {
                std::fill(Lx.begin(), Lx.end(), 0.0);
                int q = 0;
                for (int j = 0; j < d; ++j)
                    for (int i = j; i < d; ++i) Lx[i + (size_t)j * d] = v[q++];
                for (int j = 0; j < d; ++j)
                    for (int i = 0; i < d; ++i) {
                        double acc = 0;
                        for (int k = 0; k < d; ++k) acc += Lx[i + (size_t)k * d] * Lx[j + (size_t)k * d];
                        H[i + (size_t)j * d] = acc;
                    }
            };
            auto obj = [&](const double* v) {
                unpack(v);
                double logdet = 0;
                for (int i = 0; i < d; ++i) logdet += std::log(Lx[i + (size_t)i * d]);
                const double det = std::exp(2 * logdet);
                if (det <= MACHINE_TOL || det < 1e-3 * start_det || det > 1e3 * start_det || std::isnan(det)) return start_score + 10e-8;
                double sc;
                try { sc = s.score(H.data(), PBN_BW_FULL); } catch (const singular_error&) { return start_score + 10e-8; }
                if (std::fabs(sc) > 1e3 * std::fabs(start_score)) return start_score + 10e-8;
                return sc;
            };
            nelder_mead(nv, x, obj, 1e-4, 1e-4, 100000);
            unpack(x.data());
            std::memcpy(out, H.data(), (size_t)d * d * sizeof(double));
        } else {
            throw invalid_error("pbn_ucv_bandwidth: unknown bandwidth kind");
        }
        if (n_evals) *n_evals = s.evals;
    });
}

}  // extern "C"
