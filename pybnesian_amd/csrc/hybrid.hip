// Hybrid (mixed discrete / continuous) local scores: DiscreteAdaptator factors and discrete CPTs.
//
// Reference (paths under /root/reference/pybnesian/):
//   factors/discrete/DiscreteAdaptator.hpp:201-348  HCKDE / CLinearGaussianCPD: one base factor per configuration
//        of the discrete parents, fitted on that configuration's training rows; slogl sums the configurations
//        that have a factor (missing factor -> contributes 0); CKDEFitter swallows SingularCovarianceData,
//        LinearGaussianFitter drops factors with variance < machine_tol or inf (CKDE.hpp:752-768,
//        LinearGaussianCPD.hpp:127-138)
//   factors/discrete/discrete_indices.cpp:93-204     strides in evidence order, configuration index, slices
//   learning/scores/bic.cpp:29-96                    bic_clg, bic_discrete
//   learning/parameters/mle_DiscreteFactor.cpp:5-41, factors/discrete/DiscreteFactor.cpp:133-171  CPT MLE, slogl
//
// Device design: the grouping is integer work on the host (one counting sort of the rows by
// (region, configuration) per candidate, a19 of SURVEY.md §8a); the per-slice continuous work reuses the
// same kernels as the homogeneous path through device gather lists: pilot-shifted Gram moments per
// (region, configuration) - additive, so the statistics of a training fold are sums over the other folds -
// and, for CKDE, pack(train slice) -> pack(test slice) -> fused sweep -> finish.
#include <algorithm>
#include <cmath>
#include <numeric>

#include "hostmath.hpp"
#include "kde_model.hpp"
#include "scoring_internal.hpp"
#include "stats_kernels.hpp"

namespace pbn {
namespace score {

namespace {

struct Region { int64_t r0, r1; };

// rows of every region grouped by configuration: list(region, config) = rows[off[region*nc + config] ...)
struct Groups {
    int nc = 1;
    std::vector<int32_t> rows;
    std::vector<int64_t> off;
    int64_t count(int region, int c) const { return off[(size_t)region * nc + c + 1] - off[(size_t)region * nc + c]; }
    int64_t begin(int region, int c) const { return off[(size_t)region * nc + c]; }
};

std::vector<Region> regions_of(const pbn_scoredata* sd, int kind) {
    std::vector<Region> r;
    if (kind == PBN_SCORE_CVLIK) {
        for (int f = 0; f < sd->k; ++f) r.push_back({sd->limits[f], sd->limits[f + 1]});
    } else if (kind == PBN_SCORE_HOLDOUT) {
        r.push_back({0, sd->n_cv});
        r.push_back({sd->n_cv, sd->n_cv + sd->n_hold});
    } else {
        r.push_back({0, sd->n_cv});
    }
    return r;
}

void build_groups(const pbn_scoredata* sd, const std::vector<int>& dpar, const std::vector<Region>& regions, Groups& g,
                  std::vector<int>& strides) {
    // discrete_indices.cpp:113-132: stride_0 = 1, stride_i = stride_{i-1} * card_{i-1}, evidence order
    strides.assign(dpar.size(), 1);
    int nc = 1;
    for (size_t i = 0; i < dpar.size(); ++i) {
        strides[i] = nc;
        nc *= sd->card[dpar[i] - sd->n];
    }
    g.nc = nc;
    const size_t cells = regions.size() * (size_t)nc;
    g.off.assign(cells + 1, 0);
    auto config = [&](int64_t r) {
        int c = 0;
        for (size_t i = 0; i < dpar.size(); ++i) c += sd->codes[dpar[i] - sd->n][r] * strides[i];
        return c;
    };
    int64_t total = 0;
    for (size_t ri = 0; ri < regions.size(); ++ri) {
        for (int64_t r = regions[ri].r0; r < regions[ri].r1; ++r) ++g.off[ri * nc + config(r) + 1];
        total += regions[ri].r1 - regions[ri].r0;
    }
    for (size_t i = 0; i < cells; ++i) g.off[i + 1] += g.off[i];
    g.rows.resize((size_t)total);
    std::vector<int64_t> cur(g.off.begin(), g.off.end() - 1);
    for (size_t ri = 0; ri < regions.size(); ++ri)
        for (int64_t r = regions[ri].r0; r < regions[ri].r1; ++r) g.rows[cur[ri * nc + config(r)]++] = (int32_t)r;
}

// ---- discrete variable: DiscreteFactor MLE + slogl, bic_discrete --------------------------------------------
double score_discrete(const pbn_scoredata* sd, int kind, int var, const int* parents, int p) {
    const int n = sd->n;
    for (int i = 0; i < p; ++i)
        if (parents[i] < n)
            throw invalid_error("Local score for a discrete variable cannot be calculated because the parents/evidence contains non-discrete variables.");
    const int card0 = sd->card[var - n];
    std::vector<int> strides(p + 1);
    int joint = card0;
    strides[0] = 1;
    for (int i = 0; i < p; ++i) { strides[i + 1] = joint; joint *= sd->card[parents[i] - n]; }
    const int configs = joint / card0;
    auto index = [&](int64_t r) {
        int c = sd->codes[var - n][r];
        for (int i = 0; i < p; ++i) c += sd->codes[parents[i] - n][r] * strides[i + 1];
        return c;
    };
    std::vector<Region> regions = regions_of(sd, kind);
    std::vector<std::vector<int64_t>> counts(regions.size(), std::vector<int64_t>(joint, 0));
    for (size_t ri = 0; ri < regions.size(); ++ri)
        for (int64_t r = regions[ri].r0; r < regions[ri].r1; ++r) ++counts[ri][index(r)];
    if (kind == PBN_SCORE_BIC) {  // bic.cpp:66-96
        const auto& jc = counts[0];
        double ll = 0;
        int64_t total = 0;
        for (int k = 0; k < configs; ++k) {
            int64_t sum = 0;
            for (int i = 0; i < card0; ++i) sum += jc[(size_t)k * card0 + i];
            total += sum;
            if (sum > 0) {
                const double inv = 1.0 / (double)sum;
                for (int i = 0; i < card0; ++i) {
                    const int64_t c = jc[(size_t)k * card0 + i];
                    if (c > 0) ll += (double)c * std::log((double)c * inv);
                }
            }
        }
        return ll - std::log((double)total) * 0.5 * (card0 - 1) * configs;
    }
    // likelihood scores: fit on train counts, slogl on test counts (mle_DiscreteFactor.cpp:5-41)
    auto unit = [&](const std::vector<int64_t>& train, const std::vector<int64_t>& test) {
        double res = 0;
        for (int k = 0; k < configs; ++k) {
            int64_t sum = 0;
            for (int i = 0; i < card0; ++i) sum += train[(size_t)k * card0 + i];
            for (int i = 0; i < card0; ++i) {
                const int64_t t = test[(size_t)k * card0 + i];
                if (t == 0) continue;
                double lp;
                if (sum == 0) lp = std::log(1.0 / card0);
                else lp = std::log((double)train[(size_t)k * card0 + i]) - std::log((double)sum);
                res += (double)t * lp;  // the reference adds logprob once per test row
            }
        }
        return res;
    };
    if (kind == PBN_SCORE_HOLDOUT) return unit(counts[0], counts[1]);
    std::vector<int64_t> all(joint, 0), train(joint);
    for (auto& c : counts)
        for (int i = 0; i < joint; ++i) all[i] += c[i];
    double acc = 0;
    for (size_t f = 0; f < regions.size(); ++f) {
        for (int i = 0; i < joint; ++i) train[i] = all[i] - counts[f][i];
        acc += unit(train, counts[f]);
    }
    return acc;
}

// moments of columns `cols` (d) for every (region, configuration)
void group_moments(pbn_scoredata* sd, const Groups& g, size_t nregions, const int* cols, int d, const int32_t* dev_rows,
                   std::vector<Stats>& M) {
    M.assign(nregions * (size_t)g.nc, Stats());
    for (size_t ri = 0; ri < nregions; ++ri)
        for (int c = 0; c < g.nc; ++c) {
            Stats& st = M[ri * g.nc + c];
            st.zero(d);
            st.N = g.count((int)ri, c);
            if (st.N > 0)
                gram_raw(sd->table(), cols, d, 0, st.N, dev_rows + g.begin((int)ri, c), sd->shift_dev.p, st.S.data(), st.G.data());
        }
}

// means / centred SSE from moments that were computed for exactly the columns `cols` (index i <-> cols[i])
void local_moments(const pbn_scoredata* sd, const Stats& st, const int* cols, int d, double* mu, double* sse) {
    const double N = (double)st.N;
    for (int i = 0; i < d; ++i) mu[i] = sd->shift[cols[i]] + (st.N > 0 ? st.S[i] / N : 0.0);
    for (int j = 0; j < d; ++j)
        for (int i = 0; i < d; ++i) sse[i + (size_t)j * d] = st.G[i + (size_t)j * d] - (st.N > 0 ? st.S[i] * st.S[j] / N : 0.0);
}

double local_lg_slogl(const pbn_scoredata* sd, const Stats& test, const int* cols, int p, const double* beta, double variance) {
    const int d = p + 1;
    const double Nt = (double)test.N;
    double c = beta[0] - sd->shift[cols[0]];
    for (int j = 1; j <= p; ++j) c += beta[j] * sd->shift[cols[j]];
    auto G = [&](int i, int j) { return test.G[i + (size_t)j * d]; };
    double rss = G(0, 0), lin = test.S[0];
    for (int j = 1; j <= p; ++j) { rss -= 2 * beta[j] * G(0, j); lin -= beta[j] * test.S[j]; }
    for (int i = 1; i <= p; ++i)
        for (int j = 1; j <= p; ++j) rss += beta[i] * beta[j] * G(i, j);
    rss += -2 * c * lin + Nt * c * c;
    rss = std::max(rss, 0.0);
    return -0.5 * Nt * (std::log(variance) + LOG_2PI) - 0.5 * rss / variance;
}

}  // namespace

double score_hybrid(pbn_scoredata* sd, int kind, int var, int node_type, const int* parents, int p) {
    const int n = sd->n;
    if (kind == PBN_SCORE_BGE) throw invalid_error("BGe is not defined for networks with discrete variables.");
    if (var >= n) {
        if (node_type != PBN_NODE_DISCRETE) throw invalid_error("pbn_score_batch: discrete column scored with a continuous node type");
        return score_discrete(sd, kind, var, parents, p);
    }
    if (node_type == PBN_NODE_DISCRETE) throw invalid_error("pbn_score_batch: continuous column scored as DiscreteFactor");
    std::vector<int> dpar, cols{var};
    for (int i = 0; i < p; ++i) (parents[i] >= n ? dpar : cols).push_back(parents[i]);
    const int d = (int)cols.size(), pc = d - 1;
    if (d > 17) throw invalid_error("pbn_score_batch: too many continuous parents");
    pbn_ctx* ctx = sd->ctx;
    const pbn_table* t = sd->table();
    std::vector<Region> regions = regions_of(sd, kind);
    Groups g;
    std::vector<int> strides;
    build_groups(sd, dpar, regions, g, strides);
    sd->rows_dev.reserve(g.rows.size() + 16);
    if (!g.rows.empty())
        HIP_CHECK(hipMemcpyAsync(sd->rows_dev.p, g.rows.data(), g.rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    std::vector<Stats> M;
    group_moments(sd, g, regions.size(), cols.data(), d, sd->rows_dev.p, M);
    std::vector<double> mu(d), sse((size_t)d * d), beta(d), H((size_t)d * d);

    if (kind == PBN_SCORE_BIC) {  // bic.cpp:29-64
        if (node_type != PBN_NODE_LG) throw invalid_error("BIC: only LinearGaussianCPD / DiscreteFactor node types are implemented");
        double loglik = 0;
        int64_t valid = 0;
        for (int c = 0; c < g.nc; ++c) {
            const Stats& st = M[c];
            valid += st.N;
            if (st.N == 0) continue;
            local_moments(sd, st, cols.data(), d, mu.data(), sse.data());
            const double v = lg_fit(st.N, pc, mu.data(), sse.data(), beta.data());
            if (v < MACHINE_TOL || std::isinf(v)) return -INF;
            const double nv = (double)st.N;
            loglik += 0.5 * (1 + (double)pc - nv) - 0.5 * nv * LOG_2PI - nv * 0.5 * std::log(v);
        }
        return loglik - std::log((double)valid) * 0.5 * g.nc * (pc + 2);
    }

    // likelihood scores (cv_likelihood.cpp:11-25 / holdout_likelihood.cpp:14-23 over DiscreteAdaptator factors)
    const bool cv = kind == PBN_SCORE_CVLIK;
    const int units = cv ? (int)regions.size() : 1;
    std::vector<Stats> allc(g.nc);
    if (cv)
        for (int c = 0; c < g.nc; ++c) {
            allc[c].zero(d);
            for (size_t f = 0; f < regions.size(); ++f) allc[c].add(M[f * g.nc + c]);
        }
    double acc = 0;
    std::vector<int32_t> train_rows;
    dev_buf<double> dsums;
    int n_slots = 0;
    if (node_type == PBN_NODE_CKDE) {
        dsums.alloc((size_t)units * g.nc + 1);
        HIP_CHECK(hipMemsetAsync(dsums.p, 0, ((size_t)units * g.nc + 1) * sizeof(double), ctx->stream));
    }
    dev_buf<int32_t> train_dev;  // concatenated training gather list of the current slice (CKDE, CV)
    Stats train;
    for (int u = 0; u < units; ++u) {
        for (int c = 0; c < g.nc; ++c) {
            const Stats* tr;
            const Stats* te;
            if (cv) { stats_minus(allc[c], M[(size_t)u * g.nc + c], train); tr = &train; te = &M[(size_t)u * g.nc + c]; }
            else { tr = &M[c]; te = &M[(size_t)g.nc + c]; }
            if (tr->N == 0) continue;  // empty training slice -> no factor (DiscreteAdaptator.hpp:266-268)
            local_moments(sd, *tr, cols.data(), d, mu.data(), sse.data());
            if (node_type == PBN_NODE_LG) {
                const double v = lg_fit(tr->N, pc, mu.data(), sse.data(), beta.data());
                if (v < MACHINE_TOL || std::isinf(v)) continue;  // LinearGaussianFitter -> nullptr
                if (te->N == 0) continue;
                acc += local_lg_slogl(sd, *te, cols.data(), pc, beta.data(), v);
                continue;
            }
            // CKDE slice: CKDEFitter turns SingularCovarianceData into "no factor"
            if (tr->N <= 1) continue;
            const double inv = 1.0 / (double)(tr->N - 1);
            for (auto& x : sse) x *= inv;
            try {
                bandwidth_from_cov(sd->selector, PBN_BW_FULL, sse.data(), d, tr->N, sd->dtype, H.data());
            } catch (const singular_error&) {
                continue;
            }
            if (te->N == 0) continue;
            KdeModel m;
            try {
                kde_prepare(m, sd->dtype, d, tr->N, H.data(), PBN_BW_FULL, true, mu.data());
            } catch (const singular_error&) {
                continue;
            }
            // training gather list
            const int32_t* dev_train;
            if (cv) {
                train_rows.clear();
                for (size_t f = 0; f < regions.size(); ++f) {
                    if ((int)f == u) continue;
                    const int64_t b = g.begin((int)f, c), cnt = g.count((int)f, c);
                    train_rows.insert(train_rows.end(), g.rows.begin() + b, g.rows.begin() + b + cnt);
                }
                train_dev.reserve(train_rows.size());
                HIP_CHECK(hipMemcpyAsync(train_dev.p, train_rows.data(), train_rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
                dev_train = train_dev.p;
            } else {
                dev_train = sd->rows_dev.p + g.begin(0, c);
            }
            const int32_t* dev_test = sd->rows_dev.p + (cv ? g.begin(u, c) : g.begin(1, c));
            const KdePackBytes pb = kde_pack_bytes(sd->dtype, m.dm, m.cond, tr->N);
            auto align = [](size_t x) { return (x + 255) / 256 * 256; };
            ctx->scratch_train.reserve(align(pb.apack) + align(pb.nxpack) + align(pb.axpack) + 256);
            char* arena = ctx->scratch_train.p;
            m.Apack = arena;
            m.nxpack = arena + align(pb.apack);
            m.Axpack = m.cond ? arena + align(pb.apack) + align(pb.nxpack) : nullptr;
            kde_pack_train(ctx, m, t, cols.data(), 0, 0, 0, dev_train, /*prune=*/true);
            kde_eval_enqueue(ctx, m, t, cols.data(), 0, te->N, nullptr, dsums.p + n_slots, dev_test);
            HIP_CHECK(hipStreamSynchronize(ctx->stream));  // the host-side gather list and the arenas are reused by the next slice
            ++n_slots;
        }
    }
    if (node_type == PBN_NODE_CKDE && n_slots > 0) {
        std::vector<double> hs((size_t)n_slots);
        HIP_CHECK(hipMemcpyAsync(hs.data(), dsums.p, (size_t)n_slots * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (double v : hs) acc += v;
    }
    return acc;
}

}  // namespace score
}  // namespace pbn
