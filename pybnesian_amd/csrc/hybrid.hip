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
#include <cstdlib>
#include <memory>
#include <numeric>

#include "hostmath.hpp"
#include "kde_group.hpp"
#include "kde_kernels.hpp"
#include "kde_model.hpp"
#include "scoring_internal.hpp"
#include "stats_kernels.hpp"

namespace pbn {
namespace score {

namespace {

struct Region { int64_t r0, r1; };

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

// ---- discrete variable: DiscreteFactor MLE + slogl, bic_discrete --------------------------------------------
double score_discrete(const pbn_scoredata* sd, int kind, int var, const int* parents_in, int p) {
    const int n = sd->n;
    for (int i = 0; i < p; ++i)
        if (parents_in[i] < n)
            throw invalid_error("Local score for a discrete variable cannot be calculated because the parents/evidence contains non-discrete variables.");
    // parents in ascending order: the configurations are then visited - and their terms added - in one order whatever order the
    // parents came in, so the score is a function of (variable, parent SET) to the last bit and can be memoised like the others
    std::vector<int> psorted(parents_in, parents_in + p);
    std::sort(psorted.begin(), psorted.end());
    const int* parents = psorted.data();
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

// ---- rows grouped by (configuration, region), cached per set of discrete parents -----------------------------------------
// One counting sort per SET of discrete parents and score kind, kept on the score data (host offsets + device row list +
// the piece table of the segmented moments kernel): C5's restricted hill-climb asks for ~50 distinct sets in 1 671 local
// scores.  Layout: configuration-major, the regions (CV folds / hold-out train, test) of one configuration next to each
// other in the list, so that "all folds but u" of a configuration is two contiguous ranges of it - the training gather list of
// a slice needs no copy (PackArgs: two ranges, then through the list).  Configurations are numbered in CANONICAL order
// (parents sorted by column id, stride_0 = 1); a candidate's own numbering (evidence order, discrete_indices.cpp:113-132) is
// mapped onto it, which keeps the reference's summation order over configurations.
constexpr int SEG_PIECE = 4096;
#ifndef PBN_HYBRID_SPLIT_MAX_D
#define PBN_HYBRID_SPLIT_MAX_D 1   // fp32 CKDE slices with more variables than this: the fused sweep (C5: 21.3 s fused, 22.1 s with slices up to 4 variables split - fewer exponentials, more launches)
#endif

// -> shared: whoever enqueues device work that reads the grouping's row list keeps the pointer until that work has finished (a
// HybridBatch holds the groupings of its candidates until flush()), so the cache starting over cannot free a list in flight
std::shared_ptr<const HybridGrouping> grouping_for(pbn_scoredata* sd, int kind, const std::vector<int>& dpar_sorted, const std::vector<Region>& regions) {
    std::vector<int> key{kind};
    key.insert(key.end(), dpar_sorted.begin(), dpar_sorted.end());
    auto it = sd->groupings.find(key);
    if (it != sd->groupings.end()) return it->second;
    // each grouping holds a 4 B / row device list: beyond 256 of them (a search over very many discrete parent sets) start over
    // (PBN_HYBRID_GROUPINGS: the cap, lowered by the test that crosses the reset inside one batch)
    static const size_t cap = (size_t)std::max(1ll, knob_ll("PBN_HYBRID_GROUPINGS", 256));
    if (sd->groupings.size() >= cap) {
        HIP_CHECK(hipStreamSynchronize(sd->ctx->stream));
        sd->groupings.clear();
    }
    auto gp = std::make_shared<HybridGrouping>();
    HybridGrouping& g = *gp;
    std::vector<int> strides(dpar_sorted.size(), 1);
    int nc = 1;
    for (size_t i = 0; i < dpar_sorted.size(); ++i) { strides[i] = nc; nc *= sd->card[dpar_sorted[i] - sd->n]; }
    g.nc = nc;
    g.nregions = (int)regions.size();
    const size_t cells = (size_t)nc * regions.size();
    g.off.assign(cells + 1, 0);
    auto config = [&](int64_t r) {
        int c = 0;
        for (size_t i = 0; i < dpar_sorted.size(); ++i) c += sd->codes[dpar_sorted[i] - sd->n][r] * strides[i];
        return c;
    };
    int64_t total = 0;
    for (size_t ri = 0; ri < regions.size(); ++ri) {
        for (int64_t r = regions[ri].r0; r < regions[ri].r1; ++r) ++g.off[(size_t)config(r) * regions.size() + ri + 1];
        total += regions[ri].r1 - regions[ri].r0;
    }
    for (size_t i = 0; i < cells; ++i) g.off[i + 1] += g.off[i];
    std::vector<int32_t> rows((size_t)total);
    std::vector<int64_t> cur(g.off.begin(), g.off.end() - 1);
    for (size_t ri = 0; ri < regions.size(); ++ri)
        for (int64_t r = regions[ri].r0; r < regions[ri].r1; ++r) rows[cur[(size_t)config(r) * regions.size() + ri]++] = (int32_t)r;
    // pieces of <= SEG_PIECE rows, cell by cell
    std::vector<int32_t> piece, piece_off(cells + 1, 0);
    for (size_t c = 0; c < cells; ++c) {
        for (int64_t r = g.off[c]; r < g.off[c + 1]; r += SEG_PIECE) {
            piece.push_back((int32_t)c);
            piece.push_back((int32_t)r);
            piece.push_back((int32_t)std::min<int64_t>(r + SEG_PIECE, g.off[c + 1]));
        }
        piece_off[c + 1] = (int32_t)(piece.size() / 3);
    }
    g.npieces = (int)(piece.size() / 3);
    pbn_ctx* ctx = sd->ctx;
    g.rows.alloc(rows.size() + 16);
    g.piece.alloc(piece.size() + 3);
    g.piece_off.alloc(piece_off.size());
    if (!rows.empty()) HIP_CHECK(hipMemcpyAsync(g.rows.p, rows.data(), rows.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    if (!piece.empty()) HIP_CHECK(hipMemcpyAsync(g.piece.p, piece.data(), piece.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(g.piece_off.p, piece_off.data(), piece_off.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));   // the host vectors go out of scope
    sd->groupings[key] = gp;
    return gp;
}

// ---- segmented Gram: pilot-shifted moments of D columns for EVERY (configuration, region) cell in one launch pair ---------
// (the batched per-configuration fit of SURVEY.md §7: replaces one Gram launch + sync per cell).  A block takes one piece of
// one cell: lanes stride over the piece, gather their rows through the grouped list and add sums and upper-triangle products
// into registers; fixed butterfly + wave order -> the piece's partial; a second kernel adds a cell's pieces in order.
// Deterministic, no atomics.  D <= 8 (variable + 7 continuous parents); wider candidates take the per-cell MFMA Gram.
struct SegArgs {
    const void* base;
    int64_t ld;
    int cols[8];
    double shift[8];
    const int32_t* rows;
    const int32_t* piece;      // [npieces][3]
    const int32_t* piece_off;  // [ncells + 1]
    int npieces, ncells;
    double* partial;           // [npieces][S]
    double* out;               // [ncells][S]
};

__device__ __forceinline__ double seg_wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

template <typename T, int D>
__global__ __launch_bounds__(256) void seg_moments_kernel(SegArgs a) {
    constexpr int S = D + D * (D + 1) / 2;
    __shared__ double red[4][S];
    const int pc = blockIdx.x;
    if (pc >= a.npieces) return;
    const int r0 = a.piece[3 * pc + 1], r1 = a.piece[3 * pc + 2];
    double acc[S];
#pragma unroll
    for (int i = 0; i < S; ++i) acc[i] = 0.0;
    const T* col[D];
#pragma unroll
    for (int i = 0; i < D; ++i) col[i] = (const T*)a.base + (int64_t)a.cols[i] * a.ld;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const int64_t row = a.rows[r];
        double x[D];
#pragma unroll
        for (int i = 0; i < D; ++i) x[i] = (double)col[i][row] - a.shift[i];
        int pos = D;
#pragma unroll
        for (int i = 0; i < D; ++i) {
            acc[i] += x[i];
#pragma unroll
            for (int j = i; j < D; ++j) { acc[pos] = __builtin_fma(x[i], x[j], acc[pos]); ++pos; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const double v = seg_wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < S) a.partial[(size_t)pc * S + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

__global__ __launch_bounds__(64) void seg_reduce_kernel(SegArgs a, int S) {
    const int cell = blockIdx.x / S, st = blockIdx.x - cell * S;
    if (cell >= a.ncells) return;
    const int b0 = a.piece_off[cell], b1 = a.piece_off[cell + 1];
    double v = 0.0;
    for (int b = b0 + (int)threadIdx.x; b < b1; b += 64) v += a.partial[(size_t)b * S + st];
    v = seg_wave_sum(v);
    if (threadIdx.x == 0) a.out[(size_t)cell * S + st] = v;
}

template <typename T>
void launch_seg(const SegArgs& a, int d, hipStream_t st) {
    const dim3 grid((unsigned)a.npieces), block(256);
    switch (d) {
        case 1: hipLaunchKernelGGL((seg_moments_kernel<T, 1>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((seg_moments_kernel<T, 2>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((seg_moments_kernel<T, 3>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((seg_moments_kernel<T, 4>), grid, block, 0, st, a); break;
        case 5: hipLaunchKernelGGL((seg_moments_kernel<T, 5>), grid, block, 0, st, a); break;
        case 6: hipLaunchKernelGGL((seg_moments_kernel<T, 6>), grid, block, 0, st, a); break;
        case 7: hipLaunchKernelGGL((seg_moments_kernel<T, 7>), grid, block, 0, st, a); break;
        default: hipLaunchKernelGGL((seg_moments_kernel<T, 8>), grid, block, 0, st, a); break;
    }
    HIP_CHECK(hipGetLastError());
}

// The moments of ALL continuous columns for every cell of a grouping, once: a hill-climb scores dozens of candidates over the same
// discrete parent set (C5: 1 671 hybrid evaluations over a few dozen groupings), each of which used to launch its own segmented
// moments and wait for them.  One gathered, segmented Gram (launch_gram_segments: gram_gring_kernel through the grouping's row list)
// gives every cell's n + n^2 numbers; a candidate's d x d moments are entries of them.
// Shifted moments of the columns `cols` (nc of them, <= 64) for EVERY cell of the grouping in one segmented, gathered Gram launch: a cell is a
// segment of the grouped row list cut into pieces of 4 096 rows, the pieces' partials added in order per cell (launch_gram_segments).
// out[cell]: S[i], G[i + j nc] indexed by position in `cols`.
static void cells_gram(pbn_scoredata* sd, const HybridGrouping& g, const int* cols, int nc, std::vector<Stats>& out) {
    pbn_ctx* ctx = sd->ctx;
    const pbn_table* t = sd->table();
    const size_t cells = (size_t)g.nc * g.nregions;
    const int nct = (nc + 15) / 16, WS = gram_ws(nct);
    constexpr int64_t PIECE = 4096;
    std::vector<int32_t> blk, off(cells + 1, 0);
    for (size_t c = 0; c < cells; ++c) {
        for (int64_t r = g.off[c]; r < g.off[c + 1]; r += PIECE) {
            const int32_t slot = (int32_t)(blk.size() / 4);
            blk.push_back((int32_t)c); blk.push_back((int32_t)r); blk.push_back((int32_t)std::min<int64_t>(r + PIECE, g.off[c + 1])); blk.push_back(slot);
        }
        off[c + 1] = (int32_t)(blk.size() / 4);
    }
    const int nblk = (int)(blk.size() / 4);
    out.assign(cells, Stats());
    for (size_t c = 0; c < cells; ++c) { out[c].zero(nc); out[c].N = g.off[c + 1] - g.off[c]; }
    if (nblk == 0) return;
    dev_buf<int32_t> dblk(blk.size() + off.size());
    HIP_CHECK(hipMemcpyAsync(dblk.p, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    HIP_CHECK(hipMemcpyAsync(dblk.p + blk.size(), off.data(), off.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
    ctx->scratch_red.reserve(((size_t)nblk + cells) * WS);
    double* partial = ctx->scratch_red.p;
    double* outd = partial + (size_t)nblk * WS;
    GramArgs a{};
    a.base = t->data; a.ld = t->ld; a.n_cols = nc; a.row0 = 0; a.rows = g.rows.p; a.n = g.off[cells];
    for (int i = 0; i < nc; ++i) a.gc.cols[i] = cols[i];
    a.rows_per_block = PIECE; a.blk = dblk.p; a.shift = sd->shift_dev.p /* by table column */; a.partial = partial; a.num_cus = ctx->num_cus;
    { KernelTimer kt(ctx, PBN_K_GRAM); launch_gram_segments(a, t->dtype, nblk, dblk.p + blk.size(), (int)cells, outd, ctx->stream); }
    std::vector<double> h(cells * (size_t)WS);
    HIP_CHECK(hipMemcpyAsync(h.data(), outd, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (size_t c = 0; c < cells; ++c) {
        const double* w = h.data() + c * (size_t)WS;
        Stats& st = out[c];
        for (int i = 0; i < nc; ++i) st.S[i] = w[WS - nct * 16 + i];
        int pr = 0;
        for (int I = 0; I < nct; ++I)
            for (int J = I; J < nct; ++J, ++pr) {
                const double* tile = w + (size_t)pr * 256;
                for (int e = 0; e < 256; ++e) {
                    const int reg = e >> 6, lane = e & 63;
                    const int r = I * 16 + (lane >> 4) + 4 * reg, cc = J * 16 + (lane & 15);
                    if (r >= nc || cc >= nc || (I == J && r > cc)) continue;
                    st.G[r + (size_t)cc * nc] = tile[e];
                    st.G[cc + (size_t)r * nc] = tile[e];
                }
            }
    }
}

static bool ensure_full_moments(pbn_scoredata* sd, const HybridGrouping& g) {
    if (g.full_state != 0) return g.full_state > 0;
    static const bool on = PBN_TUNE(HYBRID_FULLMOMENTS, 1) != 0;
    const int n = sd->n;
    const size_t cells = (size_t)g.nc * g.nregions;
    // (not with validity masks: a row that is null in another column still counts for this candidate's own)
    if (!on || sd->has_nulls || n > 64 || cells * ((size_t)n * n + n) > ((size_t)1 << 24)) { g.full_state = -1; return false; }
    std::vector<int> all((size_t)n);
    for (int i = 0; i < n; ++i) all[i] = i;
    cells_gram(sd, g, all.data(), n, g.full);
    g.full_state = 1;
    return true;
}

// moments of columns `cols` (d) for every cell of the grouping: M[cell], cell = configuration * nregions + region
void group_moments(pbn_scoredata* sd, const HybridGrouping& g, const int* cols, int d, std::vector<Stats>& M) {
    const size_t cells = (size_t)g.nc * g.nregions;
    M.assign(cells, Stats());
    for (size_t c = 0; c < cells; ++c) { M[c].zero(d); M[c].N = g.off[c + 1] - g.off[c]; }
    if (ensure_full_moments(sd, g)) {
        const int n = sd->n;
        for (size_t c = 0; c < cells; ++c) {
            const Stats& f = g.full[c];
            Stats& st = M[c];
            for (int j = 0; j < d; ++j) {
                st.S[j] = f.S[cols[j]];
                for (int i = 0; i < d; ++i) st.G[i + (size_t)j * d] = f.G[cols[i] + (size_t)cols[j] * n];
            }
        }
        return;
    }
    pbn_ctx* ctx = sd->ctx;
    const pbn_table* t = sd->table();
    static const bool batched = PBN_TUNE(HYBRID_SEGMENTED, 1) != 0;
    if (d <= 8 && batched && g.npieces > 0) {
        const int S = d + d * (d + 1) / 2;
        ctx->scratch_red.reserve((size_t)(g.npieces + cells) * S);
        SegArgs a{};
        a.base = t->data; a.ld = t->ld;
        for (int i = 0; i < d; ++i) { a.cols[i] = cols[i]; a.shift[i] = sd->shift[cols[i]]; }
        a.rows = g.rows.p; a.piece = g.piece.p; a.piece_off = g.piece_off.p; a.npieces = g.npieces; a.ncells = (int)cells;
        a.partial = ctx->scratch_red.p; a.out = ctx->scratch_red.p + (size_t)g.npieces * S;
        {
            KernelTimer kt(ctx, PBN_K_GRAM);
            if (t->dtype == PBN_F64) launch_seg<double>(a, d, ctx->stream); else launch_seg<float>(a, d, ctx->stream);
            hipLaunchKernelGGL(seg_reduce_kernel, dim3((unsigned)(cells * S)), dim3(64), 0, ctx->stream, a, S);
            HIP_CHECK(hipGetLastError());
        }
        std::vector<double> h(cells * S);
        HIP_CHECK(hipMemcpyAsync(h.data(), a.out, h.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (size_t c = 0; c < cells; ++c) {
            const double* o = h.data() + c * S;
            Stats& st = M[c];
            int pos = d;
            for (int i = 0; i < d; ++i) {
                st.S[i] = o[i];
                for (int j = i; j < d; ++j) { st.G[i + (size_t)j * d] = st.G[j + (size_t)i * d] = o[pos]; ++pos; }
            }
        }
        return;
    }
    // more than 8 columns (or the segmented register kernel switched off): the candidate's own columns through the segmented MFMA Gram - one
    // launch for all cells (round 3 took one Gram launch + one wait per cell here)
    static const bool cellwise = PBN_TUNE(HYBRID_CELLWISE_GRAM, 0) != 0;
    if (!cellwise && d <= 64) {
        cells_gram(sd, g, cols, d, M);
        return;
    }
    for (size_t c = 0; c < cells; ++c)
        if (M[c].N > 0) gram_raw(t, cols, d, 0, M[c].N, g.rows.p + g.off[c], sd->shift_dev.p, M[c].S.data(), M[c].G.data());
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

struct Term { double value = 0; int slot = -1; };
struct Slice { Term joint, marg; bool has_marg; int part; };
// what a check-after redo needs, flat (one candidate makes ~100 terms, a search thousands of candidates: no per-term allocations): the
// term's bandwidth and centre in `rstore`, its columns in `rcols`
struct Redo { int slot, nv; size_t off, coff; int64_t N, r0, n0, r1, te0, nte; const int32_t* rows; };

}  // namespace

// The CKDE candidates of one pbn_score_batch call that are in flight together (hill-climbing asks for the cells of a delta-cache update in
// one call): their grouped pools join ONE kde_group_run - one keys / sort / pack / prepass / sweep / finish chain per arena-full of pools
// instead of one per candidate - their small slices keep going round the issue lanes, and the host waits ONCE, at flush().  Result slots
// are numbered across the batch; `scheduled` lets a later candidate of the batch share a term an earlier one already enqueued (the marginal
// A({y}) of every child of y, the joint of x | {y} and y | {x}) exactly as the set-function cache does between batches.
struct HybridBatch {
    pbn_scoredata* sd;
    pbn_ctx* ctx;
    bool f32 = false;          // fp32 table with the check-after on: max-norm slots behind the sums
    bool active = false;       // slots zeroed, candidates may enqueue
    size_t capacity = 0;       // result slots of the batch (sums; the max-norms start at dsums + capacity)
    double* dsums = nullptr;
    double* dmax = nullptr;
    int lanes = 1;
    GroupBatch gb;
    std::vector<std::vector<int>> slot_key;        // [slot] -> set-function cache key
    std::map<std::vector<int>, int> scheduled;     // key -> slot, terms enqueued by this batch
    std::vector<Redo> redo_info;
    std::vector<double> rstore;
    std::vector<int> rcols;
    struct Job { std::vector<Slice> slices; bool has_parts; HybridParts parts; bool has_sink; HybridSink sink; };
    std::vector<Job> jobs;
    std::vector<std::shared_ptr<const HybridGrouping>> keep;   // groupings whose device row lists the enqueued pools / slices / redo records read
    double last_value = 0;     // score of the last finished job (the synchronous form of score_hybrid)

    HybridBatch(pbn_scoredata* s, bool f32_) : sd(s), ctx(s->ctx), f32(f32_) {}
    ~HybridBatch() {           // abandoned with work in flight (an exception on the way): nothing may outlive the scratch it writes
        if (active) {
            ctx->sync_lanes(pbn_ctx::MAX_PARKED);
            (void)hipStreamSynchronize(ctx->stream);
            ctx->drop_staged();
        }
    }
    // room for a candidate of up to max_slots terms (finishing what is in flight when the slots run out)
    // -> true when the slots were zeroed just now (on the context's stream: the issue lanes have to wait for that, and for nothing else)
    bool begin_candidate(size_t max_slots) {
        if (active && slot_key.size() + max_slots > capacity) flush();
        const bool fresh = !active;
        if (!active) {
            static const size_t min_slots = (size_t)std::max(1ll, knob_ll("PBN_HYBRID_BATCH_SLOTS", 1ll << 15));
            capacity = std::max(std::max(capacity, max_slots), min_slots);
            ctx->scratch_sums.reserve(2 * capacity);   // (grow-only; nothing of this batch is in flight here)
            dsums = ctx->scratch_sums.p;
            dmax = f32 ? dsums + capacity : nullptr;
            HIP_CHECK(hipMemsetAsync(dsums, 0, 2 * capacity * sizeof(double), ctx->stream));
            active = true;
        }
        return fresh;
    }
    std::vector<double> deal_load;   // jobs with one process per GPU: cost dealt to every rank so far (score_hybrid: `owned`)
    size_t pool_bytes = 0;     // arena bytes of the pools collected since the last chain
    // the collected pools' chain, enqueued without waiting for it: the device works on it while the host prepares the next candidates
    void kick() {
        if (gb.pools.empty()) return;
        kde_group_run(ctx, sd->table(), gb, dsums, dmax);   // on the context's own stream, next to the lanes' per-slice chains
        gb = GroupBatch{};
        pool_bytes = 0;
    }
    void flush();
};

void HybridBatch::flush() {
    if (!active) return;
    const pbn_table* t = sd->table();
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    kick();
    const size_t ns = slot_key.size();
    std::vector<double> hs(std::max<size_t>(1, ns)), hmax(f32 ? ns : 0);
    if (lanes > 1) ctx->sync_lanes(lanes - 1);
    if (ns) HIP_CHECK(hipMemcpyAsync(hs.data(), dsums, ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    if (f32 && ns) HIP_CHECK(hipMemcpyAsync(hmax.data(), dmax, ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_CHECK(hipStreamSynchronize(ctx->stream));
    const bool f64 = sd->dtype == PBN_F64;
    if (f32 || f64) {
        // check-after (score_hybrid), through the per-slice chain.  fp32 tables: evaluations whose training rows reach beyond what fp32
        // fragments hold, once more on fp64 fragments; fp64 tables: sums that are a cancellation to ~0 (kde_sum_needs_precision), once more
        // at the accuracy of the per-row path
        bool any = false;
        for (const Redo& r : redo_info) {
            if (f32 ? !kde_wants_widening(hmax[(size_t)r.slot], r.nv) : !kde_sum_needs_precision(hs[(size_t)r.slot], r.nte)) continue;
            const int* v = rcols.data() + r.coff;
            KdeModel m;
            kde_prepare(m, sd->dtype, r.nv, r.N, rstore.data() + r.off, PBN_BW_FULL, false, rstore.data() + r.off + (size_t)r.nv * r.nv);
            if (f32) kde_widen(m);
            else ++sd->precise_redos;
            const KdePackBytes pb = kde_pack_bytes(m.fdtype(), m.dm, false, m.N);
            ctx->scratch_train.reserve(align(pb.apack) + align(pb.nxpack) + 256);
            m.Apack = ctx->scratch_train.p;
            m.nxpack = ctx->scratch_train.p + align(pb.apack);
            m.Axpack = nullptr;
            HIP_CHECK(hipMemsetAsync(dsums + r.slot, 0, sizeof(double), ctx->stream));
            kde_pack_train(ctx, m, t, v, r.r0, r.n0, r.r1, r.rows, /*prune=*/true);
            kde_eval_enqueue(ctx, m, t, v, r.te0, r.nte, nullptr, dsums + r.slot, r.rows, nullptr, /*precise=*/f64);
            ++sd->kde_sweeps;
            any = true;
        }
        if (any) {
            HIP_CHECK(hipMemcpyAsync(hs.data(), dsums, ns * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
        }
    }
    ctx->drop_staged();
    active = false;
    for (size_t i = 0; i < ns; ++i) sd->kde_cache[slot_key[i]] = hs[i];
    for (const Job& j : jobs) {
        // per-part sums in slice order, then the parts in part order: what a job with one process per GPU adds up, too
        double pacc[PBN_HYBRID_PARTS] = {};
        for (const Slice& sl : j.slices) {
            const double jv = sl.joint.slot >= 0 ? hs[sl.joint.slot] : sl.joint.value;
            const double mv = !sl.has_marg ? 0.0 : (sl.marg.slot >= 0 ? hs[sl.marg.slot] : sl.marg.value);
            pacc[sl.part] += jv - mv;
        }
        double acc = 0;
        for (int q = 0; q < PBN_HYBRID_PARTS; ++q) acc += pacc[q];
        if (j.has_parts)
            for (int q = 0; q < PBN_HYBRID_PARTS; ++q) j.parts.out[q] = pacc[q];
        if (j.has_sink) {
            *j.sink.out = acc;
            if (!j.sink.memo_key.empty()) sd->score_memo[j.sink.memo_key] = acc;
        }
        last_value = acc;
    }
    jobs.clear();
    keep.clear();              // (everything that read them has finished: lanes and stream were waited for above)
    slot_key.clear();
    scheduled.clear();
    redo_info.clear();
    rstore.clear();
    rcols.clear();
    lanes = 1;
}

HybridBatch* hybrid_batch_begin(pbn_scoredata* sd) {
    static const bool check_after = PBN_TUNE(F32_CHECK, 1) != 0;   // 0: measurement only
    return new HybridBatch(sd, sd->dtype == PBN_F32 && check_after);
}
void hybrid_batch_flush(HybridBatch* hb) { if (hb) hb->flush(); }
void hybrid_batch_end(HybridBatch* hb) noexcept { delete hb; }

double score_hybrid(pbn_scoredata* sd, int kind, int var, int node_type, const int* parents, int p, const HybridParts* parts, HybridBatch* hb,
                    const HybridSink* sink, bool* deferred) {
    if (deferred) *deferred = false;
    const int n = sd->n;
    if (parts && (node_type != PBN_NODE_CKDE || var >= n || (kind != PBN_SCORE_CVLIK && kind != PBN_SCORE_HOLDOUT)))
        throw invalid_error("pbn_score_batch_parts: CKDE candidates of a likelihood score only");
    if (kind == PBN_SCORE_BGE) throw invalid_error("BGe is not defined for networks with discrete variables.");
    if (var >= n) {
        if (node_type != PBN_NODE_DISCRETE) throw invalid_error("pbn_score_batch: discrete column scored with a continuous node type");
        return score_discrete(sd, kind, var, parents, p);
    }
    if (node_type == PBN_NODE_DISCRETE) throw invalid_error("pbn_score_batch: continuous column scored as DiscreteFactor");
    std::vector<int> dpar, cols{var};
    for (int i = 0; i < p; ++i) (parents[i] >= n ? dpar : cols).push_back(parents[i]);
    // continuous parents in ascending order (the discrete ones are canonicalised below): the score of (variable | parent SET) is then
    // one number down to the last bit - the memo returns what a fresh evaluation would, and a job that shares the slices over its ranks
    // (pbn_score_batch_parts) adds up the same doubles
    std::sort(cols.begin() + 1, cols.end());
    std::sort(dpar.begin(), dpar.end());   // and the configurations in the canonical numbering: the slices are visited - and added - in one order
    const int d = (int)cols.size(), pc = d - 1;
    // the JOINT term of a CKDE slice is evaluated over all of its columns in ascending order - x | {y} + D and y | {x} + D share it, and
    // the shared value must not depend on who asked first; jperm[i] = index in `cols` of the i-th smallest column
    std::vector<int> jperm(d), jcols(d);
    {
        const int vpos = (int)(std::lower_bound(cols.begin() + 1, cols.end(), var) - (cols.begin() + 1));   // parents below the variable
        for (int i = 0; i < d; ++i) jperm[i] = i < vpos ? i + 1 : (i == vpos ? 0 : i);
        for (int i = 0; i < d; ++i) jcols[i] = cols[jperm[i]];
    }
    if (d > 64) throw invalid_error("pbn_score_batch: more than 63 continuous parents in one hybrid candidate (the per-cell Gram takes 64 columns)");
    pbn_ctx* ctx = sd->ctx;
    const pbn_table* t = sd->table();
    std::vector<Region> regions = regions_of(sd, kind);
    const int R = (int)regions.size();
    // canonical grouping (parents sorted) and the map from the candidate's configuration numbering onto it
    std::vector<int> dsorted(dpar);
    std::sort(dsorted.begin(), dsorted.end());
    const std::shared_ptr<const HybridGrouping> g_hold = grouping_for(sd, kind, dsorted, regions);
    const HybridGrouping& g = *g_hold;
    std::vector<int> canon(g.nc);
    {
        std::vector<int> cstride(dsorted.size(), 1);
        int acc = 1;
        for (size_t i = 0; i < dsorted.size(); ++i) { cstride[i] = acc; acc *= sd->card[dsorted[i] - n]; }
        for (int c = 0; c < g.nc; ++c) {   // c in evidence order: stride_0 = 1, stride_i = stride_{i-1} * card_{i-1}
            int rest = c, cc = 0;
            for (size_t i = 0; i < dpar.size(); ++i) {
                const int card = sd->card[dpar[i] - n], digit = rest % card;
                rest /= card;
                const size_t pos = std::lower_bound(dsorted.begin(), dsorted.end(), dpar[i]) - dsorted.begin();
                cc += digit * cstride[pos];
            }
            canon[c] = cc;
        }
    }
    auto cell = [&](int c, int r) { return (size_t)canon[c] * R + r; };
    std::vector<Stats> M;
    group_moments(sd, g, cols.data(), d, M);
    std::vector<double> mu(d), sse((size_t)d * d), beta(d), H((size_t)d * d);
    // double-double refit of an ill-conditioned slice from its rows (lg_accurate.hip): the rows of configuration c are one
    // block of the grouped list, region u two ranges around its own rows
    auto refit = [&](int c, int u, int64_t ntrain, double* b) {
        const int64_t base = g.off[cell(c, 0)], end = g.off[cell(c, R - 1) + 1];
        if (u < 0) return lg_fit_accurate(t, cols.data(), d, base, ntrain, 0, ntrain, g.rows.p, b);   // one region: [base, base + ntrain)
        const int64_t f0 = g.off[cell(c, u)], f1 = g.off[cell(c, u) + 1];
        (void)end;
        return lg_fit_accurate(t, cols.data(), d, base, f0 - base, f1, ntrain, g.rows.p, b);
    };

    if (kind == PBN_SCORE_BIC) {  // bic.cpp:29-64
        if (node_type != PBN_NODE_LG) throw invalid_error("BIC: only LinearGaussianCPD / DiscreteFactor node types are implemented");
        double loglik = 0;
        int64_t valid = 0;
        for (int c = 0; c < g.nc; ++c) {
            const Stats& st = M[cell(c, 0)];
            valid += st.N;
            if (st.N == 0) continue;
            local_moments(sd, st, cols.data(), d, mu.data(), sse.data());
            bool suspect = false;
            double v = lg_fit(st.N, pc, mu.data(), sse.data(), beta.data(), &suspect);
            if (suspect && lg_guard_on() && d <= 16) v = refit(c, -1, st.N, beta.data());
            if (v < MACHINE_TOL || std::isinf(v)) return -INF;
            const double nv = (double)st.N;
            loglik += 0.5 * (1 + (double)pc - nv) - 0.5 * nv * LOG_2PI - nv * 0.5 * std::log(v);
        }
        return loglik - std::log((double)valid) * 0.5 * g.nc * (pc + 2);
    }

    // likelihood scores (cv_likelihood.cpp:11-25 / holdout_likelihood.cpp:14-23 over DiscreteAdaptator factors)
    const bool cv = kind == PBN_SCORE_CVLIK;
    const int units = cv ? R : 1;
    std::vector<Stats> allc(g.nc);
    if (cv)
        for (int c = 0; c < g.nc; ++c) {
            allc[c].zero(d);
            for (int f = 0; f < R; ++f) allc[c].add(M[cell(c, f)]);
        }
    double acc = 0;
    Stats train;
    // CKDE slices: slogl = A(joint set) - A(parent set), A = sum of log KDE over the slice's test rows (CKDE.hpp:256-287), each
    // from a plain sweep pruned on its own box and remembered by [region, dimension of the bandwidth rule, variable set |
    // kind, discrete parents, configuration]: the joint of x | {y} + D and of y | {x} + D is the same number, the marginal
    // A({y}) serves every child of y under D.  All sweeps of the candidate are enqueued back to back - gather lists are
    // device resident - with ONE synchronisation at the end.
    std::vector<Slice> slices;
    // fp32 tables: behind the sums, one slot per sum for |z|^2 of the farthest whitened training row of its evaluation (reported by the
    // pack kernels) - an evaluation that kde_wants_widening() flags is redone on fp64 fragments before its value is used (decided per
    // evaluation, nothing remembered per column set: scoring.hip does the same for plain terms)
    const size_t max_slots = (size_t)units * g.nc * 2 + 1;
    // the batch this candidate's evaluations join: the caller's (its score arrives at flush()), or one of its own, flushed below
    std::unique_ptr<HybridBatch> own;
    if (node_type == PBN_NODE_CKDE && !hb) { own.reset(hybrid_batch_begin(sd)); hb = own.get(); }
    const bool f32 = node_type == PBN_NODE_CKDE && hb->f32;
    struct { double* p = nullptr; } dsums;   // slots in the context's scratch_sums (grow-only; no allocation per candidate)
    bool fresh_slots = false;
    if (node_type == PBN_NODE_CKDE) {
        fresh_slots = hb->begin_candidate(max_slots);
        dsums.p = hb->dsums;
        if (hb->keep.empty() || hb->keep.back() != g_hold) hb->keep.push_back(g_hold);   // (after a flush begin_candidate may have made: `keep` is cleared there)
    }
    double* const dmax = f32 ? hb->dmax : nullptr;
    std::vector<double> Hterm, muterm;   // bandwidth / centre of the term being built (reused)
    // the slices' sweeps go round-robin over the context's issue lanes (common.hpp): a sweep's tail overlaps the next slice
    const int lanes = (node_type == PBN_NODE_CKDE && !ctx->profiling) ? score_lanes(t->n_rows) : 1;
    if (lanes > 1) {
        ctx->ensure_lanes(lanes - 1);
        // (not per candidate: a lane that waited for the stream again would also wait for the grouped chains handed over since)
        if (fresh_slots || lanes > hb->lanes) ctx->lanes_wait_for_stream(lanes - 1);
        hb->lanes = std::max(hb->lanes, lanes);
    }
    size_t issued = 0;
    auto key_of = [&](int region, int c, const int* v, int nv) {
        std::vector<int> k(v, v + nv);
        std::sort(k.begin(), k.end());
        k.insert(k.begin(), {region, d});
        k.push_back(-1);
        k.push_back(kind);
        k.insert(k.end(), dsorted.begin(), dsorted.end());
        k.push_back(-2);
        k.push_back(canon[c]);
        return k;
    };
    auto align = [](size_t x) { return (x + 255) / 256 * 256; };
    // Grouped evaluation (kde_group.hip): the slices of one configuration and one term (joint / marginal variable set) share a
    // POOL - the configuration's block of the grouped row list, its regions the folds (or hold-out train / test) - and are
    // evaluated by one launch chain for the whole candidate instead of one chain per (fold, configuration, term).  A pool takes
    // this path when every training slice of the configuration is large enough for the pruned sweeps.
    GroupBatch gb;
    std::vector<std::vector<GUnit>> pool_units;
    std::vector<int> pool_of((size_t)g.nc * 2, -1);   // (configuration, which) -> pool
    std::vector<char> elig((size_t)g.nc * 2, 0);
    if (node_type == PBN_NODE_CKDE) {
        for (int c = 0; c < g.nc; ++c) {
            int64_t min_train = -1;
            for (int u = 0; u < units; ++u) {
                const int64_t ntr = cv ? allc[c].N - M[cell(c, u)].N : M[cell(c, 0)].N;
                const int64_t nte = cv ? M[cell(c, u)].N : M[cell(c, 1)].N;
                if (ntr <= 1 || nte == 0) continue;
                min_train = min_train < 0 ? ntr : std::min(min_train, ntr);
            }
            for (int which = 0; which < (pc > 0 ? 2 : 1); ++which)
                elig[(size_t)c * 2 + which] = min_train > 0 && kde_group_applies(sd->dtype, d - which, min_train, R);
        }
    }
    // v: the term's columns, colidx: their indices in `cols` (the joint in ascending order, the marginal = cols[1:])
    auto group_unit = [&](int c, int u, int which, const KdeModel& m, const int* v, const int* colidx, int nv, int64_t ntrain, int64_t ntest, int slot) {
        int& pi = pool_of[(size_t)c * 2 + which];
        if (pi < 0) {
            pi = (int)gb.pools.size();
            GPool P{};
            const int64_t base = g.off[cell(c, 0)];
            P.rows = g.rows.p; P.row_base = base;
            P.n = (int32_t)(g.off[cell(c, R - 1) + 1] - base);
            P.R = R; P.d = nv; P.kd = std::min(nv, 4);
            for (int r = 0; r < R; ++r) P.rb[r] = (int32_t)(g.off[cell(c, r)] - base);
            P.rb[R] = P.n;
            for (int r = 0; r < PBN_GROUP_MAX_R; ++r) P.test_unit[r] = -1;
            for (int i = 0; i < nv; ++i) P.cols[i] = v[i];
            // pool-level standardisation for the Morton keys: the configuration's own covariance of the term's columns (training
            // part for the hold-out score); a singular one falls back to the diagonal - the keys only order the rows
            const Stats& all = cv ? allc[c] : M[cell(c, 0)];
            std::vector<double> gm(d), gs((size_t)d * d), sub((size_t)nv * nv), L((size_t)nv * nv), Li((size_t)nv * nv);
            local_moments(sd, all, cols.data(), d, gm.data(), gs.data());
            for (int j = 0; j < nv; ++j)
                for (int i = 0; i < nv; ++i) sub[i + (size_t)j * nv] = gs[colidx[i] + (size_t)colidx[j] * d] / (double)std::max<int64_t>(1, all.N - 1);
            if (!hm::cholesky(sub.data(), nv, L.data())) {
                std::fill(L.begin(), L.end(), 0.0);
                for (int i = 0; i < nv; ++i) L[i + (size_t)i * nv] = std::sqrt(std::max(sub[i + (size_t)i * nv], 1e-300));
            }
            hm::lower_inverse(L.data(), nv, Li.data());
            for (int i = 0; i < nv; ++i) {
                P.mug[i] = gm[colidx[i]];
                for (int j = 0; j < nv; ++j) P.Wg[i * nv + j] = j <= i ? Li[i + (size_t)j * nv] : 0.0;
            }
            gb.pools.push_back(P);
            pool_units.emplace_back();
        }
        GUnit U{};
        U.test_region = cv ? u : 1;
        U.train_mask = cv ? (((R >= 64) ? ~0ull : ((1ull << R) - 1ull)) & ~(1ull << u)) : 1ull;
        U.N = (int32_t)ntrain; U.nq = (int32_t)ntest;
        U.lognorm = m.lognorm;
        for (int i = 0; i < nv * nv; ++i) U.W[i] = m.W[i];
        for (int i = 0; i < nv; ++i) U.mu[i] = m.mu[i];
        U.sum_slot = slot;
        gb.pools[pi].test_unit[U.test_region] = (int32_t)pool_units[pi].size();
        pool_units[pi].push_back(U);
    };
    // the parts this call evaluates (scoring_internal.hpp): all of them, or those dealt to one rank of a job
    unsigned long long owned = ~0ull;
    if (parts && parts->world > 1) {
        double cost[PBN_HYBRID_PARTS] = {};
        for (int u = 0; u < units; ++u)
            for (int c = 0; c < g.nc; ++c) {
                const double ntr = (double)(cv ? allc[c].N - M[cell(c, u)].N : M[cell(c, 0)].N);
                const double nte = (double)(cv ? M[cell(c, u)].N : M[cell(c, 1)].N);
                // training rows a query meets, joint + marginal term: all of them in an unpruned sweep; in a pruned one the rows within the
                // margin of the query, ~ N h^d = N^(4 / (d + 4)) under the normal-reference bandwidth (constants: profiles/r4/prune_visits.txt)
                auto met = [&](int dd, int which) {
                    if (dd < 1) return 0.0;
                    static const double cd[5] = {0, 4.6, 18.0, 67.0, 243.0};
                    if (dd > 4 || !elig[(size_t)c * 2 + which]) return ntr;
                    return std::min(ntr, cd[dd] * std::pow(ntr, 4.0 / (dd + 4.0)));
                };
                cost[((int64_t)c * units + u) % PBN_HYBRID_PARTS] += nte * (met(d, 0) + met(pc, 1));
            }
        int order[PBN_HYBRID_PARTS];
        for (int q = 0; q < PBN_HYBRID_PARTS; ++q) order[q] = q;
        std::stable_sort(order, order + PBN_HYBRID_PARTS, [&](int a, int b) { return cost[a] > cost[b]; });
        // the ranks' loads carry over from the batch's earlier candidates: ten equal folds dealt to eight ranks leave two of them with
        // double work, and the next candidate's parts then go to the other six first
        std::vector<double>& load = hb->deal_load;
        if (load.size() != (size_t)parts->world) load.assign((size_t)parts->world, 0.0);
        owned = 0;
        for (int i = 0; i < PBN_HYBRID_PARTS; ++i) {
            const int q = order[i];
            int best = 0;
            for (int r = 1; r < parts->world; ++r)
                if (load[r] < load[best]) best = r;
            load[best] += cost[q];
            if (best == parts->rank) owned |= 1ull << q;
        }
    }
    for (int u = 0; u < units; ++u) {
        for (int c = 0; c < g.nc; ++c) {
            const Stats* tr;
            const Stats* te;
            if (cv) { stats_minus(allc[c], M[cell(c, u)], train); tr = &train; te = &M[cell(c, u)]; }
            else { tr = &M[cell(c, 0)]; te = &M[cell(c, 1)]; }
            // the slice's part: folds of one configuration are consecutive parts
            const int part = (int)(((int64_t)c * units + u) % PBN_HYBRID_PARTS);
            if (!((owned >> part) & 1ull)) continue;
            if (tr->N == 0) continue;  // empty training slice -> no factor (DiscreteAdaptator.hpp:266-268)
            local_moments(sd, *tr, cols.data(), d, mu.data(), sse.data());
            if (node_type == PBN_NODE_LG) {
                bool suspect = false;
                double v = lg_fit(tr->N, pc, mu.data(), sse.data(), beta.data(), &suspect);
                const bool refitted = suspect && lg_guard_on() && d <= 16;
                if (refitted) v = refit(c, cv ? u : -1, tr->N, beta.data());
                if (v < MACHINE_TOL || std::isinf(v)) continue;  // LinearGaussianFitter -> nullptr
                if (te->N == 0) continue;
                if (refitted)   // huge coefficients: the test rows' log-likelihood from the rows, not from their moments
                    acc += lg_slogl_from_rows(t, cols.data(), d, g.off[cell(c, cv ? u : 1)], te->N, g.rows.p, beta.data(), v);
                else
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
            // rows: configuration block [base, ...) of the grouped list; training = everything of it but region u (CV) or
            // region 0 (hold-out), test = region u / region 1
            const int64_t base = g.off[cell(c, 0)];
            int64_t tr_row0 = base, tr_n0 = tr->N, tr_row1 = 0, te0 = g.off[cell(c, 1)];
            if (cv) { tr_n0 = g.off[cell(c, u)] - base; tr_row1 = g.off[cell(c, u) + 1]; te0 = g.off[cell(c, u)]; }
            Slice sl{{}, {}, pc > 0, part};
            std::vector<int> midx(std::max(pc, 1));
            for (int i = 0; i < pc; ++i) midx[i] = i + 1;
            std::vector<double>& Ht = Hterm;
            std::vector<double>& mut = muterm;
            for (int which = 0; which < (pc > 0 ? 2 : 1); ++which) {   // 0: joint over its columns in ascending order, 1: marginal over cols[1:]
                Term& term = which ? sl.marg : sl.joint;
                const int* v = which ? cols.data() + 1 : jcols.data();
                const int* colidx = which ? midx.data() : jperm.data();
                const int nv = d - which;
                const std::vector<int> key = key_of(u, c, v, nv);
                auto itc = sd->kde_cache.find(key);
                if (itc != sd->kde_cache.end()) { term.value = itc->second; continue; }
                auto its = hb->scheduled.find(key);   // enqueued by an earlier candidate of the batch: its slot
                if (its != hb->scheduled.end()) { term.slot = its->second; continue; }
                KdeModel m;
                Ht.resize((size_t)nv * nv); mut.resize((size_t)nv);
                try {
                    if (which) {
                        for (int jj = 0; jj < pc; ++jj) {
                            mut[jj] = mu[jj + 1];
                            for (int ii = 0; ii < pc; ++ii) Ht[ii + (size_t)jj * pc] = H[(ii + 1) + (size_t)(jj + 1) * d];
                        }
                    } else {
                        for (int jj = 0; jj < d; ++jj) {
                            mut[jj] = mu[jperm[jj]];
                            for (int ii = 0; ii < d; ++ii) Ht[ii + (size_t)jj * d] = H[jperm[ii] + (size_t)jperm[jj] * d];
                        }
                    }
                    kde_prepare(m, sd->dtype, nv, tr->N, Ht.data(), PBN_BW_FULL, false, mut.data());
                } catch (const singular_error&) {
                    term.slot = -2;   // no factor for this slice
                    break;
                }
                term.slot = (int)hb->slot_key.size();
                hb->slot_key.push_back(key);
                hb->scheduled.emplace(key, term.slot);
                if (f32 || sd->dtype == PBN_F64) {
                    hb->redo_info.push_back(Redo{term.slot, nv, hb->rstore.size(), hb->rcols.size(), tr->N, tr_row0, tr_n0, tr_row1, te0, te->N, g.rows.p});
                    hb->rstore.insert(hb->rstore.end(), Ht.begin(), Ht.end());
                    hb->rstore.insert(hb->rstore.end(), mut.begin(), mut.end());
                    hb->rcols.insert(hb->rcols.end(), v, v + nv);
                }
                if (elig[(size_t)c * 2 + which]) {   // evaluated with the candidate's other grouped slices, after the loops
                    group_unit(c, u, which, m, v, colidx, nv, tr->N, te->N, term.slot);
                    ++sd->kde_sweeps;
                    continue;
                }
                const KdePackBytes pb = kde_pack_bytes(m.fdtype(), m.dm, false, tr->N);
                LaneSwitch lane(ctx, (int)(issued++ % (size_t)lanes));   // before the lane's scratch is touched
                ctx->scratch_train.reserve(align(pb.apack) + align(pb.nxpack) + 256);
                char* arena = ctx->scratch_train.p;
                m.Apack = arena;
                m.nxpack = arena + align(pb.apack);
                m.Axpack = nullptr;
                kde_pack_train(ctx, m, t, v, tr_row0, tr_n0, tr_row1, g.rows.p, /*prune=*/true, dmax ? dmax + term.slot : nullptr);
                kde_eval_enqueue(ctx, m, t, v, te0, te->N, nullptr, dsums.p + term.slot, g.rows.p);
                ++sd->kde_sweeps;
            }
            if (sl.joint.slot == -2 || sl.marg.slot == -2) continue;
            slices.push_back(sl);
        }
    }
    if (node_type != PBN_NODE_CKDE) return acc;
    for (size_t pi = 0; pi < gb.pools.size(); ++pi) {   // the candidate's pools behind the batch's
        gb.pools[pi].unit0 = (int32_t)hb->gb.units.size();
        gb.pools[pi].nunits = (int32_t)pool_units[pi].size();
        hb->gb.units.insert(hb->gb.units.end(), pool_units[pi].begin(), pool_units[pi].end());
        hb->gb.pools.push_back(gb.pools[pi]);
        hb->pool_bytes += kde_group_pool_bytes(hb->gb, hb->gb.pools.back());
    }
    if (hb->pool_bytes >= kde_group_arena_budget()) hb->kick();   // an arena-full: one chain's worth
    HybridBatch::Job job{std::move(slices), parts != nullptr, parts ? *parts : HybridParts{0, 0, nullptr}, sink != nullptr, sink ? *sink : HybridSink{nullptr, {}}};
    hb->jobs.push_back(std::move(job));
    if (own) {   // synchronous form: this candidate alone
        own->flush();
        return own->last_value;
    }
    if (deferred) *deferred = true;
    return 0.0;
}

}  // namespace score
}  // namespace pbn
