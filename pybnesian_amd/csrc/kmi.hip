// k-nearest-neighbour conditional mutual information on rank-transformed data with permutation p-values
// (learning/independences/continuous/mutual_information.{hpp,cpp}: KMutualInformation; kdtree/kdtree.hpp supplies the
// neighbour queries there).  SURVEY.md §8 f1.
//
// The reference ranks every column (0 .. N-1), finds for every row i the Chebyshev distance eps_i to its k-th neighbour
// in the joint (x, y, z) rank space with a kd-tree, counts the rows strictly inside eps_i in the (x, z), (y, z) and z
// subspaces with another kd-tree walk, and averages digammas of the counts (Frenzel-Pompe / KSG estimator).  The trees
// only accelerate exact set counts, so here both steps are brute-force kernels over all N^2 pairs on integer ranks - every
// thread owns one row, the other rows stream through LDS tiles - which is the same pairwise-distance shape as the KDE
// sweep and gives bit-identical counts.  The permutation p-values keep the reference's host procedure call for call
// (std::mt19937, std::shuffle, std::uniform_real_distribution<float>, std::sort of libstdc++, which this library is
// built against too); each permuted sample costs one upload of the permuted x ranks and the two kernels.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <numeric>
#include <random>
#include <thread>
#include <exception>

#include "common.hpp"

using namespace pbn;

#define KMI_MAX_DIM 16     // x, y and up to 14 conditioning variables
#define KMI_MAX_K 64
#define KMI_TILE 256

struct pbn_kmi {
    pbn::ctx_ptr ctx;
    int64_t N = 0;
    int n_vars = 0, k = 0, shuffle_neighbors = 5, samples = 1000;
    uint32_t seed = 0;
    std::vector<std::vector<float>> ranks;     // [n_vars][N] rank of every row in every column (rank_data, :17-52)
    std::vector<std::vector<double>> values;   // [n_vars][N] original values (neighbours in z for the conditional shuffle)
    dev_buf<float> d_ranks;                    // [n_vars][N]
    dev_buf<double> d_values;                  // [n_vars][N]
    dev_buf<float> d_x;                        // [N] permuted x ranks of the current sample
    dev_buf<int32_t> d_eps, d_cnt;             // [N], [3][N]
    dev_buf<int32_t> d_nbr;                    // [N][shuffle_neighbors]
    dev_buf<float> d_cand;                     // [slices][k + 1][N]
    dev_buf<float> d_sorted;                   // window form: [dims - 1][N] columns in window-axis rank order
    dev_buf<int32_t> d_inv;                    // window form: [N] row at every window-axis rank
    int inv_of = -1;                           // variable d_inv was built for (-1: none) - the window axis is never the permuted column
    std::vector<double> harmonic;              // digamma(n) = harmonic[n - 1] - gamma for integer n
    int64_t evaluations = 0;
};

namespace {

struct KmiArgs {
    const float* col[KMI_MAX_DIM];   // col[0] = x (possibly the permuted copy), col[1] = y, col[2..] = z
    int dims;
    int64_t n;
    int k;
    int32_t* eps;
    int32_t* cnt;   // [3][n]: n_xz, n_yz, n_z
    int slices;     // the other rows are cut into `slices` ranges (grid.y) so that small tables fill the device too
    float* cand;    // slices > 1: [slices][k + 1][n] the k + 1 smallest distances every slice found
};

// eps_i = Chebyshev distance from row i to its k-th neighbour (the row itself, at distance 0, is the 0-th): the (k+1)-th
// smallest of the N distances.  Every thread keeps its k+1 smallest distances in a sorted array; once it is full only a
// closer row costs an insertion.
// D: number of columns when it is one of the instantiated small values (coordinates in registers, loops unrolled), 0 = any
// (runtime a.dims); TILE: rows per block - small tables use 64 so that more compute units get a block.  The sorted
// candidate lists live in LDS ([slot][thread], conflict-free): a lane that inserts drags its whole wave through the
// insertion, so the list has to be cheap to touch (in scratch memory this kernel was 6x slower).
template <int D, int TILE>
__global__ __launch_bounds__(TILE) void kmi_eps_kernel(KmiArgs a) {
    constexpr int MAXD = D ? D : KMI_MAX_DIM;
    __shared__ float tile[MAXD][TILE];
    extern __shared__ float best[];   // [k + 1][TILE]
    const int dims = D ? D : a.dims;
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * TILE + tid;
    float mine[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) mine[d] = (d < dims && i < a.n) ? a.col[d][i] : 0.f;
    const int keep = a.k + 1;
    for (int q = 0; q < keep; ++q) best[q * TILE + tid] = INFINITY;
    float worst = INFINITY;   // best[keep - 1]
    auto offer = [&](float dist) {
        if (dist < worst) {
            int q = keep - 1;
            while (q > 0 && best[(q - 1) * TILE + tid] > dist) { best[q * TILE + tid] = best[(q - 1) * TILE + tid]; --q; }
            best[q * TILE + tid] = dist;
            worst = best[(keep - 1) * TILE + tid];
        }
    };
    auto distance = [&](int t) {
        float dist = 0.f;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < dims) dist = fmaxf(dist, fabsf(mine[d] - tile[d][t]));
        return dist;
    };
    const int64_t per = ((a.n + a.slices - 1) / a.slices + TILE - 1) / TILE * TILE;
    const int64_t jbeg = (int64_t)blockIdx.y * per, jend = jbeg + per < a.n ? jbeg + per : a.n;
    for (int64_t j0 = jbeg; j0 < jend; j0 += TILE) {
        const int64_t j = j0 + tid;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < dims) tile[d][tid] = j < jend ? a.col[d][j] : 0.f;
        __syncthreads();
        const int cnt = (int)((jend - j0 < TILE) ? jend - j0 : TILE);
        int t = 0;
        for (; t + 4 <= cnt; t += 4) {   // four distances in flight, then the (rarely taken) insertions
            const float d0 = distance(t), d1 = distance(t + 1), d2 = distance(t + 2), d3 = distance(t + 3);
            if (fminf(fminf(d0, d1), fminf(d2, d3)) < worst) { offer(d0); offer(d1); offer(d2); offer(d3); }
        }
        for (; t < cnt; ++t) offer(distance(t));
        __syncthreads();
    }
    if (i >= a.n) return;
    if (a.slices == 1) { a.eps[i] = (int32_t)worst; return; }
    for (int q = 0; q < keep; ++q) a.cand[((size_t)blockIdx.y * keep + q) * a.n + i] = best[q * TILE + tid];
}

// slices > 1: the (k+1)-th smallest of the slices' candidates
template <int TILE>
__global__ __launch_bounds__(TILE) void kmi_eps_merge_kernel(KmiArgs a) {
    extern __shared__ float best[];   // [k + 1][TILE]
    const int tid = threadIdx.x;
    const int64_t i = (int64_t)blockIdx.x * TILE + tid;
    if (i >= a.n) return;
    const int keep = a.k + 1;
    for (int q = 0; q < keep; ++q) best[q * TILE + tid] = INFINITY;
    float worst = INFINITY;
    for (int c = 0; c < a.slices * keep; ++c) {
        const float dist = a.cand[(size_t)c * a.n + i];
        if (dist < worst) {
            int q = keep - 1;
            while (q > 0 && best[(q - 1) * TILE + tid] > dist) { best[q * TILE + tid] = best[(q - 1) * TILE + tid]; --q; }
            best[q * TILE + tid] = dist;
            worst = best[(keep - 1) * TILE + tid];
        }
    }
    a.eps[i] = (int32_t)worst;
}

// counts of rows strictly inside eps_i: in z (Chebyshev over the conditioning columns), and of those the ones whose x /
// y rank is also strictly inside (kdtree.hpp:474-520; mutual_information.cpp:63-105 for one conditioning variable)
template <int D, int TILE>
__global__ __launch_bounds__(TILE) void kmi_count_kernel(KmiArgs a) {
    constexpr int MAXD = D ? D : KMI_MAX_DIM;
    __shared__ float tile[MAXD][TILE];
    const int dims = D ? D : a.dims;
    const int64_t i = (int64_t)blockIdx.x * TILE + threadIdx.x;
    float mine[MAXD];
#pragma unroll
    for (int d = 0; d < MAXD; ++d) mine[d] = (d < dims && i < a.n) ? a.col[d][i] : 0.f;
    const float eps = i < a.n ? (float)a.eps[i] : 0.f;
    int nxz = 0, nyz = 0, nz = 0;
    const int64_t per = ((a.n + a.slices - 1) / a.slices + TILE - 1) / TILE * TILE;
    const int64_t jbeg = (int64_t)blockIdx.y * per, jend = jbeg + per < a.n ? jbeg + per : a.n;
    for (int64_t j0 = jbeg; j0 < jend; j0 += TILE) {
        const int64_t j = j0 + threadIdx.x;
#pragma unroll
        for (int d = 0; d < MAXD; ++d)
            if (d < dims) tile[d][threadIdx.x] = j < jend ? a.col[d][j] : 0.f;
        __syncthreads();
        const int cnt = (int)((jend - j0 < TILE) ? jend - j0 : TILE);
#pragma unroll 4
        for (int t = 0; t < cnt; ++t) {   // branch-free: the compiler keeps several rows' LDS reads in flight
            float dz = 0.f;
#pragma unroll
            for (int d = 2; d < MAXD; ++d)
                if (d < dims) dz = fmaxf(dz, fabsf(mine[d] - tile[d][t]));
            const int in = dz < eps;
            nz += in;
            nxz += in & (int)(fabsf(mine[0] - tile[0][t]) < eps);
            nyz += in & (int)(fabsf(mine[1] - tile[1][t]) < eps);
        }
        __syncthreads();
    }
    if (i >= a.n) return;
    if (a.slices == 1) { a.cnt[i] = nxz; a.cnt[a.n + i] = nyz; a.cnt[2 * a.n + i] = nz; return; }
    atomicAdd(&a.cnt[i], nxz); atomicAdd(&a.cnt[a.n + i], nyz); atomicAdd(&a.cnt[2 * a.n + i], nz);   // integer sums: order-free
}

// ---- sorted-window form (round 6; tables of at least KMI_WINDOW_MIN_ROWS rows) -----------------------------------------------------------
// Every column holds the ranks 0 .. N-1 exactly once (rank_data: ordinal ranks), so along ONE axis w the rows at Chebyshev distance < e of
// row i are among the 2 e - 1 rows whose w-rank lies within e of i's - the reference's own trick for one conditioning variable
// (mutual_information.cpp:60-98 walks sort_z) and what its kd-tree does in general.  The columns are gathered once per evaluation into w-rank
// order (w = the first conditioning variable, or y without one: never the permuted x), a thread owns rank r and walks r +- 1, r +- 2, ... -
// coalesced, cache-resident reads of consecutive elements - keeping its k + 1 smallest distances as the all-pairs kernel does; a candidate
// at step s is at distance >= s, so the walk ends at s >= the (k+1)-th smallest distance found so far: eps_i.  The counts of the rows strictly
// inside eps_i in the subspaces are a second walk over s < eps_i.  O(N eps) pair evaluations instead of O(N^2) - eps ~ (k N^2 / 8)^(1/3) at
// three columns: 1e4 at 1e6 rows - with the same integer results.
#define KMI_WINDOW_MIN_ROWS 32768
struct KmiWinArgs {
    const float* sc[KMI_MAX_DIM];   // columns other than the window axis in w-rank order: sc[0] = x, sc[1] = y (with conditioning variables), then z1 ...
    const int32_t* inv;             // [n] row whose w-rank is r
    int others;                     // number of sc columns
    int has_z;                      // conditioning variables present: pass 2 counts n_xz, n_yz, n_z
    int64_t n;
    int k;
    int32_t* eps;
    int32_t* cnt;                   // [3][n] in ROW order
};
__global__ __launch_bounds__(256) void kmi_invert_kernel(const float* __restrict__ rank, int64_t n, int32_t* __restrict__ inv) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i < n) inv[(int64_t)rank[i]] = (int32_t)i;
}
__global__ __launch_bounds__(256) void kmi_gather_kernel(const float* __restrict__ col, const int32_t* __restrict__ inv, int64_t n, float* __restrict__ out) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) out[r] = col[inv[r]];
}
template <int O>   // O: number of sc columns when 1, 2 or 3 (registers, unrolled), 0 = any
__global__ __launch_bounds__(256) void kmi_window_kernel(KmiWinArgs a) {
    constexpr int MAXO = O ? O : KMI_MAX_DIM - 1;
    extern __shared__ float best[];   // [k + 1][256]
    const int others = O ? O : a.others;
    const int tid = threadIdx.x;
    const int64_t r = (int64_t)blockIdx.x * 256 + tid;
    const bool valid = r < a.n;
    float mine[MAXO];
#pragma unroll
    for (int d = 0; d < MAXO; ++d) mine[d] = (d < others && valid) ? a.sc[d][r] : 0.f;
    const int keep = a.k + 1;
    for (int q = 0; q < keep; ++q) best[q * 256 + tid] = INFINITY;
    best[tid] = 0.f;   // the row itself
    float worst = keep == 1 ? 0.f : INFINITY;
    auto offer = [&](float dist) {
        if (dist < worst) {
            int q = keep - 1;
            while (q > 0 && best[(q - 1) * 256 + tid] > dist) { best[q * 256 + tid] = best[(q - 1) * 256 + tid]; --q; }
            best[q * 256 + tid] = dist;
            worst = best[(keep - 1) * 256 + tid];
        }
    };
    // ---- walk 1: the (k+1)-th smallest distance.  Four steps (eight candidates) per termination test; waves run on until their last lane is
    // done (no barrier inside: waves are independent).  A candidate past the point where the walk could have stopped is a real row at its real
    // distance: offering it changes nothing
    const int n = (int)a.n, ri = (int)r;
    auto dist_at = [&](int j, int step, bool ok) {
        const int jc = j < 0 ? 0 : (j >= n ? n - 1 : j);
        float dist = (float)step;
#pragma unroll
        for (int d = 0; d < MAXO; ++d)
            if (d < others) dist = fmaxf(dist, fabsf(mine[d] - a.sc[d][jc]));
        return ok ? dist : INFINITY;
    };
    for (int s0 = 1; s0 < n; s0 += 4) {
        const bool go = valid && (float)s0 < worst;
        if (!__any(go)) break;
        if (go) {
            float dd[8];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int st = s0 + u;
                dd[2 * u] = dist_at(ri + st, st, ri + st < n);
                dd[2 * u + 1] = dist_at(ri - st, st, ri - st >= 0);
            }
            const float lo = fminf(fminf(fminf(dd[0], dd[1]), fminf(dd[2], dd[3])), fminf(fminf(dd[4], dd[5]), fminf(dd[6], dd[7])));
            if (lo < worst) {
#pragma unroll
                for (int u = 0; u < 8; ++u) offer(dd[u]);
            }
        }
    }
    if (!valid) return;
    const int64_t row = a.inv[r];
    a.eps[row] = (int32_t)worst;
    if constexpr (O == 1) return;   // (no conditioning variable: the marginal counts have a closed form on ranks - the host takes them)
    if (!a.has_z) return;
    // ---- walk 2: rows strictly inside eps in z (the window axis and sc[2...]), and of those the ones also inside in x / in y ----
    const float e = (float)(int32_t)worst;
    int nxz = 1, nyz = 1, nz = 1;   // the row itself
    const int last = (int)e - 1;
    for (int s0 = 1; s0 <= last; s0 += 2) {
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int st = s0 + (u >> 1);
            const int j = (u & 1) ? ri - st : ri + st;
            const bool ok = st <= last && j >= 0 && j < n;
            const int jc = j < 0 ? 0 : (j >= n ? n - 1 : j);
            float dz = 0.f;
#pragma unroll
            for (int d = 2; d < MAXO; ++d)
                if (d < others) dz = fmaxf(dz, fabsf(mine[d] - a.sc[d][jc]));
            const int in = ok & (int)(dz < e);
            nz += in;
            nxz += in & (int)(fabsf(mine[0] - a.sc[0][jc]) < e);
            nyz += in & (int)(fabsf(mine[1] - a.sc[1][jc]) < e);
        }
    }
    a.cnt[row] = nxz; a.cnt[a.n + row] = nyz; a.cnt[2 * a.n + row] = nz;
}

// the `m` nearest rows of every row in the space of the ORIGINAL conditioning values (Chebyshev), nearest first, the row
// itself included (kdtree query of shuffled_pvalue, mutual_information.hpp:178-188)
struct NbrArgs {
    const double* col[KMI_MAX_DIM];
    int dims;
    int64_t n;
    int m;
    int32_t* out;   // [n][m]
};
__global__ __launch_bounds__(KMI_TILE) void kmi_neighbors_kernel(NbrArgs a) {
    __shared__ double tile[KMI_MAX_DIM][KMI_TILE];
    const int64_t i = (int64_t)blockIdx.x * KMI_TILE + threadIdx.x;
    double mine[KMI_MAX_DIM];
    for (int d = 0; d < a.dims; ++d) mine[d] = i < a.n ? a.col[d][i] : 0.0;
    double best[KMI_MAX_K];
    int32_t who[KMI_MAX_K];
    for (int q = 0; q < a.m; ++q) { best[q] = INFINITY; who[q] = -1; }
    double worst = INFINITY;
    for (int64_t j0 = 0; j0 < a.n; j0 += KMI_TILE) {
        const int64_t j = j0 + threadIdx.x;
        for (int d = 0; d < a.dims; ++d) tile[d][threadIdx.x] = j < a.n ? a.col[d][j] : 0.0;
        __syncthreads();
        const int cnt = (int)((a.n - j0 < KMI_TILE) ? a.n - j0 : KMI_TILE);
        for (int t = 0; t < cnt; ++t) {
            double dist = 0.0;
            for (int d = 0; d < a.dims; ++d) dist = fmax(dist, fabs(mine[d] - tile[d][t]));
            if (dist < worst) {
                int q = a.m - 1;
                while (q > 0 && best[q - 1] > dist) { best[q] = best[q - 1]; who[q] = who[q - 1]; --q; }
                best[q] = dist; who[q] = (int32_t)(j0 + t);
                worst = best[a.m - 1];
            }
        }
        __syncthreads();
    }
    if (i < a.n)
        for (int q = 0; q < a.m; ++q) a.out[i * a.m + q] = who[q];
}

struct IndexLess {   // kdtree::IndexComparator
    const float* v;
    bool operator()(size_t a, size_t b) const { return v[a] < v[b]; }
};

struct Kmi {
    pbn_kmi* h;

    double digamma_int(int64_t n) const {   // psi(n) = H_{n-1} - gamma
        static const double EULER = 0.57721566490153286060651209008240243;
        if (n < 1) return -std::numeric_limits<double>::infinity();
        return h->harmonic[(size_t)n - 1] - EULER;
    }

    template <bool EPS, int D, int TILE>
    void launch_dt(const KmiArgs& a) {
        const dim3 grid((unsigned)ceil_div(a.n, TILE), (unsigned)a.slices), block(TILE);
        const size_t lds = (size_t)(a.k + 1) * TILE * sizeof(float);
        if (EPS) {
            hipLaunchKernelGGL((kmi_eps_kernel<D, TILE>), grid, block, lds, h->ctx->stream, a);
            if (a.slices > 1) hipLaunchKernelGGL((kmi_eps_merge_kernel<TILE>), dim3(grid.x), block, lds, h->ctx->stream, a);
        } else {
            if (a.slices > 1) HIP_CHECK(hipMemsetAsync(a.cnt, 0, (size_t)3 * a.n * sizeof(int32_t), h->ctx->stream));
            hipLaunchKernelGGL((kmi_count_kernel<D, TILE>), grid, block, 0, h->ctx->stream, a);
        }
    }
    template <bool EPS, int D>
    void launch_d(const KmiArgs& a) {
        if (a.n < 64 * 1024) launch_dt<EPS, D, 64>(a); else launch_dt<EPS, D, 256>(a);
    }
    template <bool EPS>
    void launch(const KmiArgs& a) {
        switch (a.dims) {
            case 2: launch_d<EPS, 2>(a); break;
            case 3: launch_d<EPS, 3>(a); break;
            case 4: launch_d<EPS, 4>(a); break;
            case 5: launch_d<EPS, 5>(a); break;
            case 6: launch_d<EPS, 6>(a); break;
            default: launch_d<EPS, 0>(a); break;
        }
    }

    // MI of columns vars = [x, y, z...]; a permuted sample passes its x ranks (device copy + the host original)
    double evaluate(const std::vector<int>& vars, const float* x_override, const float* x_host = nullptr) {
        pbn_ctx* ctx = h->ctx;
        const int64_t N = h->N;
        KmiArgs a{};
        a.dims = (int)vars.size(); a.n = N; a.k = h->k;
        for (int d = 0; d < a.dims; ++d) a.col[d] = h->d_ranks.p + (size_t)vars[d] * N;
        if (x_override) a.col[0] = x_override;
        a.eps = h->d_eps.p; a.cnt = h->d_cnt.p;
        const bool window = N >= knob_int("PBN_KMI_WINDOW_MIN_ROWS", KMI_WINDOW_MIN_ROWS);
        if (window) {
            // the window axis: the first conditioning variable, y without one (x is the column the permutation samples replace)
            const int wv = a.dims == 2 ? 1 : 2;
            const int32_t* inv_before = h->d_inv.p;
            h->d_inv.reserve((size_t)N);
            if (h->d_inv.p != inv_before) h->inv_of = -1;
            h->d_sorted.reserve((size_t)(a.dims - 1) * N);
            const dim3 grid((unsigned)ceil_div(N, 256)), block(256);
            if (h->inv_of != vars[wv]) {
                hipLaunchKernelGGL(kmi_invert_kernel, grid, block, 0, ctx->stream, a.col[wv], N, h->d_inv.p);
                h->inv_of = vars[wv];
            }
            KmiWinArgs w{};
            w.inv = h->d_inv.p; w.n = N; w.k = h->k; w.eps = a.eps; w.cnt = a.cnt; w.has_z = a.dims > 2 ? 1 : 0;
            int o = 0;
            for (int d = 0; d < a.dims; ++d) {
                if (d == wv) continue;
                float* dst = h->d_sorted.p + (size_t)o * N;
                hipLaunchKernelGGL(kmi_gather_kernel, grid, block, 0, ctx->stream, a.col[d], h->d_inv.p, N, dst);
                w.sc[o++] = dst;
            }
            w.others = o;
            const size_t lds = (size_t)(h->k + 1) * 256 * sizeof(float);
            switch (o) {
                case 1: hipLaunchKernelGGL(kmi_window_kernel<1>, grid, block, lds, ctx->stream, w); break;
                case 2: hipLaunchKernelGGL(kmi_window_kernel<2>, grid, block, lds, ctx->stream, w); break;
                case 3: hipLaunchKernelGGL(kmi_window_kernel<3>, grid, block, lds, ctx->stream, w); break;
                default: hipLaunchKernelGGL(kmi_window_kernel<0>, grid, block, lds, ctx->stream, w); break;
            }
            HIP_CHECK(hipGetLastError());
        } else {
        {   // enough blocks for ~4 per compute unit
            const int64_t bx = ceil_div(N, N < 64 * 1024 ? 64 : 256);
            a.slices = (int)std::max<int64_t>(1, std::min<int64_t>(16, (int64_t)h->ctx->num_cus * 4 / bx));
            if (a.slices > 1) { h->d_cand.reserve((size_t)a.slices * (h->k + 1) * N); a.cand = h->d_cand.p; }
        }
            launch<true>(a);
        }
        ++h->evaluations;
        std::vector<int32_t> eps((size_t)N), cnt;
        double res = 0;
        if (a.dims == 2) {   // mi_pair (mutual_information.cpp:9-43): marginal counts have a closed form on ranks
            HIP_CHECK(hipMemcpyAsync(eps.data(), h->d_eps.p, (size_t)N * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            const float* x = x_override ? x_host : h->ranks[vars[0]].data();
            const float* y = h->ranks[vars[1]].data();
            const int rows = (int)N;
            for (int i = 0; i < rows; ++i) {
                const int e = eps[i], v1 = (int)x[i], v2 = (int)y[i];
                const int nv1 = std::min(1 + v1, e) + std::min(rows - v1, e) - 1;
                const int nv2 = std::min(1 + v2, e) + std::min(rows - v2, e) - 1;
                res -= digamma_int(nv1) + digamma_int(nv2);
            }
            res /= (double)N;
            res += digamma_int(h->k) + digamma_int(N);
            return res;
        }
        if (!window) launch<false>(a);
        HIP_CHECK(hipGetLastError());
        cnt.resize((size_t)3 * N);
        HIP_CHECK(hipMemcpyAsync(cnt.data(), h->d_cnt.p, cnt.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        for (int64_t i = 0; i < N; ++i)   // mi_triple / mi_general (mutual_information.cpp:107-114,136-143)
            res += digamma_int(cnt[2 * N + i]) - digamma_int(cnt[i]) - digamma_int(cnt[N + i]);
        res /= (double)N;
        res += digamma_int(h->k);
        return res;
    }

    void check(const std::vector<int>& vars) const {
        if ((int)vars.size() > KMI_MAX_DIM) throw invalid_error("KMutualInformation: conditioning set too large");
        for (size_t a = 0; a < vars.size(); ++a) {
            if (vars[a] < 0 || vars[a] >= h->n_vars) throw invalid_error("KMutualInformation: variable index out of range");
            for (size_t b = 0; b < a; ++b)
                if (vars[a] == vars[b]) throw invalid_error("KMutualInformation: repeated variable");
        }
    }

    // KMutualInformation::pvalue (mutual_information.cpp:157-190, hpp:128-206)
    double pvalue(const std::vector<int>& vars) {
        check(vars);
        pbn_ctx* ctx = h->ctx;
        const int64_t N = h->N;
        const double original = evaluate(vars, nullptr);
        std::mt19937 rng{h->seed};
        std::vector<float> shuffled(h->ranks[vars[0]]);
        int count_greater = 0;
        if (vars.size() == 2) {
            for (int s = 0; s < h->samples; ++s) {
                std::shuffle(shuffled.begin(), shuffled.end(), rng);
                HIP_CHECK(hipMemcpyAsync(h->d_x.p, shuffled.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
                if (evaluate(vars, h->d_x.p, shuffled.data()) >= original) ++count_greater;
            }
            return (double)count_greater / h->samples;
        }
        // neighbours in the space of the original conditioning values
        const int m = h->shuffle_neighbors;
        NbrArgs na{};
        na.dims = (int)vars.size() - 2; na.n = N; na.m = m; na.out = h->d_nbr.p;
        for (int d = 0; d < na.dims; ++d) na.col[d] = h->d_values.p + (size_t)vars[d + 2] * N;
        hipLaunchKernelGGL(kmi_neighbors_kernel, dim3((unsigned)ceil_div(N, KMI_TILE)), dim3(KMI_TILE), 0, ctx->stream, na);
        HIP_CHECK(hipGetLastError());
        std::vector<int32_t> neighbors((size_t)N * m);   // column i = the m neighbours of row i (MatrixXi(m, N), column-major)
        HIP_CHECK(hipMemcpyAsync(neighbors.data(), h->d_nbr.p, neighbors.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        const float* original_x = h->ranks[vars[0]].data();
        std::vector<size_t> order((size_t)N), sorted_indices((size_t)N);
        std::iota(order.begin(), order.end(), 0);
        std::vector<bool> used((size_t)N);
        // Round 6: the RNG-bound part of a sample (the three loops below, serial in the mt19937 stream and kept call for call) and the rest of
        // it (std::sort of the permuted values into ranks, upload, the device evaluation, the digamma sum) run as a two-stage pipeline: this
        // thread draws sample s + 1 while a worker finishes sample s.  Same values, same comparisons - only `count_greater` is taken in the worker.
        std::vector<float> stage[2] = {std::vector<float>((size_t)N), std::vector<float>((size_t)N)};
        std::thread worker;
        std::exception_ptr failed;
        auto finish = [&](int b) {
            try {
                HIP_CHECK(hipSetDevice(ctx->device));
                std::vector<float>& v = stage[b];
                std::iota(sorted_indices.begin(), sorted_indices.end(), 0);
                std::sort(sorted_indices.begin(), sorted_indices.end(), IndexLess{v.data()});
                for (size_t i = 0; i < sorted_indices.size(); ++i) v[sorted_indices[i]] = (float)i;
                HIP_CHECK(hipMemcpyAsync(h->d_x.p, v.data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
                if (evaluate(vars, h->d_x.p, v.data()) >= original) ++count_greater;
            } catch (...) { failed = std::current_exception(); }
        };
        const bool pipelined = N >= 65536 && knob_int("PBN_KMI_PIPELINE", 1) != 0;
        for (int s = 0; s < h->samples; ++s) {
            std::vector<float>& shuffled = stage[s & 1];   // (the worker reads the other one)
            std::shuffle(order.begin(), order.end(), rng);
            // shuffle_dataframe (mutual_information.hpp:128-167)
            for (int64_t i = 0; i < N; ++i) std::shuffle(neighbors.begin() + i * m, neighbors.begin() + (i + 1) * m, rng);
            std::uniform_real_distribution<float> tiebreaker(-0.5, 0.5);
            for (int64_t i = 0; i < N; ++i) {
                const size_t index = order[(size_t)i];
                if (i + 24 < N) {   // the rows are visited in shuffled order: fetch the lines of a later row now (no effect on what is computed)
                    const size_t ahead = order[(size_t)i + 24];
                    __builtin_prefetch(&neighbors[ahead * m]);
                    __builtin_prefetch(&shuffled[ahead], 1);
                }
                int neighbor_index = 0;
                for (int j = 0; j < m; ++j) {
                    neighbor_index = neighbors[index * m + j];
                    if (!used[(size_t)neighbor_index]) break;
                }
                if (used[(size_t)neighbor_index]) {
                    shuffled[index] = original_x[neighbor_index] + tiebreaker(rng);
                } else {
                    shuffled[index] = original_x[neighbor_index];
                    used[(size_t)neighbor_index] = true;
                }
            }
            std::fill(used.begin(), used.end(), false);
            if (worker.joinable()) worker.join();
            if (failed) std::rethrow_exception(failed);
            if (pipelined) worker = std::thread(finish, s & 1);
            else finish(s & 1);
        }
        if (worker.joinable()) worker.join();
        if (failed) std::rethrow_exception(failed);
        return (double)count_greater / h->samples;
    }
};

}  // namespace

extern "C" {

// cols: n_vars host columns of N values (already converted to double; a float32 table converts exactly).  The ranks are
// taken as rank_data does (mutual_information.hpp:17-52): ONE index vector, std::sort'ed column after column.
int pbn_kmi_create(pbn_ctx* ctx, const double* const* cols, int n_vars, int64_t N, int k, uint32_t seed, int shuffle_neighbors,
                   int samples, pbn_kmi** out) {
    return guarded(mu_of(ctx), [&] {
        if (!ctx || !cols || !out) throw invalid_error("pbn_kmi_create: null argument");
        if (n_vars < 2) throw invalid_error("DataFrame does not contain enough continuous columns.");
        if (k < 1 || k > KMI_MAX_K || k >= N) throw invalid_error("KMutualInformation: k must be between 1 and min(64, rows - 1)");
        if (shuffle_neighbors < 1 || shuffle_neighbors > KMI_MAX_K || shuffle_neighbors > N)
            throw invalid_error("KMutualInformation: shuffle_neighbors must be between 1 and min(64, rows)");
        if (samples < 1) throw invalid_error("KMutualInformation: samples must be positive");
        if (N >= (1 << 24)) throw invalid_error("KMutualInformation: ranks are kept in float32 like the reference's: at most 2^24 rows");
        HIP_CHECK(hipSetDevice(ctx->device));
        auto h = std::make_unique<pbn_kmi>();
        h->ctx = ctx; h->N = N; h->n_vars = n_vars; h->k = k; h->seed = seed; h->shuffle_neighbors = shuffle_neighbors; h->samples = samples;
        h->ranks.assign(n_vars, std::vector<float>((size_t)N));
        h->values.assign(n_vars, std::vector<double>((size_t)N));
        std::vector<size_t> indices((size_t)N);
        std::iota(indices.begin(), indices.end(), 0);
        for (int j = 0; j < n_vars; ++j) {
            std::copy(cols[j], cols[j] + N, h->values[j].begin());
            const double* v = cols[j];
            std::sort(indices.begin(), indices.end(), [v](size_t a, size_t b) { return v[a] < v[b]; });
            for (int64_t i = 0; i < N; ++i) h->ranks[j][indices[(size_t)i]] = (float)i;
        }
        h->d_ranks.alloc((size_t)n_vars * N); h->d_values.alloc((size_t)n_vars * N);
        h->d_x.alloc((size_t)N); h->d_eps.alloc((size_t)N); h->d_cnt.alloc((size_t)3 * N); h->d_nbr.alloc((size_t)N * shuffle_neighbors);
        for (int j = 0; j < n_vars; ++j) {
            HIP_CHECK(hipMemcpyAsync(h->d_ranks.p + (size_t)j * N, h->ranks[j].data(), (size_t)N * sizeof(float), hipMemcpyHostToDevice, ctx->stream));
            HIP_CHECK(hipMemcpyAsync(h->d_values.p + (size_t)j * N, h->values[j].data(), (size_t)N * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
        h->harmonic.assign((size_t)N + 1, 0.0);
        for (int64_t n = 1; n <= N; ++n) h->harmonic[(size_t)n] = h->harmonic[(size_t)n - 1] + 1.0 / (double)n;
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = h.release();
    });
}

void pbn_kmi_destroy(pbn_kmi* h) {
    if (!h) return;
    pbn::ctx_pin pin_(h->ctx);
    std::lock_guard<std::recursive_mutex> lock_(mu_of(h));
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    delete h;
}

// KMutualInformation::mi (mutual_information.cpp:142-155)
int pbn_kmi_value(pbn_kmi* h, int v1, int v2, int n_cond, const int* cond, double* mi) {
    return guarded(mu_of(h), [&] {
        if (!h || !mi || (n_cond > 0 && !cond)) throw invalid_error("pbn_kmi_value: null argument");
        std::vector<int> vars{v1, v2};
        vars.insert(vars.end(), cond, cond + n_cond);
        Kmi e{h};
        e.check(vars);
        *mi = e.evaluate(vars, nullptr);
    });
}

// pbn_ci_pvalue_fn over a pbn_kmi handle: the permutation p-value with the handle's seed and number of samples
double pbn_kmi_pvalue(void* user, int v1, int v2, int n_cond, const int* cond) {
    pbn_kmi* h = (pbn_kmi*)user;
    double result = std::nan("");
    (void)guarded([&] {
        if (!h || (n_cond > 0 && !cond)) throw invalid_error("pbn_kmi_pvalue: null argument");
        std::vector<int> vars{v1, v2};
        vars.insert(vars.end(), cond, cond + n_cond);
        Kmi e{h};
        result = e.pvalue(vars);
    });
    return result;
}

}  // extern "C"
