// Hybrid mutual-information independence test (learning/independences/hybrid/mutual_information.{hpp,cpp}): the CI test
// of the MMHC restriction phase on mixed discrete / continuous tables (BASELINE config 5).
//
// Under the conditional-Gaussian assumption every MI(X; Y | Z) of the reference is a combination of (a) counts of the
// discrete configurations and (b) log-determinants of covariances of the continuous variables inside configurations
// and inside configurations pooled over X and / or Y.  The reference makes 2 passes over the rows per quantity
// (conditional_means_impl, conditional_covariance_impl).  Here one device pass per test gathers, per configuration of
// the discrete variables involved, the count and the pilot-shifted first and second moments of the continuous ones;
// moments are additive, so every pooled covariance is a sum of per-configuration moments on the host.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <memory>
#include <thread>
#include <exception>

#include <rocprim/device/device_radix_sort.hpp>

#include "common.hpp"
#include "hostmath.hpp"
#include "stats_kernels.hpp"

using namespace pbn;

#define MI_MAX_CONT 24
#define MI_MAX_DISC 16
#define MI_SORTED_MAX_CONT 16     // continuous variables the register-accumulator kernel is instantiated for (152 accumulators)
#define MI_SORTED_ROWS 4096       // rows of one configuration a 256-thread block sums
#define MI_GROUP_CACHE 1024       // cached row groupings (one per set of discrete variables), least recently used out

// Rows grouped by the configuration of one set of discrete variables (sorted ids, first id fastest): `perm` lists the
// rows configuration by configuration, ascending inside a configuration (stable radix sort), `off` are the segment
// bounds = the counts, `blk` cuts every segment into pieces of at most MI_SORTED_ROWS rows.  Built once per set and
// reused by every test over it; the set of no variables is the identity (perm empty).
struct DiscGroup {
    std::vector<int> vars;
    int G = 1;
    dev_buf<int32_t> perm;          // [N]
    std::vector<int64_t> off;       // [G + 1]
    dev_buf<int32_t> blk;           // [nblk][4]: configuration, first position, end position, unused
    std::vector<int> blk_off;       // [G + 1] first block of every configuration
    int nblk = 0;
    uint64_t stamp = 0;
    // configuration id in this grouping's order -> id in a test's own (x, y, z...) order, per variable order seen
    std::map<std::vector<int>, std::vector<int>> order_maps;
    // per configuration, the pilot-shifted sums and products of ALL continuous columns (Engine::ensure_full): every test over
    // this set of discrete variables reads its moments out of them
    bool full_ready = false;
    std::vector<double> fullS, fullP;   // [G][nc], [G][nc][nc]
};

struct pbn_mi {
    pbn::ctx_ptr ctx;
    const pbn_table* table = nullptr;  // continuous columns (borrowed), null when there are none
    int n_cont = 0, n_disc = 0;
    int64_t N = 0;
    bool asymptotic = true;
    std::vector<int> card;
    std::vector<char> disc_null, cont_null;   // columns holding nulls: code == card[j] / NaN; such rows drop out of a test
    bool any_null = false;
    dev_buf<int32_t> codes_dev;               // [n_disc][N]
    std::vector<double> shift;                // pilot mean of every continuous column
    int64_t device_passes = 0, host_passes = 0, device_launches = 0;
    std::vector<int> order;  // external index -> variable id for the callback form (empty = identity)
    std::map<std::vector<int>, std::unique_ptr<DiscGroup>> groups;
    uint64_t clock = 0;
    int64_t groups_built = 0, count_only = 0;
    dev_buf<int32_t> iota;       // [N] 0..N-1, the values the radix sort permutes
    dev_buf<uint32_t> keys[2];   // [N] configuration ids, unsorted / sorted
    dev_buf<int32_t> first;      // [G] first sorted position of every configuration
    dev_buf<char> sort_tmp;
    dev_buf<double> shift_dev;   // the pilot means on the device, indexed by table column (Engine::ensure_full)
    dev_buf<char> rowmajor;      // row-major mirror of the continuous columns for the gathered Gram of the groupings (Engine::ensure_full)
    bool rowmajor_tried = false;
    int64_t full_grams = 0;      // groupings whose full per-configuration moments were taken
    size_t full_bytes_held = 0;  // host bytes of the cached full moments
    // PBN_MI_TIMING=1: wall seconds per phase, printed when the handle is destroyed
    double t_group = 0, t_device = 0, t_host = 0, t_prep = 0;
    int64_t batches = 0;
};

static inline double mi_now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

namespace {

struct GroupArgs {
    const void* base;
    int64_t ld;
    int cols[MI_MAX_CONT];
    double shift[MI_MAX_CONT];
    int c;
    const int32_t* codes;
    int64_t codes_ld;
    int dvar[MI_MAX_DISC];
    int dstride[MI_MAX_DISC];
    int m;
    int64_t n;
    int G, stats;      // configurations handled by this launch row (a window of the test's configurations) and statistics each
    int g0;            // first configuration of the window
    double* partial;   // [nblocks][G * stats]
    double* out;       // [G * stats]
    int64_t chunks_per_block;
};

__device__ __forceinline__ double wave_sum(double v) {
    v += __shfl_xor(v, 1); v += __shfl_xor(v, 2); v += __shfl_xor(v, 4);
    v += __shfl_xor(v, 8); v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
    return v;
}

// One wave per block.  A chunk of 64 rows is staged in LDS (shifted values, a trailing 1, the configuration id); then
// lane s owns statistic s of every configuration - statistic s is the product of two staged entries (i_s, j_s), the
// trailing 1 turning products into sums and the count - and walks the 64 rows in order adding into its own LDS
// cells: no two lanes share a cell, no atomics, fixed order, run-to-run identical, cost independent of how many
// configurations a chunk touches.  stats per configuration: count, c sums, c(c+1)/2 products (upper triangle).
template <typename T>
__global__ __launch_bounds__(64) void group_moments_kernel(const GroupArgs* __restrict__ descs) {
    extern __shared__ double lds[];
    const GroupArgs& a = descs[blockIdx.y];   // one independence test per grid row
    const int lane = threadIdx.x;
    const int total = a.G * a.stats, c1 = a.c + 1;
    double* acc = lds;                         // [G][stats]
    double* stage = lds + total;               // [64][c + 1]
    int* gid = (int*)(stage + 64 * c1);        // [64]
    unsigned char* pi = (unsigned char*)(gid + 64);   // [stats] first factor
    unsigned char* pj = pi + a.stats;                 // [stats] second factor
    for (int i = lane; i < total; i += 64) acc[i] = 0.0;
    for (int s0 = lane; s0 < a.stats; s0 += 64) {
        int fi, fj;
        if (s0 == 0) { fi = a.c; fj = a.c; }
        else if (s0 <= a.c) { fi = s0 - 1; fj = a.c; }
        else {
            int rem = s0 - 1 - a.c, i = 0;
            while (rem >= a.c - i) { rem -= a.c - i; ++i; }
            fi = i; fj = i + rem;
        }
        pi[s0] = (unsigned char)fi; pj[s0] = (unsigned char)fj;
    }
    __syncthreads();
    const int64_t chunk0 = (int64_t)blockIdx.x * a.chunks_per_block;
    for (int64_t ch = chunk0; ch < chunk0 + a.chunks_per_block; ++ch) {
        if (ch * 64 >= a.n) break;
        const int64_t r = ch * 64 + lane;
        const int rows = (int)((a.n - ch * 64 < 64) ? a.n - ch * 64 : 64);
        if (r < a.n) {
            for (int i = 0; i < a.c; ++i) stage[lane * c1 + i] = (double)((const T*)a.base)[(int64_t)a.cols[i] * a.ld + r] - a.shift[i];
            stage[lane * c1 + a.c] = 1.0;
            int g = 0;
            for (int j = 0; j < a.m; ++j) g += a.codes[(int64_t)a.dvar[j] * a.codes_ld + r] * a.dstride[j];
            g -= a.g0;
            gid[lane] = (g >= 0 && g < a.G) ? g : -1;   // rows of other windows are skipped
        }
        __syncthreads();
        for (int s0 = lane; s0 < a.stats; s0 += 64) {
            const int fi = pi[s0], fj = pj[s0];
            for (int rr = 0; rr < rows; ++rr) {
                const int g = gid[rr];
                if (g < 0) continue;
                double* cell = acc + (size_t)g * a.stats + s0;
                *cell += stage[rr * c1 + fi] * stage[rr * c1 + fj];
            }
        }
        __syncthreads();
    }
    double* out = a.partial + (size_t)blockIdx.x * total;
    for (int i = lane; i < total; i += 64) out[i] = acc[i];
}

// one wave per statistic: lane l adds the partials of blocks l, l + 64, ... in order, then a fixed butterfly
__global__ __launch_bounds__(64) void group_reduce_kernel(const GroupArgs* __restrict__ descs, int nblocks) {
    const GroupArgs& a = descs[blockIdx.y];
    const int total = a.G * a.stats;
    const int i = blockIdx.x;
    if (i >= total) return;
    double s = 0.0;
    for (int b = threadIdx.x; b < nblocks; b += 64) s += a.partial[(size_t)b * total + i];
    s = wave_sum(s);
    if (threadIdx.x == 0) a.out[i] = s;
}


// ---- sorted-segment path ------------------------------------------------------------------------------------------
// With the rows grouped by configuration (DiscGroup) a block only ever sees ONE configuration: every lane adds its rows
// into registers (no LDS cells, no atomics, all 256 lanes busy), a fixed butterfly + wave order gives the block's
// partial, and a second kernel adds a configuration's partials in block order.  Deterministic, and the work per test no
// longer depends on the number of configurations.
struct SortedArgs {
    const void* base;
    int64_t ld;
    int cols[MI_SORTED_MAX_CONT];
    double shift[MI_SORTED_MAX_CONT];
    const int32_t* perm;     // null: identity
    const int32_t* blk;      // [nblk][4]
    const int32_t* blk_off;  // device copy, [G + 1]
    int nblk, G;
    double* partial;         // [nblk][S]
    double* out;             // [G][S]
};

struct KeyArgs {
    int m;
    int dvar[MI_MAX_DISC], stride[MI_MAX_DISC];
};
__global__ __launch_bounds__(256) void config_keys_kernel(const int32_t* __restrict__ codes, int64_t n, KeyArgs k, uint32_t* __restrict__ keys) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    uint32_t g = 0;
    for (int j = 0; j < k.m; ++j) g += (uint32_t)codes[(int64_t)k.dvar[j] * n + r] * (uint32_t)k.stride[j];
    keys[r] = g;
}
__global__ __launch_bounds__(256) void iota_kernel(int32_t* v, int64_t n) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r < n) v[r] = (int32_t)r;
}
__global__ __launch_bounds__(256) void segment_first_kernel(const uint32_t* __restrict__ keys, int64_t n, int32_t* __restrict__ first) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (r >= n) return;
    const uint32_t k = keys[r];
    if (r == 0 || keys[r - 1] != k) first[k] = (int32_t)r;
}

// NULLS: some of the test's continuous columns hold NaN for null cells; a row with a NaN in any of the C columns drops
// out (the reference filters the rows valid in all columns of the test, mutual_information.cpp:152-215) and one more
// statistic, the number of rows kept, follows the products.
template <typename T, int C, bool NULLS>
__global__ __launch_bounds__(256) void moments_sorted_kernel(const SortedArgs* __restrict__ descs) {
    constexpr int S = C + C * (C + 1) / 2 + (NULLS ? 1 : 0);   // sums, upper-triangle products (the count is the segment length)
    __shared__ double red[4][S];
    const SortedArgs& a = descs[blockIdx.y];
    if ((int)blockIdx.x >= a.nblk) return;
    const int r0 = a.blk[4 * blockIdx.x + 1], r1 = a.blk[4 * blockIdx.x + 2];
    double acc[S];
#pragma unroll
    for (int i = 0; i < S; ++i) acc[i] = 0.0;
    const T* col[C];
#pragma unroll
    for (int i = 0; i < C; ++i) col[i] = (const T*)a.base + (int64_t)a.cols[i] * a.ld;
    for (int r = r0 + (int)threadIdx.x; r < r1; r += 256) {
        const int64_t row = a.perm ? a.perm[r] : r;
        double x[C];
        bool ok = true;
#pragma unroll
        for (int i = 0; i < C; ++i) { x[i] = (double)col[i][row] - a.shift[i]; ok = ok && (x[i] == x[i]); }
        if (NULLS) {
            if (!ok) continue;
            acc[S - 1] += 1.0;
        }
        int pos = C;
#pragma unroll
        for (int i = 0; i < C; ++i) {
            acc[i] += x[i];
#pragma unroll
            for (int j = i; j < C; ++j) { acc[pos] = __builtin_fma(x[i], x[j], acc[pos]); ++pos; }
        }
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < S; ++i) {
        const double v = wave_sum(acc[i]);
        if (lane == 0) red[wave][i] = v;
    }
    __syncthreads();
    if (threadIdx.x < S) a.partial[(size_t)blockIdx.x * S + threadIdx.x] = ((red[0][threadIdx.x] + red[1][threadIdx.x]) + red[2][threadIdx.x]) + red[3][threadIdx.x];
}

// one wave per (configuration, statistic): lanes stride over the configuration's blocks, fixed butterfly
__global__ __launch_bounds__(64) void sorted_reduce_kernel(const SortedArgs* __restrict__ descs, int S) {
    const SortedArgs& a = descs[blockIdx.y];
    const int cell = blockIdx.x;
    if (cell >= a.G * S) return;
    const int g = cell / S, st = cell - g * S;
    const int b0 = a.blk_off[g], b1 = a.blk_off[g + 1];
    double v = 0.0;
    for (int b = b0 + (int)threadIdx.x; b < b1; b += 64) v += a.partial[(size_t)b * S + st];
    v = wave_sum(v);
    if (threadIdx.x == 0) a.out[cell] = v;
}

// ---- regularised upper incomplete gamma Q(a, x): chi-square survival function (boost chi_squared complement) --------
double gamma_q(double a, double x) {
    if (std::isnan(x) || std::isnan(a)) return std::numeric_limits<double>::quiet_NaN();
    if (x <= 0) return 1.0;
    if (std::isinf(x)) return 0.0;
    const double lg = std::lgamma(a);
    if (x < a + 1.0) {  // series for P, Q = 1 - P
        double ap = a, sum = 1.0 / a, del = sum;
        for (int n = 0; n < 100000; ++n) {
            ap += 1.0;
            del *= x / ap;
            sum += del;
            if (std::fabs(del) < std::fabs(sum) * 1e-17) break;
        }
        return 1.0 - sum * std::exp(-x + a * std::log(x) - lg);
    }
    // Lentz continued fraction for Q
    const double tiny = 1e-300;
    double b = x + 1.0 - a, c = 1.0 / tiny, d = 1.0 / b, h = d;
    for (int i = 1; i < 100000; ++i) {
        const double an = -i * (i - a);
        b += 2.0;
        d = an * d + b; if (std::fabs(d) < tiny) d = tiny;
        c = b + an / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < 1e-16) break;
    }
    const double q = std::exp(-x + a * std::log(x) - lg) * h;
    return q < std::numeric_limits<double>::min() ? 0.0 : q;
}

const double PI_ = 3.14159265358979323846264338327950288;

double entropy_mvn(int d, double det) {  // mutual_information.cpp:921-924
    return 0.5 * d + 0.5 * d * std::log(2 * PI_) + 0.5 * std::log(det);
}

// pooled moments of a set of configurations over the continuous variables
struct Mom {
    double n = 0;
    int c = 0;
    std::vector<double> S, P;  // sums, products (full c x c)
    explicit Mom(int c_ = 0) : c(c_), S(c_, 0.0), P((size_t)c_ * c_, 0.0) {}
    void add(const double* st) {  // one configuration's [count, sums, upper-triangle products]
        n += st[0];
        int pos = 1;
        for (int i = 0; i < c; ++i) S[i] += st[pos++];
        for (int i = 0; i < c; ++i)
            for (int j = i; j < c; ++j) {
                P[i + (size_t)j * c] += st[pos];
                if (i != j) P[j + (size_t)i * c] += st[pos];
                ++pos;
            }
    }
    // determinant of the unbiased covariance of the selected variables (cov = centred SSE / (n - 1))
    double det(const std::vector<int>& sel) const {
        const int k = (int)sel.size();
        if (k == 0) return 1.0;
        std::vector<double> cov((size_t)k * k);
        for (int b = 0; b < k; ++b)
            for (int a2 = 0; a2 < k; ++a2)
                cov[a2 + (size_t)b * k] = (P[sel[a2] + (size_t)sel[b] * c] - S[sel[a2]] * S[sel[b]] / n) / (n - 1.0);
        return hm::determinant(cov.data(), k);
    }
};

struct Query {
    bool xd, yd;
    int x, y;                   // variable ids
    std::vector<int> zD, zC;    // variable ids
};

struct Engine {
    pbn_mi* h;

    bool is_disc(int v) const { return v >= h->n_cont; }
    int card(int v) const { return h->card[v - h->n_cont]; }
    // categories as grouped on the device: a column with nulls has one more, the null bucket (code == card)
    int card_eff(int v) const { return card(v) + (h->disc_null.empty() ? 0 : (int)h->disc_null[v - h->n_cont]); }

    struct Plan {
        std::vector<int> cont, disc;
        int G = 1, c = 0, stats = 1;
        int g0 = 0, Gw = 0;   // window of configurations one launch row accumulates in LDS (Gw == 0: all of them)
        int window() const { return Gw ? Gw : G; }
        size_t lds() const { return ((size_t)window() * stats + 64 * (size_t)(c + 1)) * sizeof(double) + 64 * sizeof(int) + 2 * (size_t)stats + 16; }
        // configurations whose accumulators fit 60 KB of LDS next to the staging area
        int max_window() const {
            const size_t fixed = 64 * (size_t)(c + 1) * sizeof(double) + 64 * sizeof(int) + 2 * (size_t)stats + 16;
            return (int)((60 * 1024 - fixed) / (stats * sizeof(double)));
        }
    };
    Plan plan(const std::vector<int>& cont, const std::vector<int>& disc) const {
        Plan p;
        p.cont = cont; p.disc = disc;
        if (cont.size() > MI_MAX_CONT || disc.size() > MI_MAX_DISC) throw invalid_error("MutualInformation: conditioning set too large");
        int64_t G64 = 1;
        for (int v : disc) { G64 *= card(v); if (G64 > (1 << 24)) throw invalid_error("MutualInformation: too many discrete configurations"); }
        p.G = (int)G64; p.c = (int)cont.size(); p.stats = 1 + p.c + p.c * (p.c + 1) / 2;
        return p;
    }
    Plan plan(const Query& q) const {
        std::vector<int> disc, cont;
        if (q.xd) disc.push_back(q.x);
        if (q.yd) disc.push_back(q.y);
        disc.insert(disc.end(), q.zD.begin(), q.zD.end());
        if (!q.xd) cont.push_back(q.x);
        if (!q.yd) cont.push_back(q.y);
        cont.insert(cont.end(), q.zC.begin(), q.zC.end());
        return plan(cont, disc);
    }

    // per-configuration statistics (count, sums, upper-triangle products of the pilot-shifted continuous variables) of
    // several tests in ONE launch: grid row = test, so launch / sync latency is paid once per batch
    void group_stats_device(const std::vector<const Plan*>& plans, std::vector<std::vector<double>>& outs) {
        pbn_ctx* ctx = h->ctx;
        HIP_CHECK(hipSetDevice(ctx->device));
        const int64_t N = h->N, chunks = ceil_div(N, 64);
        const int B = (int)plans.size();
        const int nblocks = (int)std::min<int64_t>(chunks, B >= 8 ? 512 : (B >= 2 ? 2048 : 4096));
        std::vector<GroupArgs> descs(B);
        size_t part_doubles = 0, out_doubles = 0, lds = 0;
        int max_total = 0;
        for (int t = 0; t < B; ++t) {
            const Plan& p = *plans[t];
            const size_t total = (size_t)p.window() * p.stats;
            part_doubles += (size_t)nblocks * total;
            out_doubles += total;
            lds = std::max(lds, p.lds());
            max_total = std::max(max_total, (int)total);
        }
        const size_t desc_bytes = ((size_t)B * sizeof(GroupArgs) + 255) / 256 * 256;
        ctx->scratch_part.reserve(desc_bytes + (part_doubles + out_doubles) * sizeof(double));
        char* base = ctx->scratch_part.p;
        double* dpart = (double*)(base + desc_bytes);
        double* dout = dpart + part_doubles;
        size_t po = 0, oo = 0;
        for (int t = 0; t < B; ++t) {
            const Plan& p = *plans[t];
            GroupArgs& a = descs[t];
            a = GroupArgs{};
            a.base = h->table ? h->table->data : nullptr;
            a.ld = h->table ? h->table->ld : 0;
            a.c = p.c; a.m = (int)p.disc.size(); a.n = N; a.G = p.window(); a.g0 = p.g0; a.stats = p.stats;
            for (int i = 0; i < p.c; ++i) { a.cols[i] = p.cont[i]; a.shift[i] = h->shift[p.cont[i]]; }
            int stride = 1;
            for (int j = 0; j < a.m; ++j) { a.dvar[j] = p.disc[j] - h->n_cont; a.dstride[j] = stride; stride *= card(p.disc[j]); }
            a.codes = h->codes_dev.p; a.codes_ld = N;
            a.chunks_per_block = ceil_div(chunks, nblocks);
            a.partial = dpart + po; a.out = dout + oo;
            po += (size_t)nblocks * p.window() * p.stats; oo += (size_t)p.window() * p.stats;
        }
        HIP_CHECK(hipMemcpyAsync(base, descs.data(), (size_t)B * sizeof(GroupArgs), hipMemcpyHostToDevice, ctx->stream));
        const bool f64 = !h->table || h->table->dtype == PBN_F64;
        if (f64) hipLaunchKernelGGL(group_moments_kernel<double>, dim3(nblocks, B), dim3(64), lds, ctx->stream, (const GroupArgs*)base);
        else hipLaunchKernelGGL(group_moments_kernel<float>, dim3(nblocks, B), dim3(64), lds, ctx->stream, (const GroupArgs*)base);
        hipLaunchKernelGGL(group_reduce_kernel, dim3(max_total, B), dim3(64), 0, ctx->stream, (const GroupArgs*)base, nblocks);
        HIP_CHECK(hipGetLastError());
        std::vector<double> all(out_doubles);
        HIP_CHECK(hipMemcpyAsync(all.data(), dout, out_doubles * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        oo = 0;
        for (int t = 0; t < B; ++t) {
            const size_t total = (size_t)plans[t]->window() * plans[t]->stats;
            outs[t].assign(all.begin() + oo, all.begin() + oo + total);
            oo += total;
        }
        h->device_passes += B;
        ++h->device_launches;
    }

    // statistics of a list of plans: launch rows bounded by scratch memory.  A test with more configurations than fit
    // the LDS accumulators is covered by several rows, each accumulating one window of configurations (rows of other
    // windows are skipped), stitched back together here - still device only.
    void group_stats_legacy(const std::vector<Plan>& plans, std::vector<std::vector<double>>& outs) {
        outs.assign(plans.size(), {});
        if (h->N <= 0) { for (size_t t = 0; t < plans.size(); ++t) outs[t].assign((size_t)plans[t].G * plans[t].stats, 0.0); return; }
        const int64_t chunks = ceil_div(h->N, 64);
        std::vector<Plan> rows;                 // launch rows (windows)
        std::vector<std::pair<size_t, int>> row_of;   // (plan index, first configuration)
        for (size_t t = 0; t < plans.size(); ++t) {
            const Plan& p = plans[t];
            outs[t].assign((size_t)p.G * p.stats, 0.0);
            const int w = std::max(1, p.max_window());
            if (ceil_div(p.G, w) > 4096) throw invalid_error("MutualInformation: too many discrete configurations");
            for (int g0 = 0; g0 < p.G; g0 += w) {
                Plan r = p;
                r.g0 = g0; r.Gw = std::min(w, p.G - g0);
                rows.push_back(std::move(r));
                row_of.push_back({t, g0});
            }
        }
        std::vector<const Plan*> cur;
        std::vector<size_t> cur_idx;
        size_t cur_doubles = 0;
        auto flush = [&] {
            if (cur.empty()) return;
            std::vector<std::vector<double>> o(cur.size());
            group_stats_device(cur, o);
            for (size_t i = 0; i < cur.size(); ++i) {
                const auto [t, g0] = row_of[cur_idx[i]];
                std::copy(o[i].begin(), o[i].end(), outs[t].begin() + (size_t)g0 * plans[t].stats);
            }
            cur.clear(); cur_idx.clear(); cur_doubles = 0;
        };
        for (size_t r = 0; r < rows.size(); ++r) {
            const size_t need = (size_t)std::min<int64_t>(chunks, 512) * rows[r].window() * rows[r].stats;
            if (!cur.empty() && (cur.size() >= 256 || cur_doubles + need > ((size_t)1 << 25))) flush();   // <= 256 MB of partials
            cur.push_back(&rows[r]); cur_idx.push_back(r); cur_doubles += need;
        }
        flush();
    }

    // ---- sorted-segment path (DiscGroup) ---------------------------------------------------------------------------
    DiscGroup& group_for(const std::vector<int>& vars) {   // vars: sorted discrete variable ids
        auto it = h->groups.find(vars);
        if (it != h->groups.end()) { it->second->stamp = ++h->clock; return *it->second; }
        pbn_ctx* ctx = h->ctx;
        const int64_t N = h->N;
        const double tg0 = mi_now();
        auto g = std::make_unique<DiscGroup>();
        g->vars = vars;
        int64_t G64 = 1;
        for (int v : vars) G64 *= card_eff(v);
        g->G = (int)G64;
        g->off.assign((size_t)g->G + 1, 0);
        g->off[g->G] = N;
        if (!vars.empty()) {
            KeyArgs ka{};
            ka.m = (int)vars.size();
            int stride = 1;
            for (int j = 0; j < ka.m; ++j) { ka.dvar[j] = vars[j] - h->n_cont; ka.stride[j] = stride; stride *= card_eff(vars[j]); }
            const unsigned nb = (unsigned)ceil_div(N, 256);
            if (h->iota.n < (size_t)N) {
                h->iota.alloc((size_t)N);
                hipLaunchKernelGGL(iota_kernel, dim3(nb), dim3(256), 0, ctx->stream, h->iota.p, N);
                h->keys[0].alloc((size_t)N); h->keys[1].alloc((size_t)N);
            }
            hipLaunchKernelGGL(config_keys_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const int32_t*)h->codes_dev.p, N, ka, h->keys[0].p);
            g->perm.alloc((size_t)N);
            int bits = 1;
            while ((1ll << bits) < G64) ++bits;
            size_t tmp = 0;
            HIP_CHECK(rocprim::radix_sort_pairs(nullptr, tmp, h->keys[0].p, h->keys[1].p, h->iota.p, g->perm.p, (size_t)N, 0, bits, ctx->stream));
            h->sort_tmp.reserve(tmp);
            HIP_CHECK(rocprim::radix_sort_pairs((void*)h->sort_tmp.p, tmp, h->keys[0].p, h->keys[1].p, h->iota.p, g->perm.p, (size_t)N, 0, bits, ctx->stream));
            h->first.reserve((size_t)g->G);
            HIP_CHECK(hipMemsetAsync(h->first.p, 0xFF, (size_t)g->G * sizeof(int32_t), ctx->stream));
            hipLaunchKernelGGL(segment_first_kernel, dim3(nb), dim3(256), 0, ctx->stream, (const uint32_t*)h->keys[1].p, N, h->first.p);
            HIP_CHECK(hipGetLastError());
            std::vector<int32_t> first((size_t)g->G);
            HIP_CHECK(hipMemcpyAsync(first.data(), h->first.p, first.size() * sizeof(int32_t), hipMemcpyDeviceToHost, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));
            for (int c = g->G - 1; c >= 0; --c) g->off[c] = first[c] >= 0 ? first[c] : g->off[c + 1];
        }
        std::vector<int32_t> blk;
        g->blk_off.assign((size_t)g->G + 1, 0);
        for (int c = 0; c < g->G; ++c) {
            g->blk_off[c] = (int)(blk.size() / 4);
            for (int64_t r = g->off[c]; r < g->off[c + 1]; r += MI_SORTED_ROWS) {
                const int32_t slot = (int32_t)(blk.size() / 4);
                blk.push_back(c); blk.push_back((int32_t)r); blk.push_back((int32_t)std::min<int64_t>(r + MI_SORTED_ROWS, g->off[c + 1])); blk.push_back(slot);
            }
        }
        g->nblk = (int)(blk.size() / 4);
        g->blk_off[g->G] = g->nblk;
        blk.insert(blk.end(), g->blk_off.begin(), g->blk_off.end());   // device copy of blk_off behind the block table
        g->blk.alloc(blk.size());
        HIP_CHECK(hipMemcpyAsync(g->blk.p, blk.data(), blk.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        g->stamp = ++h->clock;
        ++h->groups_built;
        h->t_group += mi_now() - tg0;
        DiscGroup& ref = *g;
        h->groups.emplace(vars, std::move(g));
        return ref;
    }

    // The moments of ALL continuous columns per configuration of a grouping, in one segmented pass of the MFMA Gram kernel over
    // the grouping's row list: a test's [sums, products] are entries of them.  MMPC asks for hundreds to thousands of tests
    // per set of discrete variables (config 5: 514 k tests over 347 sets), each of which used to gather its own few columns
    // through the permutation - 4-byte reads 64 bytes apart, 5.8 of the 6.9 s of that MMPC.
    static bool full_gram_on() {
        static const bool v = PBN_TUNE(MI_FULLGRAM, 1) != 0;
        return v;
    }
    // host bytes of a grouping's full moments; all cached groupings together stay below PBN_MI_FULL_BUDGET_MB (default 1024) -
    // beyond it (thousands of configurations x dozens of columns x hundreds of groupings) later groupings use the per-test kernels
    size_t full_bytes(const DiscGroup& g) const { return (size_t)g.G * h->n_cont * (h->n_cont + 1) * sizeof(double); }
    bool full_applies(const DiscGroup& g) const {
        if (!(full_gram_on() && h->table && h->n_cont >= 1 && h->n_cont <= 64 && g.G <= 4096 && g.nblk > 0)) return false;
        if (g.full_ready) return true;
        static const size_t budget = (size_t)std::max(0ll, knob_ll("PBN_MI_FULL_BUDGET_MB", 1024)) << 20;
        return h->full_bytes_held + full_bytes(g) <= budget;
    }
    void ensure_full(DiscGroup& g) {
        if (g.full_ready) return;
        pbn_ctx* ctx = h->ctx;
        const double td0 = mi_now();
        const int nc = h->n_cont, nct = (nc + 15) / 16, WS = gram_ws(nct);
        if (h->shift_dev.n < (size_t)h->table->n_cols) {
            std::vector<double> sh((size_t)h->table->n_cols, 0.0);
            for (int i = 0; i < nc; ++i) sh[i] = h->shift[i];
            h->shift_dev.alloc(sh.size());
            HIP_CHECK(hipMemcpyAsync(h->shift_dev.p, sh.data(), sh.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        }
        ctx->scratch_red.reserve(((size_t)g.nblk + (size_t)g.G) * WS);
        double* partial = ctx->scratch_red.p;
        double* out = partial + (size_t)g.nblk * WS;
        GramArgs a{};
        a.base = h->table->data; a.ld = h->table->ld; a.n_cols = nc;
        for (int i = 0; i < nc; ++i) a.gc.cols[i] = i;
        a.row0 = 0; a.rows = g.vars.empty() ? nullptr : g.perm.p; a.n = h->N;
        // The first grouping with a row list builds a row-major mirror of the continuous columns (N x 16 ceil(nc / 16) elements, kept
        // with the handle): every grouping's rows are then read as whole contiguous rows - the same bytes for 4 configurations as for
        // 4096 - instead of one 64-byte sector per element of a sparse list.  MMPC builds hundreds of groupings per handle (config 5:
        // 347); the mirror costs one read and one write of the table, less than one gathered pass.  PBN_MI_MIRROR_MB (default 16384):
        // larger mirrors are not built and the rows are gathered from the columns.
        if (a.rows && !h->rowmajor_tried) {
            h->rowmajor_tried = true;
            static const size_t budget = (size_t)PBN_TUNE(MI_MIRROR_MB, 16384) << 20;
            const size_t bytes = rowmajor_mirror_elems(h->table->n_rows, nc) * dtype_size(h->table->dtype);
            if (bytes <= budget) {
                try {
                    h->rowmajor.alloc(bytes);
                    build_rowmajor_mirror(h->table->data, h->table->ld, a.gc, nc, h->table->n_rows, h->table->dtype, h->rowmajor.p, ctx->stream);
                } catch (const device_error&) {   // no room for the mirror: the rows are gathered from the columns
                    (void)hipGetLastError();
                    h->rowmajor.release();
                }
            }
        }
        a.rowmajor = (a.rows && h->rowmajor.n) ? h->rowmajor.p : nullptr;
        a.rows_per_block = MI_SORTED_ROWS; a.blk = g.blk.p;
        a.shift = h->shift_dev.p; a.partial = partial; a.num_cus = ctx->num_cus;
        // Launch order of the pieces.  A grouping's row list is increasing inside every configuration, so piece p of P_c pieces of
        // configuration c reads the table stripe around (p + 1/2) / P_c - the same 128-byte lines as the matching pieces of the
        // other configurations, each of which uses only its own rows of them.  Configuration-major (the order of the partial
        // slots) runs those pieces a whole configuration apart and every line comes from HBM once per configuration; stripe-major
        // runs them together, so that the line is fetched once and served on-die to the others.  PBN_MI_GRAM_ORDER: 0 =
        // configuration-major, 1 = stripe-major, 2 (default) = stripe-major with the pieces of stripe s on launch indices = s mod 8
        // (one XCD and its L2 under the usual round-robin; padding blocks without a piece keep the residues when few are needed).
        // The partial slots, and with them the order of every sum, do not change.
        static const int order = PBN_TUNE(MI_GRAM_ORDER, 2);
        dev_buf<int32_t> ordered;
        int nlaunch = g.nblk;
        if (order != 0 && g.G > 1) {
            int T = 1;
            for (int cg = 0; cg < g.G; ++cg) T = std::max(T, g.blk_off[cg + 1] - g.blk_off[cg]);
            std::vector<std::vector<int32_t>> cell((size_t)T);   // stripe -> slots of its pieces, configuration order
            for (int cg = 0; cg < g.G; ++cg) {
                const int P = g.blk_off[cg + 1] - g.blk_off[cg];
                for (int pc = 0; pc < P; ++pc) cell[(size_t)(((2 * (int64_t)pc + 1) * T) / (2 * (int64_t)P))].push_back(g.blk_off[cg] + pc);
            }
            auto piece = [&](std::vector<int32_t>& t, int32_t slot) {
                const int cg = (int)(std::upper_bound(g.blk_off.begin(), g.blk_off.end(), slot) - g.blk_off.begin()) - 1;
                const int64_t r = g.off[cg] + (int64_t)(slot - g.blk_off[cg]) * MI_SORTED_ROWS;
                t.push_back(cg); t.push_back((int32_t)r); t.push_back((int32_t)std::min<int64_t>(r + MI_SORTED_ROWS, g.off[cg + 1])); t.push_back(slot);
            };
            std::vector<int32_t> t;
            t.reserve(4 * (size_t)g.nblk);
            bool aligned = order >= 2;
            if (aligned) {   // groups of 8 stripes, round k of a group = the k-th piece of each of its stripes (padding where one has fewer)
                size_t cells = 0;
                for (int s0 = 0; s0 < T; s0 += 8) {
                    size_t deep = 0;
                    for (int s8 = 0; s8 < 8 && s0 + s8 < T; ++s8) deep = std::max(deep, cell[(size_t)(s0 + s8)].size());
                    cells += 8 * deep;
                }
                aligned = cells <= 2 * (size_t)g.nblk + 64;   // very uneven configurations: the padding would outnumber the pieces
            }
            if (aligned) {
                for (int s0 = 0; s0 < T; s0 += 8) {
                    size_t deep = 0;
                    for (int s8 = 0; s8 < 8 && s0 + s8 < T; ++s8) deep = std::max(deep, cell[(size_t)(s0 + s8)].size());
                    for (size_t k = 0; k < deep; ++k)
                        for (int s8 = 0; s8 < 8; ++s8) {
                            if (s0 + s8 < T && k < cell[(size_t)(s0 + s8)].size()) piece(t, cell[(size_t)(s0 + s8)][k]);
                            else { t.push_back(0); t.push_back(0); t.push_back(0); t.push_back(0); }
                        }
                }
            } else {
                for (int st = 0; st < T; ++st)
                    for (int32_t slot : cell[(size_t)st]) piece(t, slot);
            }
            nlaunch = (int)(t.size() / 4);
            ordered.alloc(t.size());
            HIP_CHECK(hipMemcpyAsync(ordered.p, t.data(), t.size() * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            HIP_CHECK(hipStreamSynchronize(ctx->stream));   // t is a local
            a.blk = ordered.p;
        }
        launch_gram_segments(a, h->table->dtype, nlaunch, g.blk.p + 4 * (size_t)g.nblk, g.G, out, ctx->stream);
        std::vector<double> hs((size_t)g.G * WS);
        HIP_CHECK(hipMemcpyAsync(hs.data(), out, hs.size() * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        g.fullS.assign((size_t)g.G * nc, 0.0);
        g.fullP.assign((size_t)g.G * nc * nc, 0.0);
        for (int cg = 0; cg < g.G; ++cg) {
            const double* w = hs.data() + (size_t)cg * WS;
            double* S = g.fullS.data() + (size_t)cg * nc;
            double* P = g.fullP.data() + (size_t)cg * nc * nc;
            for (int i = 0; i < nc; ++i) S[i] = w[WS - nct * 16 + i];
            int p = 0;
            for (int I = 0; I < nct; ++I)
                for (int J = I; J < nct; ++J, ++p) {
                    const double* tile = w + (size_t)p * 256;
                    for (int e = 0; e < 256; ++e) {
                        const int reg = e >> 6, lane = e & 63;
                        const int r = I * 16 + (lane >> 4) + 4 * reg, c = J * 16 + (lane & 15);
                        if (r >= nc || c >= nc || (I == J && r > c)) continue;
                        P[r + (size_t)c * nc] = tile[e];
                        P[c + (size_t)r * nc] = tile[e];
                    }
                }
        }
        g.full_ready = true;
        h->full_bytes_held += full_bytes(g);
        ++h->full_grams;
        ++h->device_launches;
        h->t_device += mi_now() - td0;
    }

    template <int C>
    void launch_sorted(bool f64, bool nulls, int max_nblk, int B, const SortedArgs* d) {
        const dim3 grid(max_nblk, B), block(256);
        if (f64 && !nulls) hipLaunchKernelGGL((moments_sorted_kernel<double, C, false>), grid, block, 0, h->ctx->stream, d);
        else if (f64) hipLaunchKernelGGL((moments_sorted_kernel<double, C, true>), grid, block, 0, h->ctx->stream, d);
        else if (!nulls) hipLaunchKernelGGL((moments_sorted_kernel<float, C, false>), grid, block, 0, h->ctx->stream, d);
        else hipLaunchKernelGGL((moments_sorted_kernel<float, C, true>), grid, block, 0, h->ctx->stream, d);
    }

    // one launch pair for tests with the same number of continuous variables; `items` = (plan index, its group)
    void sorted_batch(int c, bool nulls, const std::vector<Plan>& plans, const std::vector<std::pair<size_t, DiscGroup*>>& items,
                      const std::vector<const std::vector<int>*>& cmap, std::vector<std::vector<double>>& outs) {
        pbn_ctx* ctx = h->ctx;
        const double td0 = mi_now();
        const int S0 = c + c * (c + 1) / 2, S = S0 + (nulls ? 1 : 0), B = (int)items.size();
        std::vector<SortedArgs> descs(B);
        size_t part = 0, outd = 0;
        int max_nblk = 1, max_cells = 1;
        for (int i = 0; i < B; ++i) {
            const DiscGroup& g = *items[i].second;
            part += (size_t)g.nblk * S; outd += (size_t)g.G * S;
            max_nblk = std::max(max_nblk, g.nblk); max_cells = std::max(max_cells, g.G * S);
        }
        const size_t desc_bytes = ((size_t)B * sizeof(SortedArgs) + 255) / 256 * 256;
        ctx->scratch_part.reserve(desc_bytes + (part + outd) * sizeof(double));
        char* base = ctx->scratch_part.p;
        double* dpart = (double*)(base + desc_bytes);
        double* dout = dpart + part;
        size_t po = 0, oo = 0;
        for (int i = 0; i < B; ++i) {
            const Plan& p = plans[items[i].first];
            const DiscGroup& g = *items[i].second;
            SortedArgs& a = descs[i];
            a = SortedArgs{};
            a.base = h->table->data; a.ld = h->table->ld;
            for (int k = 0; k < c; ++k) { a.cols[k] = p.cont[k]; a.shift[k] = h->shift[p.cont[k]]; }
            a.perm = g.vars.empty() ? nullptr : g.perm.p;
            a.blk = g.blk.p; a.blk_off = g.blk.p + 4 * (size_t)g.nblk;
            a.nblk = g.nblk; a.G = g.G;
            a.partial = dpart + po; a.out = dout + oo;
            po += (size_t)g.nblk * S; oo += (size_t)g.G * S;
        }
        HIP_CHECK(hipMemcpyAsync(base, descs.data(), (size_t)B * sizeof(SortedArgs), hipMemcpyHostToDevice, ctx->stream));
        const bool f64 = h->table->dtype == PBN_F64;
        const SortedArgs* d = (const SortedArgs*)base;
        switch (c) {
#define PBN_MI_CASE(C) case C: launch_sorted<C>(f64, nulls, max_nblk, B, d); break;
            PBN_MI_CASE(1) PBN_MI_CASE(2) PBN_MI_CASE(3) PBN_MI_CASE(4) PBN_MI_CASE(5) PBN_MI_CASE(6) PBN_MI_CASE(7) PBN_MI_CASE(8)
            PBN_MI_CASE(9) PBN_MI_CASE(10) PBN_MI_CASE(11) PBN_MI_CASE(12) PBN_MI_CASE(13) PBN_MI_CASE(14) PBN_MI_CASE(15)
            default: launch_sorted<16>(f64, nulls, max_nblk, B, d); break;
#undef PBN_MI_CASE
        }
        hipLaunchKernelGGL(sorted_reduce_kernel, dim3(max_cells, B), dim3(64), 0, ctx->stream, d, S);
        HIP_CHECK(hipGetLastError());
        std::vector<double> all(outd);
        HIP_CHECK(hipMemcpyAsync(all.data(), dout, outd * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        oo = 0;
        for (int i = 0; i < B; ++i) {
            const size_t t = items[i].first;
            const DiscGroup& g = *items[i].second;
            const int stats = plans[t].stats;
            for (int cg = 0; cg < g.G; ++cg) {
                const int gt = (*cmap[t])[cg];
                if (gt < 0) continue;   // a configuration with a null category
                double* dst = outs[t].data() + (size_t)gt * stats + 1;
                const double* src = all.data() + oo + (size_t)cg * S;
                for (int k = 0; k < S0; ++k) dst[k] = src[k];
                if (nulls) dst[-1] = src[S0];   // rows of the configuration that are valid in all continuous columns
            }
            oo += (size_t)g.G * S;
        }
        ++h->device_launches;
        h->t_device += mi_now() - td0;
    }

    // Per-configuration statistics of a list of tests.  The counts are the segment lengths of the test's DiscGroup (no
    // data pass at all for a purely discrete test); the continuous moments come from the sorted-segment kernels, batched
    // by the number of continuous variables.  More than MI_SORTED_MAX_CONT continuous variables: the LDS-cell kernel.
    void group_stats_many(const std::vector<Plan>& plans, std::vector<std::vector<double>>& outs) {
        outs.assign(plans.size(), {});
        if (h->N <= 0) { for (size_t t = 0; t < plans.size(); ++t) outs[t].assign((size_t)plans[t].G * plans[t].stats, 0.0); return; }
        while (h->groups.size() > MI_GROUP_CACHE) {   // least recently used out (never during a batch: groups are referenced below)
            auto lru = h->groups.begin();
            for (auto it = h->groups.begin(); it != h->groups.end(); ++it)
                if (it->second->stamp < lru->second->stamp) lru = it;
            if (lru->second->full_ready) h->full_bytes_held -= std::min(h->full_bytes_held, full_bytes(*lru->second));
            h->groups.erase(lru);
        }
        std::vector<Plan> legacy;
        std::vector<size_t> legacy_idx;
        std::vector<const std::vector<int>*> cmap(plans.size(), nullptr);
        std::vector<std::vector<std::pair<size_t, DiscGroup*>>> by_c(2 * MI_SORTED_MAX_CONT + 2);   // index 2 c + (continuous nulls ? 1 : 0)
        for (size_t t = 0; t < plans.size(); ++t) {
            const Plan& p = plans[t];
            if (p.c > MI_SORTED_MAX_CONT || p.G > (1 << 22)) {
                if (h->any_null) throw invalid_error("MutualInformation: tables with nulls support at most 16 continuous variables per test");
                legacy.push_back(p); legacy_idx.push_back(t); continue;
            }
            std::vector<int> vars = p.disc;
            std::sort(vars.begin(), vars.end());
            DiscGroup& g = group_for(vars);
            outs[t].assign((size_t)p.G * p.stats, 0.0);
            // canonical configuration id (sorted variables, first fastest) -> the test's own (x, y, z order)
            auto om = g.order_maps.find(p.disc);
            if (om == g.order_maps.end()) {
                std::vector<int> map((size_t)g.G);
                const int m = (int)vars.size();
                std::vector<int> tstride(m), cards(m), ecards(m);
                for (int j = 0; j < m; ++j) {
                    cards[j] = card(vars[j]); ecards[j] = card_eff(vars[j]);
                    int stride = 1;
                    for (int v : p.disc) { if (v == vars[j]) break; stride *= card(v); }
                    tstride[j] = stride;
                }
                for (int cg = 0; cg < g.G; ++cg) {
                    int rem = cg, gt = 0;
                    for (int j = 0; j < m; ++j) {
                        const int digit = rem % ecards[j];
                        rem /= ecards[j];
                        if (digit >= cards[j]) { gt = -1; break; }   // the null bucket: these rows are not part of the test
                        gt += digit * tstride[j];
                    }
                    map[cg] = gt;
                }
                om = g.order_maps.emplace(p.disc, std::move(map)).first;
            }
            cmap[t] = &om->second;
            const std::vector<int>& map = om->second;
            for (int cg = 0; cg < g.G; ++cg)
                if (map[cg] >= 0) outs[t][(size_t)map[cg] * p.stats] = (double)(g.off[cg + 1] - g.off[cg]);
            if (p.c == 0) { ++h->count_only; continue; }
            bool nulls = false;
            for (int v : p.cont) nulls = nulls || (!h->cont_null.empty() && h->cont_null[v]);
            if (!nulls && full_applies(g)) {   // entries of the grouping's full moments
                ensure_full(g);
                const int nc = h->n_cont;
                for (int cg = 0; cg < g.G; ++cg) {
                    if (map[cg] < 0) continue;
                    double* dst = outs[t].data() + (size_t)map[cg] * p.stats + 1;
                    const double* S = g.fullS.data() + (size_t)cg * nc;
                    const double* P = g.fullP.data() + (size_t)cg * nc * nc;
                    int pos = p.c;
                    for (int i = 0; i < p.c; ++i) {
                        dst[i] = S[p.cont[i]];
                        for (int j = i; j < p.c; ++j) dst[pos++] = P[p.cont[i] + (size_t)p.cont[j] * nc];
                    }
                }
                continue;
            }
            by_c[2 * p.c + (nulls ? 1 : 0)].push_back({t, &g});
        }
        for (int key = 2; key <= 2 * MI_SORTED_MAX_CONT + 1; ++key)
            for (size_t i = 0; i < by_c[key].size(); i += 256) {
                std::vector<std::pair<size_t, DiscGroup*>> items(by_c[key].begin() + i, by_c[key].begin() + std::min(by_c[key].size(), i + 256));
                sorted_batch(key / 2, (key & 1) != 0, plans, items, cmap, outs);
            }
        h->device_passes += (int64_t)(plans.size() - legacy.size());
        if (!legacy.empty()) {
            std::vector<std::vector<double>> lo;
            const double tl = mi_now();
            group_stats_legacy(legacy, lo);
            h->t_device += mi_now() - tl;
            for (size_t i = 0; i < legacy.size(); ++i) outs[legacy_idx[i]].swap(lo[i]);
        }
    }

    void group_stats(const std::vector<int>& cont, const std::vector<int>& disc, int G, std::vector<double>& out) {
        std::vector<Plan> one{plan(cont, disc)};
        (void)G;
        std::vector<std::vector<double>> o;
        group_stats_many(one, o);
        out.swap(o[0]);
    }

    // MI(X; Y | Z) (mutual_information.cpp:926-1055 no conditioning, :1139-1312 one variable, :1391-1658 general);
    // every overload of the reference is the general formula with the matching emptiness of zD / zC.
    double mi(const Query& q, double* rows = nullptr) {
        std::vector<Plan> one{plan(q)};
        std::vector<std::vector<double>> o;
        group_stats_many(one, o);
        if (rows) *rows = rows_of(one[0], o[0]);
        return mi_from_stats(q, one[0], o[0]);
    }

    // rows of the test: all of them, or - with nulls - those valid in every variable of the test (the counts add up to it)
    double rows_of(const Plan& pl, const std::vector<double>& st) const {
        if (!h->any_null) return (double)h->N;
        double n = 0;
        for (int g = 0; g < pl.G; ++g) n += st[(size_t)g * pl.stats];
        return n;
    }

    double mi_from_stats(const Query& q, const Plan& pl, const std::vector<double>& st) {
        const std::vector<int>& disc = pl.disc;
        const int G = pl.G, c = pl.c, zc = (int)q.zC.size();
        const int stats = pl.stats;
        const double N = rows_of(pl, st);
        auto S = [&](int g) { return st.data() + (size_t)g * stats; };
        std::vector<int> selz(zc);
        double mi = 0.0;
        if (q.xd && q.yd) {
            const int cx = card(q.x), cy = card(q.y), vars = cx * cy, zcat = G / vars;
            for (int i = 0; i < zc; ++i) selz[i] = i;
            for (int k = 0; k < zcat; ++k) {
                const int off = k * vars;
                double Nz = 0;
                std::vector<double> Nxz(cx, 0.0), Nyz(cy, 0.0);
                for (int i = 0; i < cx; ++i)
                    for (int j = 0; j < cy; ++j) {
                        const double nn = S(off + i + j * cx)[0];
                        Nxz[i] += nn; Nyz[j] += nn; Nz += nn;
                    }
                if (Nz == 0) continue;
                const double pz = Nz / N;
                for (int i = 0; i < cx; ++i) {
                    const double pxz = Nxz[i] / N;
                    for (int j = 0; j < cy; ++j) {
                        const double nn = S(off + i + j * cx)[0];
                        if (nn == 0) continue;
                        const double pyz = Nyz[j] / N, pxyz = nn / N;
                        double term = std::log((pz * pxyz) / (pxz * pyz));
                        if (zc > 0) { Mom mo(c); mo.add(S(off + i + j * cx)); term -= entropy_mvn(zc, mo.det(selz)); }
                        mi += pxyz * term;
                    }
                }
                if (zc == 0) continue;
                for (int i = 0; i < cx; ++i) {
                    if (Nxz[i] == 0) continue;
                    Mom mo(c);
                    for (int j = 0; j < cy; ++j) mo.add(S(off + i + j * cx));
                    mi += (Nxz[i] / N) * entropy_mvn(zc, mo.det(selz));
                }
                for (int j = 0; j < cy; ++j) {
                    if (Nyz[j] == 0) continue;
                    Mom mo(c);
                    for (int i = 0; i < cx; ++i) mo.add(S(off + i + j * cx));
                    mi += (Nyz[j] / N) * entropy_mvn(zc, mo.det(selz));
                }
                Mom mo(c);
                for (int i = 0; i < vars; ++i) mo.add(S(off + i));
                mi -= pz * entropy_mvn(zc, mo.det(selz));
            }
            // the purely discrete overloads return the sum as it is (mutual_information.cpp:955,1445); the
            // conditional-Gaussian ones clamp at zero (:1530)
            return zc == 0 ? mi : std::max(mi, 0.0);
        }
        if (q.xd != q.yd) {  // one discrete (leading in disc), one continuous (leading in cont)
            const int cx = card(disc[0]), zcat = G / cx;
            std::vector<int> sely(zc + 1);
            for (int i = 0; i <= zc; ++i) sely[i] = i;
            for (int i = 0; i < zc; ++i) selz[i] = i + 1;
            for (int k = 0; k < zcat; ++k) {
                const int off = k * cx;
                double Nz = 0;
                for (int i = 0; i < cx; ++i) Nz += S(off + i)[0];
                if (Nz == 0) continue;
                const double pz = Nz / N;
                Mom pool(c);
                for (int i = 0; i < cx; ++i) {
                    const double nn = S(off + i)[0];
                    pool.add(S(off + i));
                    if (nn == 0) continue;
                    Mom mo(c);
                    mo.add(S(off + i));
                    const double pxz = nn / N;
                    mi -= pxz * entropy_mvn(zc + 1, mo.det(sely));
                    if (zc > 0) mi += pxz * entropy_mvn(zc, mo.det(selz));
                }
                mi += pz * entropy_mvn(zc + 1, pool.det(sely));
                if (zc > 0) mi -= pz * entropy_mvn(zc, pool.det(selz));
            }
            return std::max(mi, 0.0);
        }
        // both continuous: cont = [x, y, zC...]
        std::vector<int> selxyz(zc + 2), selxz(zc + 1), selyz(zc + 1);
        for (int i = 0; i < zc + 2; ++i) selxyz[i] = i;
        selxz[0] = 0; selyz[0] = 1;
        for (int i = 0; i < zc; ++i) { selxz[i + 1] = i + 2; selyz[i + 1] = i + 2; selz[i] = i + 2; }
        for (int k = 0; k < G; ++k) {
            const double Nz = S(k)[0];
            if (Nz == 0) continue;
            const double pz = Nz / N;
            Mom mo(c);
            mo.add(S(k));
            mi += pz * (entropy_mvn(zc + 1, mo.det(selxz)) + entropy_mvn(zc + 1, mo.det(selyz)) - entropy_mvn(zc + 2, mo.det(selxyz)));
            if (zc > 0) mi -= pz * entropy_mvn(zc, mo.det(selz));
        }
        // without any conditioning the reference uses -1/2 log(1 - cor^2) unclamped (:1055-1062); same value
        return (q.zD.empty() && zc == 0) ? mi : std::max(mi, 0.0);
    }

    // degrees of freedom of 2 N MI (mutual_information.cpp:1093-1123, 1314-1375, 1660-1731)
    double df(const Query& q) const {
        double llz = 1;
        for (int v : q.zD) llz *= card(v);
        const double zc = (double)q.zC.size();
        if (q.xd && q.yd) {
            const double base = (card(q.x) - 1.0) * (card(q.y) - 1.0) * llz;
            return h->asymptotic ? base * (1 + 0.5 * (zc * (zc + 3))) : base * (1 + 0.5 * (zc * (zc + 1)));
        }
        if (q.xd != q.yd) {
            const double llx = card(q.xd ? q.x : q.y);
            return h->asymptotic ? (llx - 1) * llz * (zc + 2) : (llx - 1) * llz * (zc + 1);
        }
        return llz;
    }

    Query make(int v1, int v2, int k, const int* cond) const {
        const int nv = h->n_cont + h->n_disc;
        auto chk = [&](int v) { if (v < 0 || v >= nv) throw invalid_error("MutualInformation: variable index out of range"); };
        chk(v1); chk(v2);
        Query q;
        q.xd = is_disc(v1); q.yd = is_disc(v2); q.x = v1; q.y = v2;
        if (!q.xd && q.yd) { std::swap(q.x, q.y); std::swap(q.xd, q.yd); }  // mi(): the discrete one is passed first
        for (int i = 0; i < k; ++i) { chk(cond[i]); (is_disc(cond[i]) ? q.zD : q.zC).push_back(cond[i]); }
        return q;
    }
};

}  // namespace

extern "C" {

// table: the continuous columns (device, may be NULL when n_cont == 0); codes[j]: N int32 category indices of the
// j-th discrete column (HOST), cardinality[j] its number of categories.  Variable ids: continuous columns first
// (table order), then the discrete ones.
int pbn_mi_create(pbn_ctx* ctx, const pbn_table* table, int64_t n_rows, int n_disc, const int32_t* const* codes,
                  const int* cardinality, int asymptotic_df, pbn_mi** out) {
    return guarded(mu_of(ctx), [&] {
        if (!ctx || !out || (n_disc > 0 && (!codes || !cardinality))) throw invalid_error("pbn_mi_create: null argument");
        if (table && table->n_rows != n_rows) throw invalid_error("pbn_mi_create: row counts differ");
        HIP_CHECK(hipSetDevice(ctx->device));
        auto h = std::make_unique<pbn_mi>();
        h->ctx = ctx; h->table = table; h->n_cont = table ? table->n_cols : 0; h->n_disc = n_disc; h->N = n_rows;
        h->asymptotic = asymptotic_df != 0;
        h->card.assign(cardinality, cardinality + n_disc);
        h->disc_null.assign(n_disc, 0);
        h->cont_null.assign(h->n_cont, 0);
        h->codes_dev.alloc((size_t)std::max<int64_t>(1, (int64_t)n_disc * n_rows));
        std::vector<int32_t> bucketed;
        for (int j = 0; j < n_disc; ++j) {
            bool has_null = false;
            for (int64_t r = 0; r < n_rows; ++r) {
                if (codes[j][r] < -1 || codes[j][r] >= cardinality[j]) throw invalid_error("pbn_mi_create: category index out of range");
                has_null = has_null || codes[j][r] == -1;
            }
            const int32_t* src = codes[j];
            if (has_null) {   // a null cell (code -1) goes to one more bucket, card[j], which no test ever reads
                bucketed.assign(codes[j], codes[j] + n_rows);
                for (int32_t& c : bucketed) if (c < 0) c = cardinality[j];
                src = bucketed.data();
                h->disc_null[j] = 1; h->any_null = true;
            }
            HIP_CHECK(hipMemcpyAsync(h->codes_dev.p + (size_t)j * n_rows, src, (size_t)n_rows * sizeof(int32_t), hipMemcpyHostToDevice, ctx->stream));
            if (has_null) HIP_CHECK(hipStreamSynchronize(ctx->stream));   // `bucketed` is reused
        }
        h->shift.assign(h->n_cont, 0.0);
        if (table && n_rows > 0) {
            dev_buf<double> dshift((size_t)h->n_cont);
            for (int c0 = 0; c0 < h->n_cont; c0 += 64) {
                GramCols gc{};
                const int d = std::min(64, h->n_cont - c0);
                for (int i = 0; i < d; ++i) gc.cols[i] = c0 + i;
                launch_pilot(table->data, table->ld, gc, d, 0, nullptr, n_rows, table->dtype, dshift.p, ctx->stream);
            }
            HIP_CHECK(hipMemcpyAsync(h->shift.data(), dshift.p, h->n_cont * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        }
        HIP_CHECK(hipStreamSynchronize(ctx->stream));
        *out = h.release();
    });
}

// Continuous columns that hold NaN for null cells (flags[i] != 0) and the pilot shift to use for them (their own pilot
// would be NaN).  A test then runs over the rows valid in all of its variables, as the reference's contains_null
// overloads do (hybrid/mutual_information.cpp:152-215).
int pbn_mi_set_continuous_nulls(pbn_mi* h, const unsigned char* flags, const double* shift) {
    return guarded(mu_of(h), [&] {
        if (!h || !flags || !shift) throw invalid_error("pbn_mi_set_continuous_nulls: null argument");
        for (int i = 0; i < h->n_cont; ++i) {
            h->cont_null[i] = flags[i] ? 1 : 0;
            if (flags[i]) { h->shift[i] = shift[i]; h->any_null = true; }
        }
    });
}

void pbn_mi_destroy(pbn_mi* h) {
    if (!h) return;
    pbn::ctx_pin pin_(h->ctx);
    std::lock_guard<std::recursive_mutex> lock_(mu_of(h));
    (void)hipSetDevice(h->ctx->device);
    (void)hipStreamSynchronize(h->ctx->stream);
    if (PBN_TUNE(MI_TIMING, 0) == 1)
        std::fprintf(stderr, "[pbn_mi] tests %lld (count-only %lld), batches %lld, launches %lld, groupings built %lld: build %.2f s, "
                     "plan/map %.2f s, device %.2f s, host statistics %.2f s\n", (long long)h->device_passes, (long long)h->count_only,
                     (long long)h->batches, (long long)h->device_launches, (long long)h->groups_built, h->t_group, h->t_prep, h->t_device, h->t_host);
    delete h;
}

// MI(v1; v2 | cond) and the degrees of freedom of its chi-square statistic; either output may be NULL.
int pbn_mi_value(pbn_mi* h, int v1, int v2, int n_cond, const int* cond, double* mi, double* df) {
    return guarded(mu_of(h), [&] {
        if (!h || (n_cond > 0 && !cond)) throw invalid_error("pbn_mi_value: null argument");
        Engine e{h};
        const Query q = e.make(v1, v2, n_cond, cond);
        if (mi) *mi = e.mi(q);
        if (df) *df = e.df(q);
    });
}

// pbn_ci_pvalue_fn over a pbn_mi handle: P(chi2_df > 2 N MI) (mutual_information.cpp:1125-1137,1377-1389,1733-1750)
double pbn_mi_pvalue(void* user, int v1, int v2, int n_cond, const int* cond) {
    pbn_mi* h = (pbn_mi*)user;
    double mi = 0, df = 0;
    std::vector<int> mapped;
    if (h && !h->order.empty()) {
        const int no = (int)h->order.size();
        if (v1 < 0 || v2 < 0 || v1 >= no || v2 >= no) return std::nan("");
        v1 = h->order[v1]; v2 = h->order[v2];
        mapped.resize(n_cond);
        for (int i = 0; i < n_cond; ++i) {
            if (cond[i] < 0 || cond[i] >= no) return std::nan("");
            mapped[i] = h->order[cond[i]];
        }
        cond = mapped.data();
    }
    double rows = 0;
    const int rc = guarded([&] {
        if (!h || (n_cond > 0 && !cond)) throw invalid_error("pbn_mi_pvalue: null argument");
        Engine e{h};
        const Query q = e.make(v1, v2, n_cond, cond);
        mi = e.mi(q, &rows);
        df = e.df(q);
    });
    if (rc != PBN_OK) return std::nan("");
    return gamma_q(0.5 * df, 0.5 * (mi * 2.0 * rows));
}

// LinearCorrelation::pvalue on a table with nulls (continuous/linearcorrelation.cpp:20-122, the pvalue_impl branch): the
// covariance of [v1, v2, cond...] over the rows valid in all of them - one pass of the NaN-skipping moments kernel - then
// the same partial-correlation t-test as the cached form, with `valid rows - 2 - |cond|` degrees of freedom.
// pbn_ci_pvalue_fn signature over a pbn_mi handle whose variables are all continuous.
double pbn_mi_lincor_pvalue(void* user, int v1, int v2, int n_cond, const int* cond) {
    pbn_mi* h = (pbn_mi*)user;
    double result = std::nan("");
    (void)guarded([&] {
        if (!h || (n_cond > 0 && !cond)) throw invalid_error("pbn_mi_lincor_pvalue: null argument");
        std::vector<int> vars{v1, v2};
        vars.insert(vars.end(), cond, cond + n_cond);
        if (!h->order.empty())
            for (int& v : vars) {
                if (v < 0 || v >= (int)h->order.size()) throw invalid_error("LinearCorrelation: variable index out of range");
                v = h->order[v];
            }
        for (int v : vars)
            if (v < 0 || v >= h->n_cont) throw invalid_error("LinearCorrelation: variable is not continuous");
        Engine e{h};
        std::vector<double> st;
        e.group_stats(vars, {}, 1, st);
        const int c = (int)vars.size();
        const double n = st[0];
        if (!(n > c)) throw invalid_error("LinearCorrelation: not enough valid rows");
        std::vector<double> cov((size_t)c * c);
        int pos = 1 + c;
        for (int i = 0; i < c; ++i)
            for (int j = i; j < c; ++j) {
                const double v = (st[pos++] - st[1 + i] * st[1 + j] / n) / (n - 1.0);
                cov[i + (size_t)j * c] = cov[j + (size_t)i * c] = v;
            }
        pbn_lincor* lc = nullptr;
        if (pbn_lincor_from_cov(c, (int64_t)n, cov.data(), &lc) != PBN_OK) throw device_error(pbn_last_error());
        std::vector<int> zc(std::max(1, n_cond));
        for (int i = 0; i < n_cond; ++i) zc[i] = 2 + i;
        result = pbn_lincor_pvalue(lc, 0, 1, n_cond, zc.data());
        pbn_lincor_destroy(lc);
    });
    return result;
}

// Batched form: n_tests independence tests in as few launches as scratch memory allows (pbn_ci_pvalue_batch_fn).
// cond_off has n_tests + 1 entries into cond.  NaN in out[i] marks a failed test (pbn_last_error has the last reason).
void pbn_mi_pvalue_batch(void* user, int n_tests, const int* v1, const int* v2, const int* cond_off, const int* cond, double* out) {
    pbn_mi* h = (pbn_mi*)user;
    for (int i = 0; i < n_tests; ++i) out[i] = std::nan("");
    (void)guarded([&] {
        if (!h || !v1 || !v2 || !cond_off || !out) throw invalid_error("pbn_mi_pvalue_batch: null argument");
        Engine e{h};
        auto map = [&](int v) {
            if (h->order.empty()) return v;
            if (v < 0 || v >= (int)h->order.size()) throw invalid_error("MutualInformation: variable index out of range");
            return h->order[v];
        };
        std::vector<Query> qs(n_tests);
        std::vector<Engine::Plan> plans(n_tests);
        std::vector<int> cm;
        for (int i = 0; i < n_tests; ++i) {
            cm.clear();
            for (int j = cond_off[i]; j < cond_off[i + 1]; ++j) cm.push_back(map(cond[j]));
            qs[i] = e.make(map(v1[i]), map(v2[i]), (int)cm.size(), cm.data());
            plans[i] = e.plan(qs[i]);
        }
        std::vector<std::vector<double>> st;
        const double t0 = mi_now(), g0 = h->t_group, d0 = h->t_device;
        e.group_stats_many(plans, st);
        const double t1 = mi_now();
        h->t_prep += (t1 - t0) - (h->t_group - g0) - (h->t_device - d0);
        // determinants, entropies and chi-square tails of the tests: independent host arithmetic on their statistics - spread over
        // a few threads when the batch is large (config 5's MMPC: 0.9 s of it on one thread, once the moments cost 0.1 s)
        auto finish = [&](int i) {
            const double mi = e.mi_from_stats(qs[i], plans[i], st[i]);
            out[i] = gamma_q(0.5 * e.df(qs[i]), 0.5 * (mi * 2.0 * e.rows_of(plans[i], st[i])));
        };
        static const int max_threads = [] {
            const int hw = (int)std::thread::hardware_concurrency();
            const int n = knob_int("PBN_MI_THREADS", std::min(hw > 0 ? hw : 1, 16));
            return n < 1 ? 1 : n;
        }();
        const int nth = std::min(max_threads, n_tests / 64);
        if (nth <= 1) {
            for (int i = 0; i < n_tests; ++i) finish(i);
        } else {
            std::vector<std::thread> pool;
            std::vector<std::exception_ptr> errs((size_t)nth);
            for (int w = 0; w < nth; ++w)
                pool.emplace_back([&, w] {
                    try {
                        for (int i = w; i < n_tests; i += nth) finish(i);
                    } catch (...) {
                        errs[w] = std::current_exception();
                    }
                });
            for (auto& th : pool) th.join();
            for (auto& ep : errs)
                if (ep) std::rethrow_exception(ep);
        }
        h->t_host += mi_now() - t1;
        ++h->batches;
    });
}

// Index space of pbn_mi_pvalue: external index i stands for variable ids[i] (n == 0 restores the identity).
int pbn_mi_set_order(pbn_mi* h, int n, const int* ids) {
    return guarded(mu_of(h), [&] {
        if (!h || (n > 0 && !ids)) throw invalid_error("pbn_mi_set_order: null argument");
        for (int i = 0; i < n; ++i)
            if (ids[i] < 0 || ids[i] >= h->n_cont + h->n_disc) throw invalid_error("pbn_mi_set_order: variable id out of range");
        h->order.assign(ids, ids + n);
    });
}

// ChiSquare::pvalue (learning/independences/discrete/chi_square.cpp:8-139) over the discrete columns of a pbn_mi handle:
// Pearson's statistic summed over the configurations of the conditioning set, df = (|X|-1)(|Y|-1) prod |Z|.  Counts come
// from the same device pass as the mutual information (no continuous statistics).  Expected counts are formed in
// double; the reference multiplies two int marginals (chi_square.cpp:21,62,116), which overflows beyond ~46 000 rows
// per cell pair.  pbn_ci_pvalue_fn signature, indices mapped through pbn_mi_set_order when set.
double pbn_chisq_pvalue(void* user, int v1, int v2, int n_cond, const int* cond) {
    pbn_mi* h = (pbn_mi*)user;
    double result = std::nan("");
    const int rc = guarded([&] {
        if (!h || (n_cond > 0 && !cond)) throw invalid_error("pbn_chisq_pvalue: null argument");
        std::vector<int> vars{v1, v2};
        vars.insert(vars.end(), cond, cond + n_cond);
        if (!h->order.empty())
            for (int& v : vars) {
                if (v < 0 || v >= (int)h->order.size()) throw invalid_error("ChiSquare: variable index out of range");
                v = h->order[v];
            }
        Engine e{h};
        int64_t G = 1;
        for (int v : vars) {
            if (v < h->n_cont || v >= h->n_cont + h->n_disc) throw invalid_error("ChiSquare: variable is not categorical");
            G *= e.card(v);
            if (G > (1 << 24)) throw invalid_error("ChiSquare: too many discrete configurations");
        }
        std::vector<double> st;
        e.group_stats({}, vars, (int)G, st);
        const int cx = e.card(vars[0]), cy = e.card(vars[1]), vc = cx * cy, zc = (int)(G / vc);
        double statistic = 0;
        for (int k = 0; k < zc; ++k) {
            const double* c = st.data() + (size_t)k * vc;
            std::vector<double> mx(cx, 0.0), my(cy, 0.0);
            double tot = 0;
            for (int i = 0; i < cx; ++i)
                for (int j = 0; j < cy; ++j) { mx[i] += c[i + j * cx]; my[j] += c[i + j * cx]; tot += c[i + j * cx]; }
            if (tot == 0) continue;
            const double inv = 1.0 / tot;
            for (int i = 0; i < cx; ++i)
                for (int j = 0; j < cy; ++j) {
                    const double expected = mx[i] * my[j] * inv;
                    if (expected != 0) { const double dd = c[i + j * cx] - expected; statistic += dd * dd / expected; }
                }
        }
        if (n_cond > 1 && statistic < 1.4901161193847656e-08) { result = 1.0; return; }   // chi_square.cpp:130-134
        const double df = (cx - 1.0) * (cy - 1.0) * zc;
        result = gamma_q(0.5 * df, 0.5 * statistic);
    });
    return rc == PBN_OK ? result : std::nan("");
}

// Joint counts of discrete variables (factors/discrete/discrete_indices.cpp:134-150 joint_counts): out[sum_i code_i *
// stride_i], the first variable fastest, prod(cardinality) entries, rows with a null in any of the variables left out.
// The counts are the segment lengths of the cached row grouping of that variable set.
int pbn_mi_counts(pbn_mi* h, int n_vars, const int* vars, double* out) {
    return guarded(mu_of(h), [&] {
        if (!h || !vars || !out || n_vars < 1) throw invalid_error("pbn_mi_counts: null argument");
        std::vector<int> v(vars, vars + n_vars);
        if (!h->order.empty())
            for (int& x : v) {
                if (x < 0 || x >= (int)h->order.size()) throw invalid_error("pbn_mi_counts: variable index out of range");
                x = h->order[x];
            }
        for (int x : v)
            if (x < h->n_cont || x >= h->n_cont + h->n_disc) throw invalid_error("pbn_mi_counts: variable is not categorical");
        Engine e{h};
        std::vector<double> st;
        e.group_stats({}, v, 1, st);
        std::copy(st.begin(), st.end(), out);
    });
}

int pbn_mi_stats(const pbn_mi* h, int64_t* device_passes, int64_t* host_passes) {
    return guarded(mu_of(h), [&] {
        if (!h) throw invalid_error("pbn_mi_stats: null argument");
        if (device_passes) *device_passes = h->device_passes;
        if (host_passes) *host_passes = h->host_passes;
    });
}

}  // extern "C"
