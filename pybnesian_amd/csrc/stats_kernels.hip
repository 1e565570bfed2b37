// Column statistics on gfx950: pilot-shifted centred Gram (sum of cross products) + column sums in ONE
// pass over the rows, on the f64 matrix pipe (v_mfma_f64_16x16x4_f64).
//
// Replaces DataFrame::means / cov / sse of the reference (/root/reference/pybnesian/dataset/
// dataset.hpp:208-234, 340-512), which centre a copy of every column and then run d(d+1)/2
// length-N dot products on one CPU thread.  Consumers: bandwidth selectors
// (kde/NormalReferenceRule.hpp:72-134), MLE<LinearGaussianCPD> (learning/parameters/
// mle_LinearGaussianCPD.hpp:11-193, via the normal equations), BGe's cached SSE (learning/scores/
// bge.hpp:52-72).
//
// Layout: the table is column-major (Arrow columns).  For a 16-row chunk a lane (c = lane&15,
// kq = lane>>4) reads rows 4kq..4kq+3 of column 16*I + c; k-step j of the chunk multiplies, for every
// column-tile pair I <= J, A[i][k] = x[row 4k+j][col 16I+i] with B[k][jj] = x[row 4k+j][col 16J+jj].
// Row order inside the contraction is irrelevant, so no transpose through LDS is needed.
// Numerics: values are shifted by a per-column pilot mean (mean of the first <=1024 rows; the shift array
// is indexed by TABLE column so that Grams of different row ranges / column subsets are additive) before the
// products, so  SSE = G_shift - S S^T / N  has no catastrophic cancellation (S = shifted column sums).
#include <cstdlib>
#include <type_traits>

#include "common.hpp"
#include "stats_kernels.hpp"

namespace pbn {

typedef double d4 __attribute__((ext_vector_type(4)));

template <typename T>
__global__ __launch_bounds__(256) void pilot_mean_kernel(const void* base, int64_t ld, GramCols gc, int n_cols,
                                                          int64_t row0, const int32_t* rows, int64_t n,
                                                          double* shift) {
    const int c = blockIdx.x;
    if (c >= n_cols) return;
    const T* col = (const T*)base + (int64_t)gc.cols[c] * ld;
    const int64_t m = n < 1024 ? n : 1024;
    double v = 0.0;
    for (int64_t i = threadIdx.x; i < m; i += 256) v += (double)col[rows ? (int64_t)rows[i] : row0 + i];
    __shared__ double red[256];
    red[threadIdx.x] = v;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) shift[gc.cols[c]] = m > 0 ? red[0] / (double)m : 0.0;  // indexed by TABLE column
}

template <int NCT>
struct NPairs {
    static constexpr int value = NCT * (NCT + 1) / 2;
};

// partial layout per block: [NP][256] tile accumulators (lane-major: element lane*4 + reg), then [NCT*16] sums
// 16 bytes of a column with only element alignment guaranteed (global_load_dwordx4 asks for no more): vector types with
// their alignment lowered, so that the load stays one instruction
typedef double d2_elem_aligned __attribute__((ext_vector_type(2), aligned(8)));
typedef float f4_elem_aligned __attribute__((ext_vector_type(4), aligned(4)));
template <typename T>
__device__ __forceinline__ void load_rows4(const T* p, T (&out)[4]);
template <>
__device__ __forceinline__ void load_rows4<double>(const double* p, double (&out)[4]) {
    const d2_elem_aligned lo = *(const d2_elem_aligned*)p, hi = *(const d2_elem_aligned*)(p + 2);
    out[0] = lo[0]; out[1] = lo[1]; out[2] = hi[0]; out[3] = hi[1];
}
template <>
__device__ __forceinline__ void load_rows4<float>(const float* p, float (&out)[4]) {
    const f4_elem_aligned v = *(const f4_elem_aligned*)p;
    out[0] = v[0]; out[1] = v[1]; out[2] = v[2]; out[3] = v[3];
}
template <typename T, int NCT, bool GATHER>
__global__ __launch_bounds__(256, 2) void gram_kernel(GramArgs a) {
    constexpr int NP = NPairs<NCT>::value;
    const int lane = threadIdx.x & 63;
    const int wave = threadIdx.x >> 6;
    const int c = lane & 15, kq = lane >> 4;

    const int64_t rb0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t rb1 = rb0 + a.rows_per_block;
    if (rb1 > a.n) rb1 = a.n;

    const T* colp[NCT];
    double sh[NCT];
    bool cvalid[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) {
        const int ci = I * 16 + c;
        cvalid[I] = ci < a.n_cols;
        const int src = cvalid[I] ? a.gc.cols[ci] : a.gc.cols[0];
        colp[I] = (const T*)a.base + (int64_t)src * a.ld + (GATHER ? 0 : a.row0);
        sh[I] = cvalid[I] ? a.shift[src] : 0.0;
    }

    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0, 0, 0, 0};
    double csum[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) csum[I] = 0.0;

    // software-pipelined over 16-row chunks: the loads of chunk i+1 are in flight under the MFMAs of chunk i
    auto load_chunk = [&](int64_t r, T (&raw)[NCT][4], bool (&ok)[4]) {
        const int64_t rl = r + 4 * kq;
        int64_t src[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            ok[j] = rl + j < rb1;
            src[j] = ok[j] ? (GATHER ? (int64_t)a.rows[rl + j] : rl + j) : (GATHER ? (int64_t)a.rows[rb0] : rb0);
        }
        if (!GATHER && ok[3]) {
            // the lane's four rows are consecutive in memory: one 16- / two 16-byte loads per column instead of four
            // 4- / 8-byte ones (element alignment is all the hardware asks for), half the sectors the L1 has to serve
#pragma unroll
            for (int I = 0; I < NCT; ++I) load_rows4<T>(colp[I] + rl, raw[I]);
            return;
        }
#pragma unroll
        for (int I = 0; I < NCT; ++I)
#pragma unroll
            for (int j = 0; j < 4; ++j) raw[I][j] = colp[I][src[j]];
    };
    T raw[NCT][4];
    bool ok[4];
    int64_t r = rb0 + wave * 16;
    if (r < rb1) load_chunk(r, raw, ok);
    for (; r < rb1; r += 64) {
        double x[NCT][4];
#pragma unroll
        for (int I = 0; I < NCT; ++I)
#pragma unroll
            for (int j = 0; j < 4; ++j) x[I][j] = (ok[j] && cvalid[I]) ? (double)raw[I][j] - sh[I] : 0.0;
        if (r + 64 < rb1) load_chunk(r + 64, raw, ok);
#pragma unroll
        for (int I = 0; I < NCT; ++I) csum[I] += (x[I][0] + x[I][1]) + (x[I][2] + x[I][3]);
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            int p = 0;
#pragma unroll
            for (int I = 0; I < NCT; ++I)
#pragma unroll
                for (int J = I; J < NCT; ++J) {
                    acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[I][j], x[J][j], acc[p], 0, 0, 0);
                    ++p;
                }
        }
    }

    // ---- block combine (fixed wave order) -------------------------------------------------------
    // one LDS image, waves add into it in wave order 0,1,2,3 (deterministic)
    __shared__ double lds[NP * 256 + NCT * 16];
    constexpr int WS = NP * 256 + NCT * 16;
    double cs[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) {
        double s = csum[I];
        s += __shfl_xor(s, 16);
        s += __shfl_xor(s, 32);
        cs[I] = s;
    }
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = p * 256 + i * 64 + lane;
                    lds[e] = (w == 0) ? acc[p][i] : lds[e] + acc[p][i];
                }
            }
            if (kq == 0) {
#pragma unroll
                for (int I = 0; I < NCT; ++I) {
                    const int e = NP * 256 + I * 16 + c;
                    lds[e] = (w == 0) ? cs[I] : lds[e] + cs[I];
                }
            }
        }
        __syncthreads();
    }
    double* out = a.partial + (int64_t)blockIdx.x * WS;
    for (int e = threadIdx.x; e < WS; e += 256) out[e] = lds[e];
}

// ------------------------------------------------------------------------------------------------------------------
// gram_lds_kernel: the same statistics with the rows staged through LDS and the K-STEPS split over the waves.
// gram_kernel above gives every wave its own rows: all NCT(NCT+1)/2 accumulator tiles AND four k-steps of operands and
// their prefetch live in registers (NCT = 4: 196 VGPRs, 2 waves / SIMD, each alternating between its load phase and its
// MFMA phase: 0.52 of the HBM peak at 64 columns).  Here a 256-thread block loads a chunk of 32 rows x all columns ONCE,
// coalesced (4 threads per column, 64 contiguous bytes each), subtracts the pilot shift and writes doubles to LDS; wave w
// then multiplies k-steps 2w and 2w + 1 of the chunk (rows 8w .. 8w + 7) into ALL tile pairs: per k-step NCT operand
// fragments from LDS feed NCT(NCT+1)/2 MFMAs, every wave has the same work (20 MFMAs per chunk at NCT = 4), and the
// registers hold the accumulators (80) plus one k-step of operands - ~120 VGPRs, 4 waves / SIMD, 4 blocks per CU.  The
// global loads of chunk i + 1 are in flight (registers) under the MFMAs of chunk i (double-buffered LDS, one barrier per
// chunk); the four waves' accumulators are added through LDS in wave order at the end (deterministic).
// Two earlier splits were measured and dropped: tile PAIRS over the waves (10 pairs = 3 + 3 + 2 + 2: the barrier per chunk
// waits for the 3-pair waves, and every wave reads two fragments per MFMA from LDS - 216 us for the MFMA side alone against
// 130 us of MFMA work; with the compiler's ds_read2_b64 fusion the LDS reads alone took as long as the MFMAs), and a
// prefetch distance of two chunks (158 VGPRs, a block per CU lost: 320 us).
// LDS image: [k-step s (8)][column group I][half h = k >> 1][column c (16)][k & 1] - the 32 lanes of a half wave
// (lane = 16 k + c, k in {0, 1} or {2, 3}) read 32 CONSECUTIVE doubles of k-step s, group I: one conflict-free
// ds_read_b64 (2 LDS cycles, 256 B/clk; MI355X_MICROARCH.md "LDS").  The k-steps are 2064 B apart on purpose: at <= 2040 B
// or at a multiple of 512 B the compiler fuses reads of neighbouring k-steps into ds_read2[st64]_b64, which run at half the
// rate (8 cycles per KiB, 32-bank mapping).
// ------------------------------------------------------------------------------------------------------------------
constexpr int GL_ROWS = 32;                         // rows per chunk
constexpr int GL_KSTRIDE = 258;                     // doubles between k-steps in the LDS image: 4 groups x 64, + 16 B

template <typename T, int NCT, bool GATHER>
__global__ __launch_bounds__(256, 4) void gram_lds_kernel(GramArgs a) {
    constexpr int NP = NPairs<NCT>::value;
    constexpr int NC = NCT * 16;
    constexpr int WS = NP * 256 + NCT * 16;
    constexpr int IMG = (GL_ROWS / 4) * GL_KSTRIDE;
    static_assert(2 * IMG >= WS, "the final combine reuses the LDS images");
    __shared__ double lds[2 * IMG];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // wave-uniform: scalar
    const int c = lane & 15, kq = lane >> 4;

    const int64_t rb0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t rb1 = rb0 + a.rows_per_block;
    if (rb1 > a.n) rb1 = a.n;

    // loader role: thread t -> column t >> 2 (of the first NC columns), rows (t & 3) * 8 .. + 7 of the chunk
    const int lcol = tid >> 2, lrg = tid & 3;
    const bool lvalid = lcol < a.n_cols;
    const bool lactive = lcol < NC;
    const int lsrc = lvalid ? a.gc.cols[lcol] : a.gc.cols[0];
    const T* lp = (const T*)a.base + (int64_t)lsrc * a.ld + (GATHER ? 0 : a.row0);
    const double lsh = lvalid ? a.shift[lsrc] : 0.0;
    double csum = 0.0;

    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0, 0, 0, 0};
    // this lane's operand element of k-step 2 * wave, column group 0 (doubles into an image)
    const int off0 = (2 * wave) * GL_KSTRIDE + (kq >> 1) * 32 + c * 2 + (kq & 1);

    T raw[8] = {};
    auto load_chunk = [&](int64_t r) {   // rows r + lrg * 8 .. + 7 of this thread's column into registers
        if (!lactive || a.debug_skip >= 2) return;
        const int64_t rl = r + lrg * 8;
        if (!GATHER && rl + 8 <= rb1) {   // straight into the prefetch registers: a copy would make the compiler wait for the loads here
            load_rows4<T>(lp + rl, reinterpret_cast<T(&)[4]>(raw[0]));
            load_rows4<T>(lp + rl + 4, reinterpret_cast<T(&)[4]>(raw[4]));
            return;
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const bool ok = rl + j < rb1;
            const int64_t src = ok ? (GATHER ? (int64_t)a.rows[rl + j] : rl + j) : (GATHER ? (int64_t)a.rows[rb0] : rb0);
            raw[j] = lp[src];
        }
    };
    auto store_chunk = [&](int64_t r, int buf) {   // shift, widen, column sums, LDS image
        if (!lactive || a.debug_skip == 3) return;
        const int64_t rl = r + lrg * 8;
        double x[8];
        double s = 0.0;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            x[j] = (lvalid && rl + j < rb1) ? (double)raw[j] - lsh : 0.0;
            s += x[j];
        }
        csum += s;
        // rows lrg * 8 + j: k-steps 2 lrg (j < 4) and 2 lrg + 1, k = j & 3; (k = 0, 1) and (k = 2, 3) are 16-byte pairs
        double* dst = &lds[buf * IMG + (2 * lrg) * GL_KSTRIDE + (lcol >> 4) * 64 + (lcol & 15) * 2];
        typedef double d2v __attribute__((ext_vector_type(2)));
        *(d2v*)(dst) = d2v{x[0], x[1]};
        *(d2v*)(dst + 32) = d2v{x[2], x[3]};
        *(d2v*)(dst + GL_KSTRIDE) = d2v{x[4], x[5]};
        *(d2v*)(dst + GL_KSTRIDE + 32) = d2v{x[6], x[7]};
    };

    int64_t r = rb0;
    int buf = 0;
    if (r < rb1) { load_chunk(r); store_chunk(r, 0); }
    __syncthreads();
    for (; r < rb1; r += GL_ROWS) {
        const bool more = r + GL_ROWS < rb1;
        if (more) load_chunk(r + GL_ROWS);             // in flight under the MFMAs below
        const double* img = lds + buf * IMG + off0;
        if (a.debug_skip != 1) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                double x[NCT];
#pragma unroll
                for (int I = 0; I < NCT; ++I) x[I] = img[s2 * GL_KSTRIDE + I * 64];
                int p = 0;
#pragma unroll
                for (int I = 0; I < NCT; ++I)
#pragma unroll
                    for (int J = I; J < NCT; ++J) {
                        acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[I], x[J], acc[p], 0, 0, 0);
                        ++p;
                    }
            }
        }
        if (more) store_chunk(r + GL_ROWS, buf ^ 1);
        __syncthreads();
        buf ^= 1;
    }

    // ---- block combine: one LDS image, the waves add into it in wave order 0, 1, 2, 3 (deterministic) -----------------
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int e = p * 256 + i * 64 + lane;
                    lds[e] = (w == 0) ? acc[p][i] : lds[e] + acc[p][i];
                }
            }
        }
        __syncthreads();
    }
    double* out = a.partial + (int64_t)blockIdx.x * WS;
    for (int e = threadIdx.x; e < NP * 256; e += 256) out[e] = lds[e];
    // column sums: the 4 loader threads of a column are adjacent lanes
    double s = csum;
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (lactive && lrg == 0) out[NP * 256 + lcol] = s;
}

// Sum the block partials (deterministic): stage 1 sums groups of 16 consecutive blocks in place into the first
// block of each group (grid.y = groups), stage 2 sums the group leaders in group order.
__global__ __launch_bounds__(256) void gram_reduce_kernel(double* __restrict__ partial, int nblocks, int WS, int stride,
                                                           double* __restrict__ out) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= WS) return;
    const int b0 = blockIdx.y * stride * 16;
    double v = 0.0;
    for (int i = 0; i < 16; ++i) {
        const int b = b0 + i * stride;
        if (b < nblocks) v += partial[(int64_t)b * WS + e];
    }
    if (out) out[e] = v; else partial[(int64_t)b0 * WS + e] = v;
}

static bool gram_uses_lds() {
    static const bool v = [] { const char* e = getenv("PBN_GRAM_LDS"); return !(e && *e) || atoi(e) != 0; }();
    return v;
}

template <typename T, bool GATHER>
static void launch_gram_t(const GramArgs& a, int nct, int nblocks, hipStream_t st) {
    dim3 grid(nblocks), block(256);
    if (gram_uses_lds()) {
        switch (nct) {
            case 1: hipLaunchKernelGGL((gram_lds_kernel<T, 1, GATHER>), grid, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL((gram_lds_kernel<T, 2, GATHER>), grid, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL((gram_lds_kernel<T, 3, GATHER>), grid, block, 0, st, a); break;
            case 4: hipLaunchKernelGGL((gram_lds_kernel<T, 4, GATHER>), grid, block, 0, st, a); break;
            default: throw invalid_error("gram: at most 64 columns per launch");
        }
        HIP_CHECK(hipGetLastError());
        return;
    }
    switch (nct) {
        case 1: hipLaunchKernelGGL((gram_kernel<T, 1, GATHER>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((gram_kernel<T, 2, GATHER>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((gram_kernel<T, 3, GATHER>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((gram_kernel<T, 4, GATHER>), grid, block, 0, st, a); break;
        default: throw invalid_error("gram: at most 64 columns per launch");
    }
    HIP_CHECK(hipGetLastError());
}

int gram_ws(int nct) { return nct * (nct + 1) / 2 * 256 + nct * 16; }

void launch_pilot(const void* base, int64_t ld, const GramCols& gc, int n_cols, int64_t row0, const int32_t* rows,
                  int64_t n, int dtype, double* shift, hipStream_t st) {
    if (dtype == PBN_F64)
        hipLaunchKernelGGL(pilot_mean_kernel<double>, dim3(n_cols), dim3(256), 0, st, base, ld, gc, n_cols, row0, rows, n, shift);
    else
        hipLaunchKernelGGL(pilot_mean_kernel<float>, dim3(n_cols), dim3(256), 0, st, base, ld, gc, n_cols, row0, rows, n, shift);
    HIP_CHECK(hipGetLastError());
}

void launch_gram(const GramArgs& a_in, int dtype, int nblocks, double* out, hipStream_t st) {
    GramArgs a = a_in;
    static const int dbg = [] { const char* e = getenv("PBN_GRAM_DEBUG"); return (e && *e) ? atoi(e) : 0; }();
    a.debug_skip = dbg;
    const int nct = (a.n_cols + 15) / 16;
    const int WS = gram_ws(nct);
    const bool gather = a.rows != nullptr;
    if (dtype == PBN_F64) {
        if (gather) launch_gram_t<double, true>(a, nct, nblocks, st); else launch_gram_t<double, false>(a, nct, nblocks, st);
    } else {
        if (gather) launch_gram_t<float, true>(a, nct, nblocks, st); else launch_gram_t<float, false>(a, nct, nblocks, st);
    }
    // tree of arity 16 over the blocks, fixed order
    int stride = 1;
    while ((nblocks + stride - 1) / stride > 16) {
        const int groups = (nblocks + stride * 16 - 1) / (stride * 16);
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((WS + 255) / 256, groups), dim3(256), 0, st, a.partial, nblocks, WS, stride, (double*)nullptr);
        stride *= 16;
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((WS + 255) / 256, 1), dim3(256), 0, st, a.partial, nblocks, WS, stride, out);
    HIP_CHECK(hipGetLastError());
}

// ---- LinearGaussianCPD logl / slogl (factors/continuous/LinearGaussianCPD.cpp:92-149), one streaming pass:
// N*(p+1)*sizeof(T) bytes, 2p+5 flops per row -> HBM-bound.  Arithmetic in double (the reference uses the
// data dtype); deterministic block tree sum for slogl.
template <typename T>
__global__ __launch_bounds__(256) void lg_logl_kernel(LgArgs a) {
    const int64_t r = (int64_t)blockIdx.x * 256 + threadIdx.x;
    double val = 0.0;
    if (r < a.n) {
        const T* base = (const T*)a.base;
        const int64_t src = a.rows ? (int64_t)a.rows[a.row0 + r] : a.row0 + r;
        double mean = a.beta[0];
        for (int j = 1; j <= a.p; ++j) mean += a.beta[j] * (double)base[(int64_t)a.gc.cols[j] * a.ld + src];
        const double y = (double)base[(int64_t)a.gc.cols[0] * a.ld + src];
        const double z = a.inv_std * (y - mean);
        val = -0.5 * z * z + a.cte;
        if (a.logl) a.logl[r] = a.want_cdf ? 0.5 * erfc(-z * 0.70710678118654752440) : val;
    }
    __shared__ double red[256];
    red[threadIdx.x] = val;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0 && a.block_sums) a.block_sums[blockIdx.x] = red[0];
}

void launch_lg_logl(const LgArgs& a, int dtype, hipStream_t st) {
    if (a.n == 0) return;
    dim3 grid((unsigned)ceil_div(a.n, 256)), block(256);
    if (dtype == PBN_F64)
        hipLaunchKernelGGL(lg_logl_kernel<double>, grid, block, 0, st, a);
    else
        hipLaunchKernelGGL(lg_logl_kernel<float>, grid, block, 0, st, a);
    HIP_CHECK(hipGetLastError());
}

// ---- row gather (arrow::compute::Take, dataset.hpp:2072-2075) -------------------------------------
template <typename T>
__global__ __launch_bounds__(256) void take_kernel(const T* __restrict__ src, int64_t ld_src, T* __restrict__ dst,
                                                   int64_t ld_dst, const int32_t* __restrict__ rows, int64_t n) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int64_t c = blockIdx.y;
    dst[c * ld_dst + i] = src[c * ld_src + rows[i]];
}

void launch_take(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* rows, int64_t n,
                 int n_cols, int dtype, hipStream_t st) {
    if (n == 0 || n_cols == 0) return;
    dim3 grid((unsigned)ceil_div(n, 256), (unsigned)n_cols), block(256);
    if (dtype == PBN_F64)
        hipLaunchKernelGGL(take_kernel<double>, grid, block, 0, st, (const double*)src, ld_src, (double*)dst, ld_dst, rows, n);
    else
        hipLaunchKernelGGL(take_kernel<float>, grid, block, 0, st, (const float*)src, ld_src, (float*)dst, ld_dst, rows, n);
    HIP_CHECK(hipGetLastError());
}

}  // namespace pbn
