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
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
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
// Wave priority for kernels that run ONE long-lived block per resident slot with the same work in each: a SIMD issues by
// priority, then AGE, so at equal priority the oldest wave's MFMAs go nearly unimpeded and the youngest gets the leftovers -
// measured with in-kernel time stamps on gram_lds_kernel (2M x 64, loads off): the four blocks of a CU finished after 88,
// 121, 155 and 178 us, the last stretch with one or two waves per SIMD, which cannot keep the matrix pipe busy.  A wave's
// priority therefore rotates with its OWN chunk count offset by its slot: whoever holds the highest runs ahead one chunk and
// drops to 0; no block leads by more than a few chunks and the blocks of a CU finish together (172-200 us after the change).
__device__ __forceinline__ int wave_slot() {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 4)" : "=s"(id));   // wave id within the SIMD = dispatch order
    return (int)id;
}
__device__ __forceinline__ void rotate_priority(int k) {
    switch (k & 3) {   // s_setprio takes an immediate
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}
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

    int64_t rb0 = (int64_t)blockIdx.x * a.rows_per_block;
    int64_t rb1 = rb0 + a.rows_per_block;
    if (rb1 > a.n) rb1 = a.n;
    if (a.blk) {   // a piece of one segment (launch_gram_segments)
        rb0 = a.blk[4 * blockIdx.x + 1]; rb1 = a.blk[4 * blockIdx.x + 2];
        if (rb1 <= rb0) return;   // a padding entry of an aligned launch order: no piece, no partial
    }

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
    int turn = wave_slot();
    for (; r < rb1; r += GL_ROWS) {
        rotate_priority(turn++);
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

    __builtin_amdgcn_s_setprio(0);
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
    double* out = a.partial + (int64_t)(a.blk ? a.blk[4 * blockIdx.x + 3] : blockIdx.x) * WS;
    for (int e = threadIdx.x; e < NP * 256; e += 256) out[e] = lds[e];
    // column sums: the 4 loader threads of a column are adjacent lanes
    double s = csum;
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    if (lactive && lrg == 0) out[NP * 256 + lcol] = s;
}

// ------------------------------------------------------------------------------------------------------------------
// gram_glds_kernel (double tables, contiguous rows): every WAVE streams its own rows through its own ring of LDS stages
// filled by the LDS-DMA (global_load_lds_dwordx4) - no barrier in the loop, no register spent on the prefetch.
// What the measurements on gram_lds_kernel said (2M x 64 fp64, profiles/r2/gram_floors.txt): (1) the shift / convert /
// ds_write_b128 phase costs a quarter of the MFMA side (206 -> 158 us without it; 16-byte LDS stores move their data at
// ~79 B/clk per CU, MI355X_MICROARCH.md "LDS"); (2) with one chunk in flight per block the loads alone take 185-215 us - the
// memory side is latency-bound, not bandwidth-bound; (3) the blocks of a CU finish one after the other (rotate_priority
// above); (4) an in-order wave that meets its VALU / scalar / branch work in one block between two MFMA bursts leaves the
// matrix pipe idle for it: 92 cycles per MFMA for a wave alone on its SIMD, 83 with a second wave, against 64.
// Here wave w of a block owns rows 16w .. 16w + 15 of each of the block's 64-row chunks: one 128-byte line of every column.
// A stage is 2 NCT DMA instructions of 64 lanes x 16 B (8 KiB at 64 columns); operand lane (c = lane & 15, kq = lane >> 4)
// reads row pairs kq and kq + 4 of column 16 I + c back with two ds_read_b128 per group: four k-steps per chunk, k-step
// 2h + e contracting rows 16w + 8h + 2kq + e - the contraction does not care which rows share a k-step as long as every
// column group agrees.  LDS is a register file extension: a ring of two stages per wave, filled two chunks ahead (a stage
// is refilled as soon as its chunk sits in registers, one chunk before it is multiplied), 16 KiB in flight per wave,
// 128 KiB per CU at two blocks - against 64 KiB with gram_lds_kernel's block-wide double buffer.  Own counted vmcnt is all
// the ordering a wave needs for its own DMA (MI355X_MICROARCH.md, "Two waves per SIMD", item 7).  The pilot shift is
// subtracted and the column sums are taken where the operands are read (4 NCT + 4 NCT DP instructions beside
// 2 NCT (NCT + 1) MFMAs per chunk); the steady state is one branch-free block with that work and the DMA instructions
// pinned between groups of NCT + 1 MFMAs.
// Chunk j of block b is chunk j gridDim.x + b of the range, so at any moment the resident blocks read one contiguous
// window of every column; only the range's very last chunk can be cut short, and its rows come through registers.
// Measured (tools/gram_bench.py, 2M x 64): 190-200 us per launch against 224-237 for gram_lds_kernel on the same boxes; MFMAs
// alone (PBN_GRAM_DEBUG=2) 165-172, DMA alone (=1) 172-185 = 5.5-5.9 TB/s.
// ------------------------------------------------------------------------------------------------------------------
typedef __attribute__((address_space(3))) void* lds_void_ptr;
typedef const __attribute__((address_space(1))) void* global_void_ptr;
typedef double d2v __attribute__((ext_vector_type(2)));

// One LDS-DMA instruction as inline asm (recipe: cdna_hip_programming.md section 5.7): hipcc does not count an asm memory
// operation, which is the point - with the builtin it drains the VMEM counter (vmcnt(0)) before the first use of any ds_read
// result while a DMA may be pending, and the ring would never have more than its own latency in flight.  The waits are counted
// by hand below (wait_vmcnt).  lds_dst: wave-uniform LDS byte address, the lanes land at lds_dst + 16 lane.
__device__ __forceinline__ void glds16(const void* gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
template <int N>
__device__ __forceinline__ void wait_vmcnt() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

constexpr int GD_STAGES = 2;          // ring depth per wave
constexpr int GD_BLOCKS_PER_CU = 2;   // 64 KiB of rings per block at 64 columns
constexpr int GD_ROWS = 64;           // rows per block chunk: 16 per wave = one 128-byte line of every column

template <int NCT>
struct TilePairs {   // pair p -> (I, J), I <= J, in the order the accumulators are numbered
    int I[NPairs<NCT>::value], J[NPairs<NCT>::value];
    constexpr TilePairs() : I{}, J{} {
        int p = 0;
        for (int i = 0; i < NCT; ++i)
            for (int j = i; j < NCT; ++j) { I[p] = i; J[p] = j; ++p; }
    }
};

// DBG (PBN_GRAM_DEBUG, measurement only): 1 = no MFMAs, 2 = no DMA (the statistics are garbage then)
template <int NCT, int DBG>
__global__ __launch_bounds__(256, GD_BLOCKS_PER_CU) void gram_glds_kernel(GramArgs a) {
    constexpr int NP = NPairs<NCT>::value;
    constexpr int NC = NCT * 16;
    constexpr int WS = NP * 256 + NCT * 16;
    constexpr int NQ = 2 * NCT;                 // DMA instructions (= 16-byte reads per lane) per stage
    constexpr int STAGE = NQ * 128;             // doubles per stage: NQ instructions x 64 lanes x 16 B
    constexpr int RING = GD_STAGES * STAGE;     // doubles per wave
    constexpr int COMBINE = NP * 256 + 4 * NCT * 64;
    constexpr int LDS_DOUBLES = 4 * RING > COMBINE ? 4 * RING : COMBINE;   // the rings, then the block combine
    constexpr TilePairs<NCT> PAIRS{};
    __shared__ double lds[LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int slot = wave_slot();
    const long long t_start = a.stamps ? wall_clock64() : 0;

    // chunk i of the block: rows first + i stride .. + GD_ROWS - 1 of the range, cut at `end`.  Plain form: the block's chunks are
    // interleaved with the other blocks' (see above); segmented form (a.blk, launch_gram_segments): the block owns the contiguous
    // piece [blk[4b + 1], blk[4b + 2]) and writes its partial to slot blk[4b + 3]
    const int64_t b = blockIdx.x;
    int64_t first, stride, end, mine, out_slot = b;
    if (a.blk) {
        first = a.blk[4 * b + 1]; end = a.blk[4 * b + 2]; stride = GD_ROWS; out_slot = a.blk[4 * b + 3];
        if (end <= first) return;   // a padding entry of an aligned launch order: no piece, no partial
        mine = (end - first + GD_ROWS - 1) / GD_ROWS;
    } else {
        const int64_t nchunks = (a.n + GD_ROWS - 1) / GD_ROWS, B = gridDim.x;
        first = b * GD_ROWS; end = a.n; stride = B * GD_ROWS;
        mine = b < nchunks ? (nchunks - 1 - b) / B + 1 : 0;
    }
    const bool partial_last = mine > 0 && first + (mine - 1) * stride + GD_ROWS > end;   // the range's last chunk, cut short
    const int64_t nfull = mine - (partial_last ? 1 : 0);   // whole chunks (LDS-DMA); the cut one goes through registers

    // operand role: lane (c, kq) holds column 16 I + c; shift, column sums
    double sh[NCT], cs[NCT];
    bool cvalid[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) {
        cvalid[I] = 16 * I + c < a.n_cols;
        sh[I] = a.shift[cvalid[I] ? a.gc.cols[16 * I + c] : a.gc.cols[0]];
        cs[I] = 0.0;
    }
    // DMA role: instruction q moves whole 128-byte lines - 16 rows of columns 8q .. 8q + 7, eight ADJACENT lanes per line
    // (the texture addresser merges neighbouring lanes; with the operand layout's lanes c, c + 16, c + 32, c + 48 on one
    // segment every lane was a request of its own).  Lane l fetches row pair m = (l & 7) ^ r, r = (column & 15) >> 1, of
    // column 8q + (l >> 3); operand lane (c, kq) finds row pair m of its column in slot 8 (c & 7) + (m ^ (c >> 1)) of
    // instruction 2I + (c >> 3) - the XOR spreads the 16 lanes of every ds_read_b128 group over all 64 banks.
    const double* dsrc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int cc = 8 * q + (lane >> 3);
        const int col = cc < a.n_cols ? a.gc.cols[cc] : a.gc.cols[0];
        const int m = (lane & 7) ^ ((cc & 15) >> 1);
        dsrc[q] = (const double*)a.base + (int64_t)col * a.ld + a.row0 + first + 16 * wave + 2 * m;
    }
    // doubles from a stage's start to this lane's row pairs kq (rows 2kq, +1) and kq + 4 (rows 8 + 2kq, +1) of group 0
    const int rd0 = (c >> 3) * 128 + ((c & 7) * 8 + (kq ^ (c >> 1))) * 2, rd1 = (c >> 3) * 128 + ((c & 7) * 8 + ((kq + 4) ^ (c >> 1))) * 2;
    // the shifts must be IN before the first DMA: an empty asm that reads them makes hipcc place its own wait here - left to
    // itself it waits at their first use inside the loop, on every trip, and its vmcnt(0) there drains the DMA ring
#pragma unroll
    for (int I = 0; I < NCT; ++I) asm volatile("" : "+v"(sh[I]));
    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0, 0, 0, 0};

    double* ring = lds + wave * RING;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_ptr)ring);   // LDS byte address
    auto dma1 = [&](int64_t i, int st, int q) {   // instruction q of chunk i of the block into stage st
        if (DBG >= 2) return;
        glds16(dsrc[q] + i * stride, ring_addr + (st * STAGE + q * 128) * 8);
    };
    auto read_stage = [&](int st, d2v (&x2)[NQ]) {   // x2[2I + h]: rows 8h + 2kq, + 1 of column 16 I + c
#pragma unroll
        for (int I = 0; I < NCT; ++I) {
            x2[2 * I] = *(const d2v*)(ring + st * STAGE + I * 256 + rd0);
            x2[2 * I + 1] = *(const d2v*)(ring + st * STAGE + I * 256 + rd1);
        }
    };
    // operands of read q of a stage (k-steps 2 (q & 1), + 1 of group q >> 1): subtract the shift, zero the columns past n_cols
    // (only the last group can hold any), column sums.  k-steps: rows 16w + {0,2,4,6}, {1,3,5,7}, {8,..,14}, {9,..,15}
    auto prepare1 = [&](const d2v (&x2)[NQ], double (&x)[4][NCT], int q) {
        const int I = q >> 1, h = q & 1;
        double lo = x2[q][0] - sh[I], hi = x2[q][1] - sh[I];
        if (I == NCT - 1) {
            lo = cvalid[I] ? lo : 0.0;
            hi = cvalid[I] ? hi : 0.0;
        }
        x[2 * h][I] = lo;
        x[2 * h + 1][I] = hi;
        cs[I] += lo + hi;
    };
    auto mfma1 = [&](const double (&x)[4][NCT], int m) {   // MFMA m of a stage's 4 NP: k-step m / NP, tile pair m % NP
        if (DBG == 1) return;
        const int k = m / NP, p = m % NP;
        acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[k][PAIRS.I[p]], x[k][PAIRS.J[p]], acc[p], 0, 0, 0);
    };
    // One chunk of the steady state, branch-free: chunk i is multiplied out of registers (xc) while chunk i + 1 is read from its
    // stage st, the stage is refilled with chunk i + 3 (chunk i + 2 is in flight in the other stage) and chunk i + 1's
    // operands (xn) are prepared.  The 4 NP MFMAs are cut into NQ groups of NCT + 1, each followed by one DMA instruction and
    // the preparation of one 16-byte read - pinned in that order (sched_barrier), so that everything but the DP
    // additions issues while an MFMA runs: an in-order wave that meets its VALU / scalar / branch work in one block between two
    // MFMA bursts leaves the matrix pipe idle for it (measured: 92 cycles per MFMA alone on a SIMD, 83 with a second wave).
    auto steady = [&](int64_t i, int st, const double (&xc)[4][NCT], double (&xn)[4][NCT]) {
        wait_vmcnt<NQ>();   // chunk i + 1 has landed; chunk i + 2 may stay in flight
        d2v x2[NQ];
        read_stage(st, x2);
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
#pragma unroll
            for (int e = 0; e < NCT + 1; ++e) mfma1(xc, q * (NCT + 1) + e);
            __builtin_amdgcn_sched_barrier(0);
            if (q == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // chunk i + 1 is in registers: its stage can be refilled
            dma1(i + 1 + GD_STAGES, st, q);
            prepare1(x2, xn, q);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    double xa[4][NCT], xb[4][NCT];
    for (int i = 0; i < GD_STAGES && i < nfull; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) dma1(i, i, q);
    if (nfull > 0) {
        if (nfull > 1) wait_vmcnt<NQ>(); else wait_vmcnt<0>();
        d2v x2[NQ];
        read_stage(0, x2);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (GD_STAGES < nfull)
#pragma unroll
            for (int q = 0; q < NQ; ++q) dma1(GD_STAGES, 0, q);
#pragma unroll
        for (int q = 0; q < NQ; ++q) prepare1(x2, xa, q);
    }
    int64_t i = 0;
    for (; i + 4 < nfull; i += 2) {   // both chunks of a trip have a chunk i + 3 to prefetch
        rotate_priority((int)(i >> 1) + slot);
        steady(i, 1, xa, xb);
        steady(i + 1, 0, xb, xa);
    }
    for (; i < nfull; ++i) {   // the last chunks: the same steps with their conditions
        const int st = (int)(i + 1) & 1;   // stage of chunk i + 1
        const bool more = i + 1 < nfull;
        d2v x2[NQ];
        if (more) {
            if (i + 2 < nfull) wait_vmcnt<NQ>(); else wait_vmcnt<0>();
            read_stage(st, x2);
        }
#pragma unroll
        for (int m = 0; m < 4 * NP; ++m) mfma1(xa, m);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + 1 + GD_STAGES < nfull)
#pragma unroll
                for (int q = 0; q < NQ; ++q) dma1(i + 1 + GD_STAGES, st, q);
#pragma unroll
            for (int q = 0; q < NQ; ++q) prepare1(x2, xb, q);
#pragma unroll
            for (int k = 0; k < 4; ++k)
#pragma unroll
                for (int I = 0; I < NCT; ++I) xa[k][I] = xb[k][I];
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (partial_last) {   // rows past the range read as the shift: x - shift = 0 exactly
        const int64_t r = first + nfull * stride + 16 * wave + 2 * kq, left = end - r;   // this lane's first row; rows from it on
        d2v x2[NQ];
#pragma unroll
        for (int I = 0; I < NCT; ++I) {
            const double* p = (const double*)a.base + (int64_t)(cvalid[I] ? a.gc.cols[16 * I + c] : a.gc.cols[0]) * a.ld + a.row0 + r;
            x2[2 * I][0] = left > 0 ? p[0] : sh[I];
            x2[2 * I][1] = left > 1 ? p[1] : sh[I];
            x2[2 * I + 1][0] = left > 8 ? p[8] : sh[I];
            x2[2 * I + 1][1] = left > 9 ? p[9] : sh[I];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) prepare1(x2, xa, q);
#pragma unroll
        for (int m = 0; m < 4 * NP; ++m) mfma1(xa, m);
    }
    __syncthreads();   // every wave is done with its ring

    // ---- block combine: the waves add their tiles in wave order; the column sums go through LDS per (wave, group, lane) -----
    double* lcs = lds + NP * 256;
#pragma unroll
    for (int I = 0; I < NCT; ++I) lcs[(wave * NCT + I) * 64 + lane] = cs[I];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = p * 256 + e4 * 64 + lane;
                    lds[e] = (w == 0) ? acc[p][e4] : lds[e] + acc[p][e4];
                }
            }
        }
        __syncthreads();
    }
    double* out = a.partial + out_slot * WS;
    for (int e = threadIdx.x; e < NP * 256; e += 256) out[e] = lds[e];
    if (tid < NC) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int k = 0; k < 4; ++k) s += lcs[(w * NCT + (tid >> 4)) * 64 + k * 16 + (tid & 15)];
        out[NP * 256 + tid] = s;
    }
    if (a.stamps && tid == 0) {   // PBN_GRAM_STAMPS: start, end (10 ns ticks) and hardware slot of every block
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        a.stamps[blockIdx.x * 3] = t_start; a.stamps[blockIdx.x * 3 + 1] = wall_clock64(); a.stamps[blockIdx.x * 3 + 2] = hwid;
    }
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

// ------------------------------------------------------------------------------------------------------------------
// gram_glds_f32_kernel: the same ring for FLOAT tables.  16 bytes are four rows, so wave w owns rows 32w .. 32w + 31 of each
// 128-row block chunk (again one 128-byte line of every column), a stage is 2 NCT DMA instructions = eight k-steps; operand lane
// (c, kq) reads row quads kq and kq + 4 of column 16 I + c (k-step 4h + e contracts rows 32w + 16h + 4kq + e).  The floats stay
// raw in registers - this chunk's and the next one's - and are widened and shifted k-step by k-step, right before the k-step's
// MFMAs (NCT conversions, subtractions and column-sum additions beside NCT (NCT + 1) / 2 MFMAs), with one DMA instruction pinned
// behind each of the first 2 NCT k-steps.  Half the bytes of the double table: the MFMA side is what bounds it.
// ------------------------------------------------------------------------------------------------------------------
typedef float f4v __attribute__((ext_vector_type(4)));
constexpr int GF_ROWS = 128;   // rows per block chunk of the float ring: 32 per wave

template <int NCT, int DBG>
__global__ __launch_bounds__(256, GD_BLOCKS_PER_CU) void gram_glds_f32_kernel(GramArgs a) {
    constexpr int NP = NPairs<NCT>::value;
    constexpr int NC = NCT * 16;
    constexpr int WS = NP * 256 + NCT * 16;
    constexpr int NQ = 2 * NCT;                 // DMA instructions (= 16-byte reads per lane) per stage
    constexpr int STAGE = NQ * 256;             // floats per stage
    constexpr int RING = GD_STAGES * STAGE;     // floats per wave
    constexpr int COMBINE = NP * 256 + 4 * NCT * 64;   // doubles
    constexpr int LDS_DOUBLES = 4 * RING / 2 > COMBINE ? 4 * RING / 2 : COMBINE;
    constexpr TilePairs<NCT> PAIRS{};
    __shared__ double lds[LDS_DOUBLES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;
    const int slot = wave_slot();
    const long long t_start = a.stamps ? wall_clock64() : 0;

    const int64_t b = blockIdx.x;   // plain / segmented form: see gram_glds_kernel
    int64_t first, stride, end, mine, out_slot = b;
    if (a.blk) {
        first = a.blk[4 * b + 1]; end = a.blk[4 * b + 2]; stride = GF_ROWS; out_slot = a.blk[4 * b + 3];
        if (end <= first) return;
        mine = (end - first + GF_ROWS - 1) / GF_ROWS;
    } else {
        const int64_t nchunks = (a.n + GF_ROWS - 1) / GF_ROWS, B = gridDim.x;
        first = b * GF_ROWS; end = a.n; stride = B * GF_ROWS;
        mine = b < nchunks ? (nchunks - 1 - b) / B + 1 : 0;
    }
    const bool partial_last = mine > 0 && first + (mine - 1) * stride + GF_ROWS > end;
    const int64_t nfull = mine - (partial_last ? 1 : 0);

    double sh[NCT], cs[NCT];
    bool cvalid[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) {
        cvalid[I] = 16 * I + c < a.n_cols;
        sh[I] = a.shift[cvalid[I] ? a.gc.cols[16 * I + c] : a.gc.cols[0]];
        cs[I] = 0.0;
    }
    // DMA role (see gram_glds_kernel): instruction q moves 32 rows of columns 8q .. 8q + 7, eight adjacent lanes per line; lane l
    // fetches row quad m = (l & 7) ^ r, r = (column & 15) >> 1
    const float* dsrc[NQ];
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int cc = 8 * q + (lane >> 3);
        const int col = cc < a.n_cols ? a.gc.cols[cc] : a.gc.cols[0];
        const int m = (lane & 7) ^ ((cc & 15) >> 1);
        dsrc[q] = (const float*)a.base + (int64_t)col * a.ld + a.row0 + first + 32 * wave + 4 * m;
    }
    // floats from a stage's start to this lane's row quads kq and kq + 4 of group 0
    const int rd0 = (c >> 3) * 256 + ((c & 7) * 8 + (kq ^ (c >> 1))) * 4, rd1 = (c >> 3) * 256 + ((c & 7) * 8 + ((kq + 4) ^ (c >> 1))) * 4;
#pragma unroll
    for (int I = 0; I < NCT; ++I) asm volatile("" : "+v"(sh[I]));   // the shifts are in before the first DMA (see gram_glds_kernel)
    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0, 0, 0, 0};

    float* ring = (float*)lds + wave * RING;
    const unsigned ring_addr = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(lds_void_ptr)ring);
    auto dma1 = [&](int64_t i, int st, int q) {
        if (DBG >= 2) return;
        glds16(dsrc[q] + i * stride, ring_addr + (st * STAGE + q * 256) * 4);
    };
    auto read_stage = [&](int st, f4v (&raw)[NQ]) {   // raw[2I + h][e]: row 16h + 4kq + e of column 16 I + c
#pragma unroll
        for (int I = 0; I < NCT; ++I) {
            raw[2 * I] = *(const f4v*)(ring + st * STAGE + I * 512 + rd0);
            raw[2 * I + 1] = *(const f4v*)(ring + st * STAGE + I * 512 + rd1);
        }
    };
    // k-step k = 4h + e of a chunk: widen, shift, zero the columns past n_cols, column sums, the tile pairs' MFMAs
    auto kstep = [&](const f4v (&raw)[NQ], int k) {
        const int h = k >> 2, e = k & 3;
        double x[NCT];
#pragma unroll
        for (int I = 0; I < NCT; ++I) {
            x[I] = (double)raw[2 * I + h][e] - sh[I];
            if (I == NCT - 1) x[I] = cvalid[I] ? x[I] : 0.0;
            cs[I] += x[I];
        }
        if (DBG == 1) return;
#pragma unroll
        for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[PAIRS.I[p]], x[PAIRS.J[p]], acc[p], 0, 0, 0);
    };
    auto steady = [&](int64_t i, int st, const f4v (&rc)[NQ], f4v (&rn)[NQ]) {
        wait_vmcnt<NQ>();   // chunk i + 1 has landed; chunk i + 2 may stay in flight
        read_stage(st, rn);
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            kstep(rc, k);
            __builtin_amdgcn_sched_barrier(0);
            if (k == 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");   // chunk i + 1 is in registers: its stage can be refilled
            if (k < NQ) dma1(i + 1 + GD_STAGES, st, k);
            __builtin_amdgcn_sched_barrier(0);
        }
    };

    f4v ra[NQ], rb[NQ];
    for (int i = 0; i < GD_STAGES && i < nfull; ++i)
#pragma unroll
        for (int q = 0; q < NQ; ++q) dma1(i, i, q);
    if (nfull > 0) {
        if (nfull > 1) wait_vmcnt<NQ>(); else wait_vmcnt<0>();
        read_stage(0, ra);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (GD_STAGES < nfull)
#pragma unroll
            for (int q = 0; q < NQ; ++q) dma1(GD_STAGES, 0, q);
    }
    int64_t i = 0;
    for (; i + 4 < nfull; i += 2) {
        rotate_priority((int)(i >> 1) + slot);
        steady(i, 1, ra, rb);
        steady(i + 1, 0, rb, ra);
    }
    for (; i < nfull; ++i) {   // the last chunks: the same steps with their conditions
        const int st = (int)(i + 1) & 1;
        const bool more = i + 1 < nfull;
        if (more) {
            if (i + 2 < nfull) wait_vmcnt<NQ>(); else wait_vmcnt<0>();
            read_stage(st, rb);
        }
#pragma unroll
        for (int k = 0; k < 8; ++k) kstep(ra, k);
        if (more) {
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (i + 1 + GD_STAGES < nfull)
#pragma unroll
                for (int q = 0; q < NQ; ++q) dma1(i + 1 + GD_STAGES, st, q);
#pragma unroll
            for (int q = 0; q < NQ; ++q) ra[q] = rb[q];
        }
    }
    __builtin_amdgcn_s_setprio(0);
    if (partial_last) {   // the cut chunk through registers: rows past the range contribute nothing
        const int64_t r = first + nfull * stride + 32 * wave + 4 * kq, left = end - r;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const int h = k >> 2, e = k & 3;
            const bool in = left > 16 * h + e;
            double x[NCT];
#pragma unroll
            for (int I = 0; I < NCT; ++I) {
                const float* p = (const float*)a.base + (int64_t)(cvalid[I] ? a.gc.cols[16 * I + c] : a.gc.cols[0]) * a.ld + a.row0 + r;
                x[I] = (in && cvalid[I]) ? (double)p[16 * h + e] - sh[I] : 0.0;
                cs[I] += x[I];
            }
            if (DBG != 1) {
#pragma unroll
                for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[PAIRS.I[p]], x[PAIRS.J[p]], acc[p], 0, 0, 0);
            }
        }
    }
    __syncthreads();   // every wave is done with its ring

    // ---- block combine, as in gram_glds_kernel -----
    double* lcs = lds + NP * 256;
#pragma unroll
    for (int I = 0; I < NCT; ++I) lcs[(wave * NCT + I) * 64 + lane] = cs[I];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = p * 256 + e4 * 64 + lane;
                    lds[e] = (w == 0) ? acc[p][e4] : lds[e] + acc[p][e4];
                }
            }
        }
        __syncthreads();
    }
    double* out = a.partial + out_slot * WS;
    for (int e = threadIdx.x; e < NP * 256; e += 256) out[e] = lds[e];
    if (tid < NC) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int k = 0; k < 4; ++k) s += lcs[(w * NCT + (tid >> 4)) * 64 + k * 16 + (tid & 15)];
        out[NP * 256 + tid] = s;
    }
    if (a.stamps && tid == 0) {
        unsigned hwid;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hwid));
        a.stamps[blockIdx.x * 3] = t_start; a.stamps[blockIdx.x * 3 + 1] = wall_clock64(); a.stamps[blockIdx.x * 3 + 2] = hwid;
    }
}

// ------------------------------------------------------------------------------------------------------------------
// gram_gring_kernel: the Gram over a ROW LIST (a.rows: null-free rows of a candidate's columns, the rows of a discrete configuration,
// a MutualInformation grouping's permutation).  A gathered element is 8 (4) bytes at its own address: the LDS-DMA moves 4 or 16
// bytes per lane, so the ring of gram_glds_kernel is kept in REGISTERS instead - the operand layout itself is the load layout.
// Wave w of a block owns list positions 16w .. 16w + 15 of each 64-position chunk; lane (c, kq) reads the four rows at positions
// 4kq .. 4kq + 3 for column 16 I + c (k-step j contracts the rows at positions j, 4 + j, 8 + j, 12 + j: the contraction does not
// care which rows share a k-step), straight into the MFMA operands: no LDS, no barrier, no transposition.  Three chunks of
// operands rotate (multiplied / landed or landing / in flight, 4 NCT loads per lane and chunk each) and the row indices run one
// chunk further ahead (one 16-byte read per lane and chunk); the compiler counts the waits (every load is its own).  The lanes of
// a load instruction touch 16 columns x 4 rows: with a dense, increasing list the 4 rows and the chunk's other 3 loads of the
// column group share a 128-byte line, with a sparse one every element is its own 64-byte sector and the pass is bound by the
// sectors it drags in.
// gram_lds_kernel<T, NCT, true> - a block-wide 32-row LDS image, one chunk of 8 single-element loads per thread in flight, the
// row index read again by each of the 64 column threads - took 637 us for a 4-category grouping of the 2M x 64 table.
// ------------------------------------------------------------------------------------------------------------------
template <int N> struct IdxVec;
template <> struct IdxVec<4> { typedef int type __attribute__((ext_vector_type(4), aligned(4))); };
template <> struct IdxVec<2> { typedef int type __attribute__((ext_vector_type(2), aligned(4))); };

template <typename T, int N> struct RowVec { typedef T type __attribute__((ext_vector_type(N), aligned(sizeof(T)))); };
template <typename T> struct RowVec<T, 1> { typedef T type; };

// RM: the rows come from a ROW-MAJOR mirror of the launch's columns (GramArgs::rowmajor, build_rowmajor_mirror) in which the NCT
// columns of a lane are adjacent: a gathered row is 16 NCT contiguous elements (512 bytes at 64 double columns), read by the 16
// lanes of a row quarter with one 8 NCT-byte load each - whole lines whatever the list looks like, where the column-major table
// gives every element of a sparse list a 64-byte sector of its own (2M x 64 doubles, rows of 4 / 64 configurations: 565 / 1210 us
// through the columns - the first figure only with the pieces launched stripe-major).
template <typename T, int NCT, bool RM>
__global__ __launch_bounds__(256, 2) void gram_gring_kernel(GramArgs a) {
    constexpr int NP = NPairs<NCT>::value;
    constexpr int NC = NCT * 16;
    constexpr int WS = NP * 256 + NCT * 16;
    // a stage holds RL rows per lane (= RL k-steps, 16 RL list positions per block step); 64 double columns: stages of two rows and a
    // ring of four, so that three are in flight beside the 80 accumulator registers (12 KiB per wave); otherwise four rows, ring of three
    constexpr int RL = (sizeof(T) == 8 && NCT == 4) ? 2 : 4;   // (RM with four rows and a ring of three: the same 199 us)
    constexpr int D = RL == 2 ? 4 : 3;
    constexpr int STEP = 16 * RL;   // list positions per block step
    typedef typename IdxVec<RL>::type idx_t;
    constexpr TilePairs<NCT> PAIRS{};
    __shared__ double lds[NP * 256 + 4 * NCT * 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 15, kq = lane >> 4;

    int64_t rb0, rb1, out_slot = blockIdx.x;
    if (a.blk) {
        rb0 = a.blk[4 * blockIdx.x + 1]; rb1 = a.blk[4 * blockIdx.x + 2]; out_slot = a.blk[4 * blockIdx.x + 3];
        if (rb1 <= rb0) return;   // a padding block of an XCD-aligned launch order: no piece, no partial
    } else {
        rb0 = (int64_t)blockIdx.x * a.rows_per_block;
        rb1 = rb0 + a.rows_per_block;
        if (rb1 > a.n) rb1 = a.n;
    }
    const int64_t nsteps = rb1 > rb0 ? (rb1 - rb0 + STEP - 1) / STEP : 0;

    const T* colp[NCT];
    double sh[NCT], cs[NCT];
    bool cvalid[NCT];
#pragma unroll
    for (int I = 0; I < NCT; ++I) {
        cvalid[I] = 16 * I + c < a.n_cols;
        const int src = cvalid[I] ? a.gc.cols[16 * I + c] : a.gc.cols[0];
        colp[I] = (const T*)a.base + (int64_t)src * a.ld;
        sh[I] = a.shift[src];
        cs[I] = 0.0;
    }
    d4 acc[NP];
#pragma unroll
    for (int p = 0; p < NP; ++p) acc[p] = d4{0, 0, 0, 0};

    const int32_t* rows = a.rows;
    const int64_t p0 = rb0 + 4 * RL * wave + RL * kq;   // this lane's first position of step 0
    // row indices of step t; `cut`: positions past the piece read as the piece's first row (their operands are zeroed in `mult`)
    auto load_idx = [&](int64_t t, idx_t& idx, bool cut) {
        const int64_t p = p0 + STEP * t;
        if (!cut) { idx = *(const idx_t*)(rows + p); return; }
#pragma unroll
        for (int j = 0; j < RL; ++j) idx[j] = rows[p + j < rb1 ? p + j : rb0];
    };
    // element address = column + row * sizeof(T) as ONE v_mad_u64_u32 on the 32-bit row (an opaque multiplier, or the compiler
    // widens the row into a register pair first: the pairs' copies end up behind the loads at the loop's end, with a vmcnt(0))
    unsigned esize = RM ? sizeof(T) * NC : sizeof(T);   // bytes from a row to the next
    asm volatile("" : "+s"(esize));
    typedef typename RowVec<T, NCT>::type rowvec_t;
    const uint64_t rmp = (uint64_t)(uintptr_t)a.rowmajor + (uint64_t)(NCT * c) * sizeof(T);   // RM: this lane's NCT columns of row 0
    auto load_data = [&](const idx_t& idx, T (&raw)[NCT][RL]) {
        if constexpr (RM) {
#pragma unroll
            for (int j = 0; j < RL; ++j) {
                const uint64_t row = rmp + (uint64_t)(uint32_t)idx[j] * (uint64_t)esize;
                if constexpr (NCT == 3) {
                    // two elements and one: a 3-vector is read as four, and the compiler gives the fourth's registers to live
                    // values - every one of them then waits for the load to land before it may be written
                    typedef typename RowVec<T, 2>::type pair_t;
                    const pair_t v = *(const __attribute__((address_space(1))) pair_t*)row;
                    raw[0][j] = v[0]; raw[1][j] = v[1];
                    raw[2][j] = *(const __attribute__((address_space(1))) T*)(row + 2 * sizeof(T));
                } else {
                    const rowvec_t v = *(const __attribute__((address_space(1))) rowvec_t*)row;
#pragma unroll
                    for (int I = 0; I < NCT; ++I) {
                        if constexpr (NCT == 1) raw[I][j] = v; else raw[I][j] = v[I];
                    }
                }
            }
        } else {
#pragma unroll
            for (int I = 0; I < NCT; ++I)
#pragma unroll
                for (int j = 0; j < RL; ++j)
                    raw[I][j] = *(const __attribute__((address_space(1))) T*)((uint64_t)(uintptr_t)colp[I] + (uint64_t)(uint32_t)idx[j] * (uint64_t)esize);   // global_load: a flat one waits for everything
        }
    };
    auto mult = [&](int64_t t, const T (&raw)[NCT][RL], bool cut) {
        const int64_t left = cut ? rb1 - (p0 + STEP * t) : RL;   // rows of the piece from this lane's first position on
#pragma unroll
        for (int j = 0; j < RL; ++j) {
            double x[NCT];
#pragma unroll
            for (int I = 0; I < NCT; ++I) {
                x[I] = (double)raw[I][j] - sh[I];
                if (I == NCT - 1) x[I] = cvalid[I] ? x[I] : 0.0;
                if (cut) x[I] = left > j ? x[I] : 0.0;
                cs[I] += x[I];
            }
#pragma unroll
            for (int p = 0; p < NP; ++p) acc[p] = __builtin_amdgcn_mfma_f64_16x16x4f64(x[PAIRS.I[p]], x[PAIRS.J[p]], acc[p], 0, 0, 0);
        }
    };

    // The ring.  Before step t: the data of steps t .. t + D - 2 are in flight or landed in stages (t .. t + D - 2) mod D, the indices
    // of steps t + D - 1 .. t + 2 D - 3 in their slots (step mod D).  Step t issues the data of step t + D - 1, asks for the indices
    // of step t + 2 D - 2 and multiplies stage t mod D.  Loads return in order: the indices a step needs were asked for D - 1 steps
    // ago, right behind the data of the step it multiplies - waiting for them leaves the D - 2 younger stages in flight (one step
    // ahead only, the wait for the indices would drain every stage each step).  Trips of D steps keep every stage and slot a
    // compile-time name; the steady trips run without a condition (the compiler counts its waits over straight-line code: a
    // conditional load would make it assume the shorter queue and wait for more than it needs), the last steps repeat the trip
    // with theirs.
    T r[D][NCT][RL];
    idx_t ix[D];
    int64_t t = 0;
    if (nsteps >= 3 * D) {
        // a long piece: the ring is filled without a condition and in this order - the waits of the loop's first trip are counted
        // from here too, and a load that may not have been issued (or an index load sunk behind the data) would make the loop wait
        // for nearly everything at its top, every trip
        idx_t first[D - 1];
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_idx(j, first[j], false);
#pragma unroll
        for (int j = D - 1; j <= 2 * D - 3; ++j) load_idx(j, ix[j % D], false);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_data(first[j], r[j]);
        __builtin_amdgcn_sched_barrier(0);
        for (; t + 3 * D - 1 < nsteps; t += D) {   // every step the trip issues or asks indices for is a whole one
#pragma unroll
            for (int s = 0; s < D; ++s) {
                load_data(ix[(s + D - 1) % D], r[(s + D - 1) % D]);
                load_idx(t + s + 2 * D - 2, ix[(s + 2 * D - 2) % D], false);
                __builtin_amdgcn_sched_barrier(0);   // the loads of a step stay in their step: hoisted, they would lengthen the stages' lives
                mult(t + s, r[s], false);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    } else {
        idx_t first[D - 1];
#pragma unroll
        for (int j = 0; j < D - 1; ++j) load_idx(j, first[j], true);
#pragma unroll
        for (int j = D - 1; j <= 2 * D - 3; ++j) load_idx(j, ix[j % D], true);
#pragma unroll
        for (int j = 0; j < D - 1; ++j)
            if (j < nsteps) load_data(first[j], r[j]);
    }
    for (; t < nsteps; t += D) {
#pragma unroll
        for (int s = 0; s < D; ++s) {
            if (t + s + D - 1 < nsteps) load_data(ix[(s + D - 1) % D], r[(s + D - 1) % D]);
            if (t + s + 2 * D - 2 < nsteps) load_idx(t + s + 2 * D - 2, ix[(s + 2 * D - 2) % D], true);
            __builtin_amdgcn_sched_barrier(0);
            if (t + s < nsteps) mult(t + s, r[s], true);
            __builtin_amdgcn_sched_barrier(0);
        }
    }

    // ---- block combine, as in gram_glds_kernel: tiles in wave order, column sums per (wave, group, lane) ------------------
    double* lcs = lds + NP * 256;
#pragma unroll
    for (int I = 0; I < NCT; ++I) lcs[(wave * NCT + I) * 64 + lane] = cs[I];
    for (int w = 0; w < 4; ++w) {
        if (wave == w) {
#pragma unroll
            for (int p = 0; p < NP; ++p) {
#pragma unroll
                for (int e4 = 0; e4 < 4; ++e4) {
                    const int e = p * 256 + e4 * 64 + lane;
                    lds[e] = (w == 0) ? acc[p][e4] : lds[e] + acc[p][e4];
                }
            }
        }
        __syncthreads();
    }
    double* out = a.partial + out_slot * WS;
    for (int e = threadIdx.x; e < NP * 256; e += 256) out[e] = lds[e];
    if (tid < NC) {
        double s = 0.0;
        for (int w = 0; w < 4; ++w)
            for (int k = 0; k < 4; ++k) s += lcs[(w * NCT + (tid >> 4)) * 64 + k * 16 + (tid & 15)];
        out[NP * 256 + tid] = s;
    }
}

// four threads per (segment, element): the segment's partial slots blk_off[g] .. blk_off[g + 1] - 1 are cut into four runs of
// ceil(count / 4), each added in slot order, and the four run sums are added in run order - a fixed association that depends on the
// segment's number of pieces only
__global__ __launch_bounds__(256) void gram_seg_reduce_kernel(const double* __restrict__ partial, const int32_t* __restrict__ blk_off, int WS,
                                                               double* __restrict__ out) {
    __shared__ double run[4][64];
    const int e = blockIdx.x * 64 + threadIdx.x, g = blockIdx.y, q = threadIdx.y;
    const int b0 = blk_off[g], b1 = blk_off[g + 1], len = (b1 - b0 + 3) / 4;
    double v = 0.0;
    if (e < WS)
        for (int b = b0 + q * len; b < b1 && b < b0 + (q + 1) * len; ++b) v += partial[(size_t)b * WS + e];
    run[q][threadIdx.x] = v;
    __syncthreads();
    if (q == 0 && e < WS) out[(size_t)g * WS + e] = ((run[0][threadIdx.x] + run[1][threadIdx.x]) + run[2][threadIdx.x]) + run[3][threadIdx.x];
}

// PBN_GRAM_LDS: 0 = gram_kernel (rows in registers), 1 = gram_lds_kernel, 2 (default) = the LDS-DMA ring kernels where they apply
// (contiguous rows: gram_glds_kernel for double tables, gram_glds_f32_kernel for float ones) and gram_lds_kernel elsewhere.
static int gram_variant() {
    static const int v = PBN_TUNE(GRAM_LDS, 2);
    return v;
}

// launches the partial-sum kernel; returns the number of blocks (= partials) it used (<= nblocks)
template <typename T, bool GATHER>
static int launch_gram_t(const GramArgs& a, int nct, int nblocks, hipStream_t st) {
    dim3 grid(nblocks), block(256);
    if constexpr (!GATHER) {
        if (gram_variant() >= 2) {
            if (a.num_cus > 0 && nblocks > GD_BLOCKS_PER_CU * a.num_cus) grid.x = GD_BLOCKS_PER_CU * a.num_cus;   // one block per slot
            const int dbg = a.debug_skip == 1 ? 1 : a.debug_skip >= 2 ? 2 : 0;
#ifdef PBN_GRAM_MEASURE   // measurement builds only (tools/gram_variants.sh): the floor variants produce garbage statistics
#define PBN_GLDS_K(K, N)                                                                                \
        if (dbg == 0) hipLaunchKernelGGL((K<N, 0>), grid, block, 0, st, a);                            \
        else if (dbg == 1) hipLaunchKernelGGL((K<N, 1>), grid, block, 0, st, a);                       \
        else hipLaunchKernelGGL((K<N, 2>), grid, block, 0, st, a);
#else
#define PBN_GLDS_K(K, N) (void)dbg; hipLaunchKernelGGL((K<N, 0>), grid, block, 0, st, a);
#endif
#define PBN_GLDS(N)                                                                                     \
    case N:                                                                                             \
        if constexpr (sizeof(T) == 8) { PBN_GLDS_K(gram_glds_kernel, N) } else { PBN_GLDS_K(gram_glds_f32_kernel, N) }   \
        break;
            switch (nct) {
                PBN_GLDS(1) PBN_GLDS(2) PBN_GLDS(3) PBN_GLDS(4)
                default: throw invalid_error("gram: at most 64 columns per launch");
            }
#undef PBN_GLDS
#undef PBN_GLDS_K
            HIP_CHECK(hipGetLastError());
            return (int)grid.x;
        }
    }
    if constexpr (GATHER) {
        if (gram_variant() >= 2) {
            switch (nct) {
                case 1: hipLaunchKernelGGL((gram_gring_kernel<T, 1, false>), grid, block, 0, st, a); break;
                case 2: hipLaunchKernelGGL((gram_gring_kernel<T, 2, false>), grid, block, 0, st, a); break;
                case 3: hipLaunchKernelGGL((gram_gring_kernel<T, 3, false>), grid, block, 0, st, a); break;
                case 4: hipLaunchKernelGGL((gram_gring_kernel<T, 4, false>), grid, block, 0, st, a); break;
                default: throw invalid_error("gram: at most 64 columns per launch");
            }
            HIP_CHECK(hipGetLastError());
            return nblocks;
        }
    }
    if (gram_variant() >= 1) {
        switch (nct) {
            case 1: hipLaunchKernelGGL((gram_lds_kernel<T, 1, GATHER>), grid, block, 0, st, a); break;
            case 2: hipLaunchKernelGGL((gram_lds_kernel<T, 2, GATHER>), grid, block, 0, st, a); break;
            case 3: hipLaunchKernelGGL((gram_lds_kernel<T, 3, GATHER>), grid, block, 0, st, a); break;
            case 4: hipLaunchKernelGGL((gram_lds_kernel<T, 4, GATHER>), grid, block, 0, st, a); break;
            default: throw invalid_error("gram: at most 64 columns per launch");
        }
        HIP_CHECK(hipGetLastError());
        return nblocks;
    }
    switch (nct) {
        case 1: hipLaunchKernelGGL((gram_kernel<T, 1, GATHER>), grid, block, 0, st, a); break;
        case 2: hipLaunchKernelGGL((gram_kernel<T, 2, GATHER>), grid, block, 0, st, a); break;
        case 3: hipLaunchKernelGGL((gram_kernel<T, 3, GATHER>), grid, block, 0, st, a); break;
        case 4: hipLaunchKernelGGL((gram_kernel<T, 4, GATHER>), grid, block, 0, st, a); break;
        default: throw invalid_error("gram: at most 64 columns per launch");
    }
    HIP_CHECK(hipGetLastError());
    return nblocks;
}

// 64 rows per block through an LDS tile: column reads and row writes both run along their contiguous direction; a thread's 4 NCT
// column reads are issued together
template <typename T, int NCT>
__global__ __launch_bounds__(256) void rowmajor_mirror_kernel(const T* __restrict__ base, int64_t ld, GramCols gc, int n_cols, int64_t n,
                                                              T* __restrict__ out) {
    constexpr int W = 16 * NCT;
    __shared__ T tile[64 * (W + 1)];
    const int tid = threadIdx.x, r = tid & 63, cq = tid >> 6;
    const int64_t r0 = (int64_t)blockIdx.x * 64;
    T v[4 * NCT];
#pragma unroll
    for (int k = 0; k < 4 * NCT; ++k) {
        const int col = 4 * k + cq;
        v[k] = (col < n_cols && r0 + r < n) ? base[(int64_t)gc.cols[col < n_cols ? col : 0] * ld + r0 + r] : (T)0;
    }
#pragma unroll
    for (int k = 0; k < 4 * NCT; ++k) {
        const int col = 4 * k + cq;
        tile[r * (W + 1) + NCT * (col & 15) + (col >> 4)] = v[k];
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 16 * NCT / 4; ++k) {
        const int e = k * 256 + tid, rr = e / W, kk = e % W;
        if (r0 + rr < n) out[(r0 + rr) * W + kk] = tile[rr * (W + 1) + kk];
    }
}

size_t rowmajor_mirror_elems(int64_t n, int n_cols) { return (size_t)n * (size_t)(16 * ((n_cols + 15) / 16)) + 16; }

void build_rowmajor_mirror(const void* base, int64_t ld, const GramCols& gc, int n_cols, int64_t n, int dtype, void* out, hipStream_t st) {
    if (n <= 0) return;
    const int nct = (n_cols + 15) / 16;
    if (nct < 1 || nct > 4) throw invalid_error("gram: at most 64 columns per mirror");
    const dim3 grid((unsigned)((n + 63) / 64)), block(256);
#define PBN_MIRROR(N)                                                                                                                         \
    case N:                                                                                                                                   \
        if (dtype == PBN_F64) hipLaunchKernelGGL((rowmajor_mirror_kernel<double, N>), grid, block, 0, st, (const double*)base, ld, gc, n_cols, n, (double*)out); \
        else hipLaunchKernelGGL((rowmajor_mirror_kernel<float, N>), grid, block, 0, st, (const float*)base, ld, gc, n_cols, n, (float*)out);       \
        break;
    switch (nct) { PBN_MIRROR(1) PBN_MIRROR(2) PBN_MIRROR(3) PBN_MIRROR(4) }
#undef PBN_MIRROR
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
    // PBN_GRAM_DEBUG (1 = no MFMAs, 2 = no loads: floors of the two halves, garbage statistics) exists in measurement builds only
    // (-DPBN_GRAM_MEASURE, tools/gram_variants.sh); the production library refuses it instead of silently returning garbage
    static const int dbg = PBN_TUNE(GRAM_DEBUG, 0);
#ifdef PBN_GRAM_MEASURE
    a.debug_skip = dbg;
#else
    if (dbg != 0) throw invalid_error("PBN_GRAM_DEBUG needs a library built with -DPBN_GRAM_MEASURE (tools/gram_variants.sh)");
    a.debug_skip = 0;
#endif
    // PBN_GRAM_STAMPS=1: gram_glds_kernel's blocks record their start / end (10 ns ticks) and hardware slot; printed to stderr
    static const bool want_stamps = PBN_TUNE(GRAM_STAMPS, 0) != 0;
    static long long* stamps_dev = nullptr;
    if (want_stamps && !stamps_dev) HIP_CHECK(hipMalloc(&stamps_dev, 4096 * 3 * sizeof(long long)));
    a.stamps = want_stamps && nblocks <= 4096 ? stamps_dev : nullptr;
    const int nct = (a.n_cols + 15) / 16;
    const int WS = gram_ws(nct);
    const bool gather = a.rows != nullptr;
    if (dtype == PBN_F64)
        nblocks = gather ? launch_gram_t<double, true>(a, nct, nblocks, st) : launch_gram_t<double, false>(a, nct, nblocks, st);
    else
        nblocks = gather ? launch_gram_t<float, true>(a, nct, nblocks, st) : launch_gram_t<float, false>(a, nct, nblocks, st);
    // tree of arity 16 over the blocks, fixed order
    int stride = 1;
    while ((nblocks + stride - 1) / stride > 16) {
        const int groups = (nblocks + stride * 16 - 1) / (stride * 16);
        hipLaunchKernelGGL(gram_reduce_kernel, dim3((WS + 255) / 256, groups), dim3(256), 0, st, a.partial, nblocks, WS, stride, (double*)nullptr);
        stride *= 16;
    }
    hipLaunchKernelGGL(gram_reduce_kernel, dim3((WS + 255) / 256, 1), dim3(256), 0, st, a.partial, nblocks, WS, stride, out);
    HIP_CHECK(hipGetLastError());
    if (a.stamps && dtype == PBN_F64 && !gather && gram_variant() >= 2) {
        std::vector<long long> h((size_t)nblocks * 3);
        HIP_CHECK(hipStreamSynchronize(st));
        HIP_CHECK(hipMemcpy(h.data(), stamps_dev, h.size() * sizeof(long long), hipMemcpyDeviceToHost));
        long long t0 = h[0];
        for (int b = 0; b < nblocks; ++b) t0 = std::min(t0, h[(size_t)b * 3]);
        fprintf(stderr, "GRAM_STAMPS %d blocks: block start end hw_id\n", nblocks);
        for (int b = 0; b < nblocks; ++b)
            fprintf(stderr, "S %d %lld %lld %llx\n", b, h[(size_t)b * 3] - t0, h[(size_t)b * 3 + 1] - t0, h[(size_t)b * 3 + 2]);
    }
}

void launch_gram_segments(const GramArgs& a_in, int dtype, int nblocks, const int32_t* blk_off, int n_seg, double* out, hipStream_t st) {
    GramArgs a = a_in;
    a.debug_skip = 0; a.stamps = nullptr;
    if (!a.blk) throw invalid_error("gram: the segmented form needs a block table");
    const int nct = (a.n_cols + 15) / 16;
    const int WS = gram_ws(nct);
    if (nblocks > 0) {
        const bool gather = a.rows != nullptr;
        const dim3 grid(nblocks), block(256);
        // the kernels of the plain form, block b on the piece blk[4b + 1 .. 2] -> partial slot blk[4b + 3]: the LDS-DMA rings for
        // contiguous rows, the register ring for row lists (PBN_GRAM_LDS < 2: gram_lds_kernel for both)
        const bool ring = gram_variant() >= 2;
#define PBN_SEG_N(K_RING, K_LDS)                                                                \
    switch (nct) {                                                                              \
        case 1: if (ring) hipLaunchKernelGGL((K_RING(1)), grid, block, 0, st, a); else hipLaunchKernelGGL((K_LDS(1)), grid, block, 0, st, a); break; \
        case 2: if (ring) hipLaunchKernelGGL((K_RING(2)), grid, block, 0, st, a); else hipLaunchKernelGGL((K_LDS(2)), grid, block, 0, st, a); break; \
        case 3: if (ring) hipLaunchKernelGGL((K_RING(3)), grid, block, 0, st, a); else hipLaunchKernelGGL((K_LDS(3)), grid, block, 0, st, a); break; \
        case 4: if (ring) hipLaunchKernelGGL((K_RING(4)), grid, block, 0, st, a); else hipLaunchKernelGGL((K_LDS(4)), grid, block, 0, st, a); break; \
        default: throw invalid_error("gram: at most 64 columns per launch");                    \
    }
#define PBN_K_GLDS64(N) gram_glds_kernel<N, 0>
#define PBN_K_GLDS32(N) gram_glds_f32_kernel<N, 0>
#define PBN_K_GRING64(N) gram_gring_kernel<double, N, false>
#define PBN_K_GRING32(N) gram_gring_kernel<float, N, false>
#define PBN_K_GROWS64(N) gram_gring_kernel<double, N, true>
#define PBN_K_GROWS32(N) gram_gring_kernel<float, N, true>
#define PBN_K_LDS64(N) gram_lds_kernel<double, N, false>
#define PBN_K_LDS32(N) gram_lds_kernel<float, N, false>
#define PBN_K_LDS64G(N) gram_lds_kernel<double, N, true>
#define PBN_K_LDS32G(N) gram_lds_kernel<float, N, true>
        const bool mirror = gather && ring && a.rowmajor != nullptr;
        if (dtype == PBN_F64) {
            if (mirror) { PBN_SEG_N(PBN_K_GROWS64, PBN_K_LDS64G) }
            else if (gather) { PBN_SEG_N(PBN_K_GRING64, PBN_K_LDS64G) }
            else { PBN_SEG_N(PBN_K_GLDS64, PBN_K_LDS64) }
        } else {
            if (mirror) { PBN_SEG_N(PBN_K_GROWS32, PBN_K_LDS32G) }
            else if (gather) { PBN_SEG_N(PBN_K_GRING32, PBN_K_LDS32G) }
            else { PBN_SEG_N(PBN_K_GLDS32, PBN_K_LDS32) }
        }
#undef PBN_SEG_N
#undef PBN_K_GLDS64
#undef PBN_K_GLDS32
#undef PBN_K_GRING64
#undef PBN_K_GRING32
#undef PBN_K_GROWS64
#undef PBN_K_GROWS32
#undef PBN_K_LDS64
#undef PBN_K_LDS32
#undef PBN_K_LDS64G
#undef PBN_K_LDS32G
    }
    if (n_seg > 0) hipLaunchKernelGGL(gram_seg_reduce_kernel, dim3((WS + 63) / 64, n_seg), dim3(64, 4), 0, st, a.partial, blk_off, WS, out);
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
