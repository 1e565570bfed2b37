// Grouped KDE evaluation (see kde_group.hpp): one launch chain for all (training set, test set) units of a batch of pools.
//
// Replaces, for the score engine's cross-validated / held-out CKDE terms, the per-(set, fold) chain of round 2
//   prune_keys -> rocprim sort -> tile_box -> pack | prune_keys -> rocprim sort -> pack -> prepass -> sweep -> finish -> reduce
// (kde_model.hip: kde_pack_train + kde_eval_enqueue; reference loop learning/scores/cv_likelihood.cpp:18-22 over
//  factors/continuous/CKDE.hpp:256-287) by
//   keys (all pools) -> ONE sort -> gather + region counts -> scan -> pack(train, all units) -> pack(query, all units)
//   -> boxes -> prepass -> sweep (flat grid over all units) -> finish -> reduce
// Stage by stage the arithmetic is the one of the single-unit chain (same whitening, same fragment layout, same sweep body,
// same merge of the split partials); what changes is the ORDER of the rows inside a unit (a compaction of the pool's Morton
// order instead of the unit's own Morton order - any order gives the same sums up to rounding, the integer offsets of the
// sweep keep the 2^x error attached to the pair) and the prepass bound (position by counting instead of by key search).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "kde_group.hpp"
#include "kde_kernels.hpp"
#include "kde_model.hpp"

namespace pbn {

namespace {

constexpr int GB = PBN_GROUP_BLOCK;
constexpr int MORTON_BITS = 24;   // low bits of a sort key; the pool's index inside the batch sits above them
// training rows scanned on either side of a query's position (kde_kernels.hip: PBN_PRUNE_WINDOW is the per-unit form's, 32).  Round 4: 8 - the
// sum bound now comes from the boxes of the surrounding tiles, the row scan only has to find a near row for the offsets; the 64-row scan was
// 0.36 s of C5's 8.2 s (8.17 -> 7.92 s, cv64 / C3 unchanged: profiles/r4/row_window_probe.txt)
#define PBN_GROUP_WINDOW 8

__host__ __device__ inline int group_key_bits(int kd) { return kd <= 1 ? 16 : MORTON_BITS / kd; }

struct GDev {   // argument block of the block-wise kernels
    const GPool* pools;
    const GUnit* units;
    const int32_t* blkpool;   // flat block -> pool
    const void* base;         // table
    int64_t ld;
    uint32_t* keys;           // [E]
    uint32_t* vals;           // [E] pool position of the element
    double* xs;               // [E][xstride] rows in sorted order
    uint8_t* reg;             // [E] region of the element
    int32_t* blkcnt;          // [flat blocks][Rs]: rows of region r in the block -> (after the scan) before the block
    char* arena;
    int xstride, Rs;
    int nunits;
    int use_sum_bound;
    int f16;                 // fp32 table: f16x2 fragments (kde_kernels.hip pack_rows_f16_kernel), KS = number of f16 MFMAs
    int KS;                   // MFMAs per (tile, group) of the chunk's sweep shape
    int window;               // training rows the prepass scans on either side of a query's position (PBN_GROUP_WINDOW)
    unsigned long long* out_max;   // nullable: [sum_slot] bits of the largest |z|^2 of a unit's training rows (f16 chunks)
    int hilbert;                   // Hilbert order instead of Z-order: 1 = at two key dimensions, 2 = at three and four as well (group_keys_kernel)
    int fine_keys;                 // 1-2 key dimensions: cells of sigma / 4096 (256) instead of sigma / 16 (group_keys_kernel)
    int tile_window;               // training TILES on either side of a query tile's position whose boxes bound the queries' sums from below (0 = off)
};

__device__ __forceinline__ int region_of(const GPool& P, int pp) {
    int lo = 0, hi = P.R;   // largest r with rb[r] <= pp
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (P.rb[mid] <= pp) lo = mid; else hi = mid; }
    return lo;
}

// ---- keys: Morton order of pool-standardised coordinates -----------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(GB) void group_keys_kernel(GDev g) {
    const int fb = blockIdx.x;
    const GPool& P = g.pools[g.blkpool[fb]];
    const int blk = fb - P.blk0;
    const int p = blk * GB + (int)threadIdx.x;
    if (blk >= P.nblk || p >= P.n) return;
    const int64_t row = P.rows ? (int64_t)P.rows[P.row_base + p] : P.row_base + p;
    const int d = P.d, kd = P.kd;
    double xc[PBN_GROUP_MAX_D];
    for (int j = 0; j < kd; ++j) xc[j] = (double)((const T*)g.base + (int64_t)P.cols[j] * g.ld)[row] - P.mug[j];
    const int bits = group_key_bits(kd);
    const double half = (double)(1 << (bits - 1)), top = (double)((1 << bits) - 1);
    // cells of sigma / 16 (8, 4) at 8 (6, fewer) bits per axis: the key range covers +-8 (4) sigma.  One or two key dimensions have 16 / 12
    // bits per axis: the same +-8 sigma in cells of sigma / 4096 (sigma / 256) - round 5: with sigma / 16 cells a cell of a 450 000-row fold held
    // ~280 rows (d = 2; ~11 000 at d = 1) in arbitrary order, i.e. a 16-row tile was 16 random rows of a box half a bandwidth wide
    // (tools/farfield_feasibility.py: median tile radius 0.34 bandwidths against 0.15 with fine cells)
    const double scale = (bits >= 12 && g.fine_keys) ? (double)(1 << (bits - 4)) : (bits >= 8 ? 16.0 : (bits >= 6 ? 8.0 : 4.0));   // (3 / 4 key dimensions: finer cells change nothing, the cells are smaller than the tiles already)
    uint32_t key = 0;
    uint32_t cells[PBN_PRUNE_PD];
    for (int i = 0; i < kd; ++i) {
        double u = 0.0;
        for (int j = 0; j <= i; ++j) u = __builtin_fma(P.Wg[i * d + j], xc[j], u);
        double c = __builtin_floor(u * scale) + half;
        c = c < 0.0 ? 0.0 : (c > top ? top : c);
        cells[i] = (uint32_t)c;
    }
    if (kd == 2 && g.hilbert) {
        // two key dimensions: position along the HILBERT curve instead of the Z-order - consecutive rows stay neighbours across every
        // quadrant boundary, where the Z-order jumps (a tile that straddles a jump has a box as large as the jump)
        uint32_t x = cells[0], y = cells[1];
        const uint32_t n1 = (1u << bits) - 1u;
        for (uint32_t sq = 1u << (bits - 1); sq > 0; sq >>= 1) {
            const uint32_t rx = (x & sq) ? 1u : 0u, ry = (y & sq) ? 1u : 0u;
            key += sq * sq * ((3u * rx) ^ ry);
            if (ry == 0) {
                if (rx == 1) { x = n1 - x; y = n1 - y; }
                const uint32_t tmp = x; x = y; y = tmp;
            }
        }
    } else if (kd >= 3 && g.hilbert >= 2) {
        key = hilbert_key(cells, kd, bits);   // three / four key dimensions (8 / 6 bits per axis): the same curve in its n-dimensional form
    } else {
        for (int i = 0; i < kd; ++i)
            for (int b = 0; b < bits; ++b) key |= ((cells[i] >> b) & 1u) << (b * kd + i);
    }
    g.keys[P.elem0 + p] = key | ((uint32_t)P.local << MORTON_BITS);
    g.vals[P.elem0 + p] = (uint32_t)p;
}

// ---- gather the rows into sorted order, region of every element, rows per (block, region) -----------------------------------
template <typename T>
__global__ __launch_bounds__(GB) void group_gather_kernel(GDev g) {
    __shared__ int cnt[PBN_GROUP_MAX_R];
    const int fb = blockIdx.x;
    const GPool& P = g.pools[g.blkpool[fb]];
    const int blk = fb - P.blk0;
    if (blk >= P.nblk) return;
    if ((int)threadIdx.x < PBN_GROUP_MAX_R) cnt[threadIdx.x] = 0;
    __syncthreads();
    const int i = blk * GB + (int)threadIdx.x;
    if (i < P.n) {
        const int pp = (int)g.vals[P.elem0 + i];
        const int64_t row = P.rows ? (int64_t)P.rows[P.row_base + pp] : P.row_base + pp;
        double* x = g.xs + (P.elem0 + i) * g.xstride;
        for (int j = 0; j < P.d; ++j) x[j] = (double)((const T*)g.base + (int64_t)P.cols[j] * g.ld)[row];
        const int r = region_of(P, pp);
        g.reg[P.elem0 + i] = (uint8_t)r;
        atomicAdd(&cnt[r], 1);   // integer counts: the order of the additions does not matter
    }
    __syncthreads();
    if ((int)threadIdx.x < P.R) g.blkcnt[(int64_t)fb * g.Rs + threadIdx.x] = cnt[threadIdx.x];
}

// ---- exclusive scan of the block counts, per (pool, region) ---------------------------------------------------------------
__global__ __launch_bounds__(64) void group_scan_kernel(GDev g) {
    const GPool& P = g.pools[blockIdx.y];
    const int r = blockIdx.x;
    if (r >= P.R) return;
    const int lane = threadIdx.x;
    int carry = 0;
    for (int b0 = 0; b0 < P.nblk; b0 += 64) {
        const int b = b0 + lane;
        int32_t* cell = g.blkcnt + (int64_t)(P.blk0 + b) * g.Rs + r;
        const int v = b < P.nblk ? *cell : 0;
        int inc = v;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int o = __shfl_up(inc, off);
            if (lane >= off) inc += o;
        }
        if (b < P.nblk) *cell = carry + inc - v;
        carry += __shfl(inc, 63);
    }
}

// whitened coordinates of one row, fragment stores shared by the two pack kernels (fp64 classic fragments, kde_kernels.hip
// pack_rows_kernel: training side norms in C-row order + weights 2^norm, query side norms by row; fp32: f16x2 fragments)
// returns |z|^2 of the row (f16 chunks; 0 otherwise)
__device__ __forceinline__ double pack_store(const GDev& g, const GUnit& U, const double* x, int d, int dest, bool query, int32_t tpos) {
    char* arena = g.arena;
    const int KS = g.KS;
    const int tile = dest >> 4, idx = dest & 15;
    double xc[PBN_GROUP_MAX_D];
    for (int j = 0; j < d; ++j) xc[j] = x[j] - U.mu[j];
    double* zrow = (double*)(arena + (query ? U.zq : U.zs)) + (int64_t)dest * d;
    if (query) ((int32_t*)(arena + U.qpos))[dest] = tpos;
    if (g.f16) {
        hpiece p1[PBN_GROUP_MAX_D], p2[PBN_GROUP_MAX_D];
        double nrm = 0.0;
        bool far = false;
        for (int c = 0; c < d; ++c) {
            double z = 0.0;
            for (int j = 0; j <= c; ++j) z = __builtin_fma(U.W[c * d + j], xc[j], z);
            bool cl;
            const double zr = split2(z, p1[c], p2[c], cl);
            far = far || cl;
            zrow[c] = zr;            // the value the fragments carry
            nrm = __builtin_fma(zr, zr, nrm);
        }
        const float nv = (float)(-0.5 * nrm);
        f16x2_store_row((hf8*)(arena + (query ? U.bpack : U.apack)), KS, tile, idx, d, p1, p2, nv, query);
        if (query) ((float*)(arena + U.ny))[(int64_t)tile * 16 + idx] = nv;
        return far ? INFINITY : nrm;   // a row beyond the f16 range: the caller's |z|^2 check sends the unit to fp64 fragments
    }
    double* pack = (double*)(arena + (query ? U.bpack : U.apack));
    double nrm = 0.0;
    for (int c = 0; c < KS * 4; ++c) {
        double z = 0.0;
        if (c < d) {
            for (int j = 0; j <= c; ++j) z = __builtin_fma(U.W[c * d + j], xc[j], z);
            zrow[c] = z;
        }
        nrm = __builtin_fma(z, z, nrm);
        pack[((int64_t)tile * KS + (c >> 2)) * 64 + (c & 3) * 16 + idx] = z;
    }
    const double nv = -0.5 * nrm;
    if (d < KS * 4) pack[((int64_t)tile * KS + (d >> 2)) * 64 + (d & 3) * 16 + idx] = query ? 1.0 : nv;   // norm in the free K slot (FOLD)
    if (query) {
        ((double*)(arena + U.ny))[(int64_t)tile * 16 + idx] = nv;
    } else {
        double* np = (double*)(arena + U.npack);
        const int lg = idx & 3, i = idx >> 2;
        np[(int64_t)tile * 16 + lg * 4 + i] = nv;
        np[(int64_t)U.ntiles * 16 + (int64_t)tile * 16 + lg * 4 + i] = nv < -1000.0 ? (double)NAN : exp2(nv);   // weights of the WMUL sweep
    }
    return 0.0;
}

// ---- training side: grid (flat blocks, units per pool).  A unit's training rows are the elements of its regions, in the pool's
// order: destination = number of such elements before it (block offsets from the scan + a ballot inside the block) ---------------
__global__ __launch_bounds__(GB) void group_pack_train_kernel(GDev g) {
    __shared__ int wcount[GB / 64];
    const int fb = blockIdx.x;
    const GPool& P = g.pools[g.blkpool[fb]];
    const int blk = fb - P.blk0;
    if (blk >= P.nblk || (int)blockIdx.y >= P.nunits) return;
    const GUnit& U = g.units[P.unit0 + blockIdx.y];
    const int i = blk * GB + (int)threadIdx.x;
    const bool valid = i < P.n;
    const int r = valid ? (int)g.reg[P.elem0 + i] : 0;
    const bool in = valid && ((U.train_mask >> r) & 1ull);
    int base = 0;
    for (int rr = 0; rr < P.R; ++rr)
        if ((U.train_mask >> rr) & 1ull) base += g.blkcnt[(int64_t)fb * g.Rs + rr];
    const unsigned long long b = __ballot(in);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) wcount[wave] = __builtin_popcountll(b);
    __syncthreads();
    for (int w = 0; w < wave; ++w) base += wcount[w];
    double z2 = 0.0;
    if (in) {
        const int dest = base + __builtin_popcountll(b & ((1ull << lane) - 1ull));
        z2 = pack_store(g, U, g.xs + (P.elem0 + i) * g.xstride, P.d, dest, false, 0);
    }
    if (g.out_max) {   // the unit's farthest whitened training row: one atomic per wave
        if (!(z2 == z2)) z2 = INFINITY;
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(z2, off);
            z2 = o > z2 ? o : z2;
        }
        // (a plain load first: once the slot holds the unit's large norms almost no wave has anything to add, and thousands of atomics
        //  on one address serialise - 0.4 s of C5's 8.7 s without the test; a stale value only costs an atomic that changes nothing)
        if (lane == 0 && z2 > 0.0) {
            unsigned long long* slot = g.out_max + U.sum_slot;
            const unsigned long long bits = (unsigned long long)__double_as_longlong(z2);
            if (bits > *(volatile unsigned long long*)slot) atomicMax(slot, bits);
        }
    }
}

// ---- query side + all padding: grid (flat blocks).  Every element is a test row of at most one unit of its pool (the unit whose
// test region is the element's); its destination is its rank inside the region, and the number of the unit's TRAINING rows
// before it - its position in the unit's training order - is what the prepass starts its neighbour scan from -------------------
__global__ __launch_bounds__(GB) void group_pack_query_kernel(GDev g) {
    __shared__ unsigned long long ball[GB / 64][PBN_GROUP_MAX_R];
    const int fb = blockIdx.x;
    const GPool& P = g.pools[g.blkpool[fb]];
    const int blk = fb - P.blk0;
    const int d = P.d, KS = g.KS;
    if (blk >= P.nblk) {
        // the pool's padding block: rows N .. 16 ntiles of every unit's training pack (norm -1e30: their terms vanish; weight
        // 0), rows nq .. 16 nqtiles of its query pack (coordinates 0, norm 0: finite sums nobody reads)
        for (int v = threadIdx.x; v < P.nunits * 32; v += GB) {
            const GUnit& U = g.units[P.unit0 + (v >> 5)];
            const int j = v & 31;
            const bool query = j >= 16;
            const int row = (query ? U.nq : U.N) + (j & 15);
            if (row >= (query ? U.nqtiles : U.ntiles) * 16) continue;
            const int tile = row >> 4, idx = row & 15;
            if (g.f16) {
                hpiece z1[PBN_GROUP_MAX_D], z2[PBN_GROUP_MAX_D];
                for (int c = 0; c < d; ++c) z1[c] = z2[c] = (hpiece)0.0f;
                const float nv = query ? 0.0f : -1e30f;
                f16x2_store_row((hf8*)(g.arena + (query ? U.bpack : U.apack)), KS, tile, idx, d, z1, z2, nv, query);
                if (query) ((float*)(g.arena + U.ny))[(int64_t)tile * 16 + idx] = 0.0f;
                continue;
            }
            double* pack = (double*)(g.arena + (query ? U.bpack : U.apack));
            for (int c = 0; c < KS * 4; ++c) pack[((int64_t)tile * KS + (c >> 2)) * 64 + (c & 3) * 16 + idx] = 0.0;
            if (d < KS * 4) pack[((int64_t)tile * KS + (d >> 2)) * 64 + (d & 3) * 16 + idx] = query ? 1.0 : -1e30;
            if (query) {
                ((double*)(g.arena + U.ny))[(int64_t)tile * 16 + idx] = 0.0;
            } else {
                double* np = (double*)(g.arena + U.npack);
                const int lg = idx & 3, i = idx >> 2;
                np[(int64_t)tile * 16 + lg * 4 + i] = -1e30;
                np[(int64_t)U.ntiles * 16 + (int64_t)tile * 16 + lg * 4 + i] = 0.0;
            }
        }
        return;
    }
    const int i = blk * GB + (int)threadIdx.x;
    const bool valid = i < P.n;
    const int r = valid ? (int)g.reg[P.elem0 + i] : -1;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int rr = 0; rr < P.R; ++rr) {
        const unsigned long long b = __ballot(r == rr);
        if (lane == 0) ball[wave][rr] = b;
    }
    __syncthreads();
    if (!valid) return;
    const int uidx = P.test_unit[r];
    if (uidx < 0) return;
    const GUnit& U = g.units[P.unit0 + uidx];
    const unsigned long long lt = (1ull << lane) - 1ull;
    auto before = [&](int rr) {   // elements of region rr before element i
        int c = g.blkcnt[(int64_t)fb * g.Rs + rr];
        for (int w = 0; w < wave; ++w) c += __builtin_popcountll(ball[w][rr]);
        return c + __builtin_popcountll(ball[wave][rr] & lt);
    };
    const int dest = before(r);
    int tpos = 0;
    for (int rr = 0; rr < P.R; ++rr)
        if ((U.train_mask >> rr) & 1ull) tpos += before(rr);
    const double z2 = pack_store(g, U, g.xs + (P.elem0 + i) * g.xstride, d, dest, true, tpos);
    // a test row beyond the f16 range of the fp32 fragments: the unit is redone on fp64 fragments like one with far training rows
    if (g.out_max && !(z2 < INFINITY)) atomicMax(g.out_max + U.sum_slot, (unsigned long long)__double_as_longlong((double)INFINITY));
}

// ---- bounding boxes of the 16-row training tiles: grid (blocks of 256 padded rows, units) ---------------------------------------
__global__ __launch_bounds__(256) void group_tile_box_kernel(GDev g) {
    const GUnit& U = g.units[blockIdx.y];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (blockIdx.x * 256 >= U.ntiles * 16) return;
    const GPool& P = g.pools[U.pool];
    const int d = P.d, pd = P.kd;
    const bool valid = r < U.N;
    double lo[PBN_PRUNE_PD], hi[PBN_PRUNE_PD];
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k) { lo[k] = INFINITY; hi[k] = -INFINITY; }
    if (valid) {
        const double* z = (const double*)(g.arena + U.zs) + (int64_t)r * d;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k)
            if (k < pd) { const double v = z[k]; if (v == v) { lo[k] = v; hi[k] = v; } }
    }
    for (int off = 1; off < 16; off <<= 1) {
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k) {
            const double l = __shfl_xor(lo[k], off), h = __shfl_xor(hi[k], off);
            lo[k] = l < lo[k] ? l : lo[k];
            hi[k] = h > hi[k] ? h : hi[k];
        }
    }
    if ((threadIdx.x & 15) == 0 && r < U.ntiles * 16) {
        double* box = (double*)(g.arena + U.box) + (int64_t)(r >> 4) * 2 * pd;
        for (int k = 0; k < pd; ++k) { box[k] = lo[k]; box[pd + k] = hi[k]; }
    }
}

// ---- boxes of the 64-tile batches the sweeps walk: batch k of split s = tiles [s * tps + 64 k, ...) of the unit: grid (batches, units) ------
__global__ __launch_bounds__(64) void group_batch_box_kernel(GDev g) {
    const GUnit& U = g.units[blockIdx.y];
    const int nbps = (U.tps + 63) / 64;
    if ((int)blockIdx.x >= U.nsplit * nbps) return;
    const GPool& P = g.pools[U.pool];
    const int pd = P.kd;
    const int split = blockIdx.x / nbps, k = blockIdx.x - split * nbps;
    const int t0 = split * U.tps, t1 = t0 + U.tps < U.ntiles ? t0 + U.tps : U.ntiles;
    const int t = t0 + 64 * k + (int)threadIdx.x;
    double lo[PBN_PRUNE_PD], hi[PBN_PRUNE_PD];
#pragma unroll
    for (int i = 0; i < PBN_PRUNE_PD; ++i) { lo[i] = INFINITY; hi[i] = -INFINITY; }
    if (t < t1) {
        const double* bx = (const double*)(g.arena + U.box) + (int64_t)t * 2 * pd;
#pragma unroll
        for (int i = 0; i < PBN_PRUNE_PD; ++i)
            if (i < pd) { lo[i] = bx[i]; hi[i] = bx[pd + i]; }
    }
    for (int off = 1; off < 64; off <<= 1) {
#pragma unroll
        for (int i = 0; i < PBN_PRUNE_PD; ++i) {
            const double l = __shfl_xor(lo[i], off), h = __shfl_xor(hi[i], off);
            lo[i] = l < lo[i] ? l : lo[i];
            hi[i] = h > hi[i] ? h : hi[i];
        }
    }
    if (threadIdx.x == 0) {
        double* bb = (double*)(g.arena + U.bbox) + (int64_t)blockIdx.x * 2 * pd;
        for (int i = 0; i < pd; ++i) { bb[i] = lo[i]; bb[pd + i] = hi[i]; }
    }
}

// ---- tile moments (round 5; units of one or two dimensions on fp64 fragments): per 16-row training tile its centroid c, the squared
// radius rho^2 = max_t |z_t - c|^2 and the coefficients of the order-8 expansion of its contribution about c,
//   sum_t 2^(-|u - delta_t|^2 / 2) = 2^(-|u|^2 / 2) sum_alpha C_alpha u^alpha,   C_alpha = a^|alpha| / alpha! sum_t 2^(-|delta_t|^2 / 2) delta_t^alpha,  a = ln 2,
// in the order the Horner scheme of kde_moment_group_kernel reads them (kde_kernels.hpp: PBN_MOM_ORDER, pbn_mom_rec).  Padding rows of the
// last tile carry no weight.  grid (blocks of 16 tiles, units), one lane per row --------------------------------------------------------------
template <int D>
__global__ __launch_bounds__(256) void group_tile_moments_kernel(GDev g) {
    const GUnit& U = g.units[blockIdx.y];
    const int r = blockIdx.x * 256 + (int)threadIdx.x;
    if (blockIdx.x * 256 >= U.ntiles * 16 || U.mom == 0) return;
    const GPool& P = g.pools[U.pool];
    if (P.d != D) return;
    constexpr int ORD = PBN_MOM_ORDER;
    const int64_t mstride = ((int64_t)U.ntiles + 63) / 64 * 64;   // structure of arrays: value k of tile t at [k * mstride + t]
    constexpr double A = 0.693147180559945309417232121458;
    const bool valid = r < U.N;
    double z[D];
#pragma unroll
    for (int k = 0; k < D; ++k) z[k] = valid ? ((const double*)(g.arena + U.zs))[(int64_t)r * D + k] : 0.0;
    // centroid of the tile's real rows
    double cnt = valid ? 1.0 : 0.0, c[D];
#pragma unroll
    for (int k = 0; k < D; ++k) c[k] = z[k];
    for (int off = 1; off < 16; off <<= 1) {
        cnt += __shfl_xor(cnt, off);
#pragma unroll
        for (int k = 0; k < D; ++k) c[k] += __shfl_xor(c[k], off);
    }
    double dl[D], r2 = 0.0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        c[k] = cnt > 0.0 ? c[k] / cnt : 0.0;
        dl[k] = valid ? z[k] - c[k] : 0.0;
        r2 = __builtin_fma(dl[k], dl[k], r2);
    }
    const double w = valid ? exp2(-0.5 * r2) : 0.0;
    double rmax = r2;
    for (int off = 1; off < 16; off <<= 1) { const double o = __shfl_xor(rmax, off); rmax = o > rmax ? o : rmax; }
    // powers of the offsets, then every coefficient as a 16-lane sum
    double px[ORD + 1], py[ORD + 1];
    px[0] = 1.0; py[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= ORD; ++i) { px[i] = px[i - 1] * dl[0]; py[i] = D == 2 ? py[i - 1] * dl[D - 1] : 0.0; }
    const int tile = r >> 4;
    const bool writer = (threadIdx.x & 15) == 0 && tile < U.ntiles;
    double* rec = (double*)(g.arena + U.mom) + tile;
    if (writer) {
#pragma unroll
        for (int k = 0; k < D; ++k) rec[k * mstride] = c[k];
        ((float*)(g.arena + U.rad2))[tile] = (float)rmax * 1.000001f + 1e-30f;   // rounded up
    }
    double fact[ORD + 1];
    fact[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= ORD; ++i) fact[i] = fact[i - 1] * (double)i;
    double apow[ORD + 1];
    apow[0] = 1.0;
#pragma unroll
    for (int i = 1; i <= ORD; ++i) apow[i] = apow[i - 1] * A;
    int k = D;
    if constexpr (D == 1) {
#pragma unroll
        for (int i = ORD; i >= 0; --i) {
            double v = w * px[i];
            for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
            if (writer) rec[k * mstride] = v * apow[i] / fact[i];
            ++k;
        }
    } else {
#pragma unroll
        for (int j = ORD; j >= 0; --j)
#pragma unroll
            for (int i = ORD - j; i >= 0; --i) {
                double v = w * px[i] * py[j];
                for (int off = 1; off < 16; off <<= 1) v += __shfl_xor(v, off);
                if (writer) rec[k * mstride] = v * apow[i + j] / (fact[i] * fact[j]);
                ++k;
            }
    }
}

// ---- per query: a lower bound of its largest exponent (and of its whole sum) from the training rows around its position in the
// training order; per 16-query tile the smallest bound and the box: grid (blocks of 256 padded queries, units) -----------------
__global__ __launch_bounds__(256) void group_prepass_kernel(GDev g) {
    const GUnit& U = g.units[blockIdx.y];
    const int q = blockIdx.x * 256 + (int)threadIdx.x;
    if (blockIdx.x * 256 >= U.nqtiles * 16) return;
    const GPool& P = g.pools[U.pool];
    const int d = P.d, pd = P.kd;
    const bool valid = q < U.nq;
    double z[PBN_GROUP_MAX_D];
    double best = -INFINITY, acc = 0.0;
    if (valid) {
        const double* zp = (const double*)(g.arena + U.zq) + (int64_t)q * d;
        for (int k = 0; k < d; ++k) z[k] = zp[k];
        const int tpos = ((const int32_t*)(g.arena + U.qpos))[q];
        const int b = tpos - g.window > 0 ? tpos - g.window : 0, e = tpos + g.window < U.N ? tpos + g.window : U.N;
        const double* zt = (const double*)(g.arena + U.zs);
        for (int t = b; t < e; ++t) {
            double d2 = 0.0;
            for (int k = 0; k < d; ++k) { const double dd = zt[(int64_t)t * d + k] - z[k]; d2 = __builtin_fma(dd, dd, d2); }
            const double ex = -0.5 * d2;
            if (ex > best) { acc = acc * exp2(best - ex) + 1.0; best = ex; }
            else acc += exp2(ex - best);
        }
    }
    // the pruning threshold may stand on the bound of the query's SUM (>= the sum over the scanned rows): what a skipped tile
    // could add is then below 2^-margin of the sum itself, not merely of its largest term
    double thr = valid ? (g.use_sum_bound && acc > 0.0 ? best + log2(acc) : best) : INFINITY;
    double lob[PBN_PRUNE_PD], hib[PBN_PRUNE_PD];
#pragma unroll
    for (int k = 0; k < PBN_PRUNE_PD; ++k) { lob[k] = (valid && k < pd) ? z[k] : INFINITY; hib[k] = (valid && k < pd) ? z[k] : -INFINITY; }
    for (int off = 1; off < 16; off <<= 1) {
        const double o = __shfl_xor(thr, off);
        thr = o < thr ? o : thr;
#pragma unroll
        for (int k = 0; k < PBN_PRUNE_PD; ++k) {
            const double l = __shfl_xor(lob[k], off), h = __shfl_xor(hib[k], off);
            lob[k] = l < lob[k] ? l : lob[k];
            hib[k] = h > hib[k] ? h : hib[k];
        }
    }
    // Round 4: a second lower bound of the 16 queries' sums, from the BOXES of the training tiles around their position in the training
    // order (the boxes cover every dimension on this path): every row of a tile T lies within maxdist(query box, box_T) of every query,
    // so S_q >= sum_T 16 * 2^(-1/2 maxdist^2) over any set of full tiles.  The 64 scanned ROWS above reach 2^6 at best where the sum of
    // a query in a dense region holds thousands of comparable terms; a window of +-256 TILES (8 192 rows, 32 boxes per lane) recovers 4-6 of
    // those bits, and every bit of the bound is a bit of pruning radius: cv64 2.52 -> 2.45 s, C3's first iteration 11.56 -> 10.95 s, C5
    // 8.37 -> 8.14 s (profiles/r4/tile_window_probe.txt; PBN_GROUP_TILE_WINDOW, 0 = off).  The largest single box term also raises the
    // queries' offsets (qlb), which keeps them within log2(16 x 512 tiles) = 13 units of the threshold (the fp32 tail path of far tiles
    // needs less than 26).
    if (g.tile_window > 0 && g.use_sum_bound && pd == d) {
        const int tp0 = __shfl(valid ? ((const int32_t*)(g.arena + U.qpos))[q] : 0, 0, 16);   // position of the tile's first query
        const int full = U.N >> 4;                                                           // tiles whose 16 rows are all real
        const int tt = tp0 >> 4;
        const int t_lo = tt - g.tile_window > 0 ? tt - g.tile_window : 0, t_hi = tt + g.tile_window < full ? tt + g.tile_window : full;
        const double* boxes = (const double*)(g.arena + U.box);
        {
            // (Round 5, measured and dropped: the box bound per QUERY - point-to-box distances, the group's threshold the smallest of its
            //  queries' bounds - is 1.5 bits tighter on C3's folds and 2-3 on cv64's (tools/sum_bound_slack.py), and the sweeps get 1-3 %
            //  faster; but every lane then walks the 512 boxes of the window itself, a latency-bound loop: cv64 2.36 -> 2.48 s, C3's first
            //  iteration 10.26 -> 10.38 s, C5 7.31 -> 7.78 s, profiles/r5/query_bounds_probe.txt.)
            const int l16 = threadIdx.x & 15;
            double bmax = -INFINITY, bacc = 0.0;
            const bool boxok = lob[0] <= hib[0];   // the tile holds a valid query
            if (boxok)
                for (int t = t_lo + l16; t < t_hi; t += 16) {
                    const double* bx = boxes + (int64_t)t * 2 * pd;
                    double d2 = 0.0;
                    for (int k = 0; k < pd; ++k) {
                        const double a1 = bx[pd + k] - lob[k], a2 = hib[k] - bx[k];
                        const double a = a1 > a2 ? a1 : a2;
                        d2 = __builtin_fma(a, a, d2);
                    }
                    const double ex = -0.5 * d2;
                    if (!(ex == ex)) continue;
                    if (ex > bmax) { bacc = bacc * exp2(bmax - ex) + 1.0; bmax = ex; }
                    else bacc += exp2(ex - bmax);
                }
            for (int off = 1; off < 16; off <<= 1) {   // merge the 16 lanes' (max, sum) pairs
                const double om = __shfl_xor(bmax, off), oa = __shfl_xor(bacc, off);
                if (om > bmax) { bacc = bacc * exp2(bmax - om) + oa; bmax = om; }
                else if (om > -INFINITY) bacc += oa * exp2(om - bmax);
            }
            if (bacc > 0.0) {
                const double tb = bmax + log2(bacc) + 4.0;   // 16 rows per tile
                if (tb > thr && thr < INFINITY) thr = tb;
                if (valid && bmax > best) best = bmax;       // a valid lower bound of this query's largest exponent, too
            }
        }
    }
    if (q < U.nqtiles * 16) ((double*)(g.arena + U.qlb))[q] = valid ? best : -INFINITY;
    if (valid && (threadIdx.x & 15) == 0) {
        const int tile = q >> 4;
        ((double*)(g.arena + U.qthr))[tile] = thr;
        double* qb = (double*)(g.arena + U.qbox) + (int64_t)tile * 2 * pd;
        for (int k = 0; k < pd; ++k) { qb[k] = lob[k]; qb[pd + k] = hib[k]; }
    }
}

// ---- merge the split partials per query (kde_finish_kernel's arithmetic), block sums: grid (blocks of 256 queries, units) ---------
__global__ __launch_bounds__(256) void group_finish_kernel(GDev g) {
    constexpr double LN2 = 0.693147180559945309417232121458;
    const GUnit& U = g.units[blockIdx.y];
    if (blockIdx.x * 256 >= U.nq) return;
    const int q = blockIdx.x * 256 + (int)threadIdx.x;
    double val = 0.0;
    if (q < U.nq) {
        const double* p = (const double*)(g.arena + U.part) + (int64_t)q * 2;
        const int64_t stride = (int64_t)U.nqtiles * 16 * 2;
        double m = p[0];
        for (int sp = 1; sp < U.nsplit_fin; ++sp) { const double m2 = p[sp * stride]; m = m > m2 ? m : m2; }
        double s = 0.0;
#pragma unroll 4
        for (int sp = 0; sp < U.nsplit_fin; ++sp) { const double* pp = p + sp * stride; s += pp[1] * exp2(pp[0] - m); }
        val = U.lognorm + LN2 * (m + log2(s));
    }
    __shared__ double red[256];
    red[threadIdx.x] = val;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) ((double*)(g.arena + U.bsum))[blockIdx.x] = red[0];
}

__global__ __launch_bounds__(256) void group_reduce_kernel(GDev g, double* out) {
    const GUnit& U = g.units[blockIdx.x];
    const double* in = (const double*)(g.arena + U.bsum);
    const int n = (U.nq + 255) / 256;
    __shared__ double red[256];
    double v = 0.0;
    for (int i = threadIdx.x; i < n; i += 256) v += in[i];
    red[threadIdx.x] = v;
    __syncthreads();
#pragma unroll
    for (int s = 128; s > 0; s >>= 1) {
        if ((int)threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
        __syncthreads();
    }
    if (threadIdx.x == 0) out[U.sum_slot] = red[0];
}

size_t al256(size_t x) { return (x + 255) / 256 * 256; }

// Training tiles per split of a unit's sweep.  PBN_GROUP_SPLIT_TILES pins it.  fp32 sweeps: 512 up to 16 384 tiles, 1 024 up to 32 768 tiles and 2 048 above (C5's 720 000-row
// slices: within noise of 512 in time, `profiles/r4/split_tiles_probe.txt` - but every query keeps a partial per split, 60 % of a candidate's
// arena at 512, so the coarser split lets an arena-full hold twice the candidates); the fp64 sweeps take 4 096 since round 5: with the two-level
// walk a longer split costs one more ballot per 4 096 tiles, and the moment pass beside the sweep halves the work per workgroup - C3's 450 000-row
// folds: first iteration 8.08 (1 024) / 7.83 (2 048) / 7.79 (4 096) / 7.97 s (16 384), six iterations 12.68 -> 12.00 s (`profiles/r5/moment_pass.txt` section 8);
// with two query groups per wave also below 16 384 tiles (cv64's 90 000-row folds: 2.059 (512) / 2.043 (1 024) / 2.05 (2 048) / 2.022 s (4 096); 256: 2.12 s).
int split_tiles_for(int ntiles, bool f64) {
    const int pinned = knob_int("PBN_GROUP_SPLIT_TILES", 0);   // read per call: the tests switch it
    if (pinned > 0) return std::max(16, pinned);
    if (f64) return 4096;
    return ntiles <= 16384 ? 512 : (ntiles <= 32768 ? 1024 : 2048);
}

// The moment pass pays from a density of training rows on: a 16-row tile must be small against the bandwidth for many of its (tile, group)
// pairs to pass the criterion.  cv64-shaped searches (64 nodes, 10 folds, first iteration) with the pass forced on against off
// (tools/r5_probe_m.sh, profiles/r5/moment_pass.txt): 90 000 training rows 2.26 -> 2.99 s, 135 000 4.08 -> 5.28 s, 180 000 6.65 -> 8.07 s,
// 270 000 12.79 -> 13.56 s; C3 (450 000) 9.04 -> 8.08 s.  Lane utilisation (tools/moment_visits.py): 40-44 of 64 at 57 600 rows with a third
// of the pairs left to the sweep, 56-62 at 450 000 with 6 % left.
// (round 5: 400 000; round 6, after the pass got 11 % faster, tools/moment_rows_probe.sh: cv64-shaped searches with the pass forced / off at
//  90 000 / 180 000 / 270 000 / 360 000 training rows: 2.45 / 2.10 s, 6.40 / 6.19 s, 11.18 / 12.02 s, 16.98 / 19.47 s - it pays from ~220 000)
int moment_pass_rows() { return knob_int("PBN_MOMENT_MIN_ROWS", 250000); }
// The training rows the moment-pass rule looks at.  A leave-one-region-out unit (a cross-validation fold: it trains on at least two
// regions) answers with the size of its pool's SMALLEST training set, T - ceil(T / k) for T = training + test rows - the same number
// for every fold of the pool, so whether a (term, fold) takes the pass does not depend on which folds of the pool share the call (fold
// sizes differ by a row: a call holding one fold and a call holding all ten must give that fold the same double).  Any other unit: its own.
static int64_t moment_rule_rows(const GUnit& U) {
    const int k = __builtin_popcountll(U.train_mask) + 1;
    if (k < 3) return U.N;
    const int64_t T = (int64_t)U.N + U.nq;
    return T - (T + k - 1) / k;
}

// one chunk: pools [p0, p1) of the (variant-sorted) order; all of one variant (same KS, fold / wmul)
void run_chunk(pbn_ctx* ctx, const pbn_table* t, GroupBatch& b, const std::vector<int>& order, size_t p0, size_t p1, double* dev_out,
               double* dev_out_max, bool force_f64) {
    const int np = (int)(p1 - p0);
    // ---- layout ----------------------------------------------------------------------------------------------------------
    std::vector<GPool> pools(np);
    std::vector<GUnit> units;
    int64_t E = 0;
    int B = 0, Rs = 1, xstride = 1, max_units = 1;
    for (int k = 0; k < np; ++k) {
        GPool P = b.pools[order[p0 + k]];
        P.local = k;
        P.elem0 = E;
        P.nblk = (P.n + GB - 1) / GB;
        P.blk0 = B;
        const int u0 = P.unit0;
        P.unit0 = (int)units.size();
        for (int u = 0; u < P.nunits; ++u) {
            GUnit U = b.units[u0 + u];
            U.pool = k;
            units.push_back(U);
        }
        E += P.n;
        B += P.nblk + 1;   // + the padding block
        Rs = std::max(Rs, P.R);
        xstride = std::max(xstride, P.d);
        max_units = std::max(max_units, P.nunits);
        pools[k] = P;
    }
    const int nu = (int)units.size();
    const int fdt = force_f64 ? PBN_F64 : t->dtype;   // type of the fragments and of the sweep (fp64 on float columns for widened units)
    const bool f16 = use_f16x2(fdt);
    const int d0 = pools[0].d, KS = f16 ? f16x2_mfmas(d0) : (d0 + 3) / 4;
    const bool fold = d0 % 4 != 0;
    const size_t frag_b = f16 ? (size_t)KS * 64 * 16 : (size_t)KS * 64 * 8;   // bytes of a 16-row tile's fragments
    // Moment pass (round 5): units of one or two dimensions on fp64 fragments - every term here is a sum (EF32 sweeps) - whose training
    // sets are dense enough for it to pay (moment_pass_rows; kde_group_run gives such sets chunks of their own).  PBN_MOMENT_PASS=0
    // switches it off (the sweep then takes every pair, as until round 4).
    bool dense = true;
    for (const GUnit& U : units) dense = dense && moment_rule_rows(U) >= moment_pass_rows();
    const bool moments = !f16 && d0 <= 2 && dense && knob_int("PBN_MOMENT_PASS", 1) != 0 && PBN_TUNE(PRUNE_GROUP_MASKS, 1) != 0;
    const bool bboxes = !f16 && PBN_TUNE(GROUP_BATCH_BOXES, 1) != 0;   // fp64 sweeps with per-group masks: one uniform test per (batch, group) first
    int64_t total_wg = 0;
    int max_ntiles = 0, max_nqtiles = 0, max_nq = 0, max_nbatch = 1;
    size_t off = 0;
    auto carve = [&](size_t bytes) { const size_t o = off; off += al256(bytes); return (int64_t)o; };
    // tables first
    const int64_t o_pools = carve((size_t)np * sizeof(GPool)), o_units = carve((size_t)nu * sizeof(GUnit)),
                  o_sweep = carve((size_t)nu * sizeof(GSweepUnit)), o_blkpool = carve((size_t)B * sizeof(int32_t));
    for (GUnit& U : units) {
        const GPool& P = pools[U.pool];
        const int d = P.d, pd = P.kd;
        U.ntiles = (U.N + 15) / 16;
        U.nqtiles = (U.nq + 15) / 16;
        const int split_tiles = split_tiles_for(U.ntiles, !f16);
        const int nsplit0 = std::max(1, (U.ntiles + split_tiles - 1) / split_tiles);
        U.tps = (U.ntiles + nsplit0 - 1) / nsplit0;
        U.nsplit = (U.ntiles + U.tps - 1) / U.tps;
        const int qg = sweep_qg(fdt, /*cond=*/false, KS, /*prune=*/true);   // query groups per wave of the chunk's sweep kernel: the ONE place that knows
                                                                             // (a count from the other type's constant ran every fp32 workgroup twice - same sums, twice the time)
        const int qblocks = (U.nqtiles + qg - 1) / qg;
        U.nwg = qblocks * U.nsplit;
        U.wg0 = total_wg;
        total_wg += ((int64_t)U.nwg + 63) / 64 * 64;
        max_ntiles = std::max(max_ntiles, U.ntiles);
        max_nqtiles = std::max(max_nqtiles, U.nqtiles);
        max_nq = std::max(max_nq, U.nq);
        max_nbatch = std::max(max_nbatch, U.nsplit * ((U.tps + 63) / 64));
        U.apack = carve((size_t)U.ntiles * frag_b);
        U.npack = carve(f16 ? 256 : (size_t)U.ntiles * 16 * 8 * 2);
        U.zs = carve((size_t)U.N * d * 8 + 8);
        U.box = carve((size_t)U.ntiles * 2 * pd * 8);
        U.bpack = carve((size_t)U.nqtiles * frag_b);
        U.ny = carve((size_t)U.nqtiles * 16 * (f16 ? 4 : 8));
        U.zq = carve((size_t)U.nq * d * 8 + 8);
        U.qpos = carve((size_t)U.nqtiles * 16 * 4);
        U.qbox = carve((size_t)U.nqtiles * 2 * pd * 8);
        U.qthr = carve((size_t)U.nqtiles * 8);
        U.qlb = carve((size_t)U.nqtiles * 16 * 8);
        U.nsplit_fin = moments ? 2 * U.nsplit : U.nsplit;   // the moment pass's partials behind the sweep's
        U.part = carve((size_t)U.nsplit_fin * U.nqtiles * 16 * 2 * 8);
        U.bbox = bboxes ? carve((size_t)U.nsplit * ((U.tps + 63) / 64) * 2 * pd * 8) : 0;
        U.mom = moments ? carve((size_t)((U.ntiles + 63) / 64 * 64) * pbn_mom_rec(d) * 8) : 0;
        U.rad2 = moments ? carve((size_t)U.ntiles * 4) : 0;
        U.bsum = carve((size_t)((U.nq + 255) / 256 + 1) * 8);
    }
    const int64_t o_wgunit = carve((size_t)(total_wg / 64 + 1) * sizeof(int32_t));
    const size_t table_end = off;
    (void)table_end;
    const int64_t o_keys_a = carve((size_t)E * 4), o_keys_b = carve((size_t)E * 4), o_vals_a = carve((size_t)E * 4), o_vals_b = carve((size_t)E * 4),
                  o_xs = carve((size_t)E * xstride * 8), o_reg = carve((size_t)E), o_blkcnt = carve((size_t)B * Rs * 4);
    if (off + 256 > ctx->scratch_group.n)   // grow in few, large steps (a hipFree + hipMalloc of gigabytes each)
        ctx->scratch_group.reserve(std::max(off + 256, std::min(ctx->scratch_group.n * 2, kde_group_arena_budget() + (64u << 20))));
    char* arena = ctx->scratch_group.p;

    // ---- host tables -> one upload ---------------------------------------------------------------------------------------
    // (the per-unit buffers were carved between the tables; the upload covers pools | units | sweep records | blkpool, then the
    //  workgroup table separately)
    std::vector<char> h((size_t)o_blkpool + al256((size_t)B * 4));
    std::memcpy(h.data() + o_pools, pools.data(), (size_t)np * sizeof(GPool));
    std::memcpy(h.data() + o_units, units.data(), (size_t)nu * sizeof(GUnit));
    GSweepUnit* hs = (GSweepUnit*)(h.data() + o_sweep);
    const bool guard = knob_int("PBN_MAGIC_GUARD", 1) != 0;   // 0: exp2_magic keeps its clamp everywhere (kde_model.hip)
    for (int u = 0; u < nu; ++u) {
        const GUnit& U = units[u];
        GSweepUnit& s = hs[u];
        s.Apack = arena + U.apack; s.nxpack = arena + U.npack; s.Bpack = arena + U.bpack; s.nypack = arena + U.ny;
        s.tile_box = (const double*)(arena + U.box); s.qtile_box = (const double*)(arena + U.qbox);
        s.qtile_thr = (const double*)(arena + U.qthr); s.qlb = (const double*)(arena + U.qlb);
        s.part = (double*)(arena + U.part); s.wg0 = U.wg0;
        s.batch_box = bboxes ? (const double*)(arena + U.bbox) : nullptr; s.nbps = (U.tps + 63) / 64;
        s.tile_rad2 = moments ? (const float*)(arena + U.rad2) : nullptr;
        s.tile_mom = moments ? (const double*)(arena + U.mom) : nullptr;
        s.zq = (const double*)(arena + U.zq);
        s.part_mom = moments ? (double*)(arena + U.part) + (size_t)U.nsplit * U.nqtiles * 16 * 2 : nullptr;
        s.ntiles = U.ntiles; s.nqtiles = U.nqtiles; s.tps = U.tps; s.nsplit = U.nsplit; s.nwg = U.nwg; s.pdims = pools[U.pool].kd; s.box_full = (guard && pools[U.pool].kd == pools[U.pool].d) ? 1 : 0;
        s.margin = (float)prune_margin(fdt, U.N, /*the engine's terms are sums*/ true); s.mom_stride = (U.ntiles + 63) / 64 * 64;
    }
    int32_t* hb = (int32_t*)(h.data() + o_blkpool);
    for (int k = 0; k < np; ++k)
        for (int j = 0; j <= pools[k].nblk; ++j) hb[pools[k].blk0 + j] = k;
    std::vector<int32_t> hw((size_t)(total_wg / 64 + 1));
    for (int u = 0; u < nu; ++u) {
        const int64_t w0 = units[u].wg0 / 64, w1 = w0 + ((int64_t)units[u].nwg + 63) / 64;
        for (int64_t w = w0; w < w1; ++w) hw[(size_t)w] = u;
    }
    hipStream_t st = ctx->stream;
    // the tables sit at the head of the arena (the units' buffers were carved behind o_blkpool); the host copies stay alive on the
    // context until the caller has synchronised (pbn_ctx::drop_staged)
    HIP_CHECK(hipMemcpyAsync(arena, h.data(), h.size(), hipMemcpyHostToDevice, st));
    std::vector<char> hwb((const char*)hw.data(), (const char*)hw.data() + hw.size() * sizeof(int32_t));
    HIP_CHECK(hipMemcpyAsync(arena + o_wgunit, hwb.data(), hwb.size(), hipMemcpyHostToDevice, st));
    ctx->staged.push_back(std::move(h));
    ctx->staged.push_back(std::move(hwb));

    GDev g{};
    g.pools = (const GPool*)(arena + o_pools); g.units = (const GUnit*)(arena + o_units); g.blkpool = (const int32_t*)(arena + o_blkpool);
    g.base = t->data; g.ld = t->ld;
    g.keys = (uint32_t*)(arena + o_keys_a); g.vals = (uint32_t*)(arena + o_vals_a);
    g.xs = (double*)(arena + o_xs); g.reg = (uint8_t*)(arena + o_reg); g.blkcnt = (int32_t*)(arena + o_blkcnt);
    g.arena = arena; g.xstride = xstride; g.Rs = Rs; g.nunits = nu;
    static const int sum_bound = PBN_TUNE(GROUP_SUM_BOUND, 1);
    g.use_sum_bound = sum_bound;
    g.f16 = f16 ? 1 : 0; g.KS = KS;
    g.out_max = f16 ? (unsigned long long*)dev_out_max : nullptr;
    static const int tile_window = std::max(0, PBN_TUNE(GROUP_TILE_WINDOW, 256));
    g.tile_window = tile_window;
    static const int fine_keys = PBN_TUNE(GROUP_FINE_KEYS, 1);
    g.fine_keys = fine_keys;
    static const int hilbert = PBN_TUNE(GROUP_HILBERT, 2);   // 1: two key dimensions only, 2: three and four as well
    g.hilbert = hilbert;
    static const int window = std::max(1, PBN_TUNE(GROUP_WINDOW, PBN_GROUP_WINDOW));
    g.window = window;

    const bool f64 = t->dtype == PBN_F64;
    {
        KernelTimer kt(ctx, PBN_K_PACK);
        if (f64) hipLaunchKernelGGL(group_keys_kernel<double>, dim3((unsigned)B), dim3(GB), 0, st, g);
        else hipLaunchKernelGGL(group_keys_kernel<float>, dim3((unsigned)B), dim3(GB), 0, st, g);
        HIP_CHECK(hipGetLastError());
        int pool_bits = 0;
        while ((1 << pool_bits) < np) ++pool_bits;
        sort_keys(ctx->scratch_sort, (const uint32_t*)(arena + o_keys_a), (uint32_t*)(arena + o_keys_b), (const int32_t*)(arena + o_vals_a),
                  (int32_t*)(arena + o_vals_b), E, MORTON_BITS + pool_bits, st);
        g.keys = (uint32_t*)(arena + o_keys_b); g.vals = (uint32_t*)(arena + o_vals_b);
        if (f64) hipLaunchKernelGGL(group_gather_kernel<double>, dim3((unsigned)B), dim3(GB), 0, st, g);
        else hipLaunchKernelGGL(group_gather_kernel<float>, dim3((unsigned)B), dim3(GB), 0, st, g);
        hipLaunchKernelGGL(group_scan_kernel, dim3((unsigned)Rs, (unsigned)np), dim3(64), 0, st, g);
        hipLaunchKernelGGL(group_pack_train_kernel, dim3((unsigned)B, (unsigned)max_units), dim3(GB), 0, st, g);
        hipLaunchKernelGGL(group_pack_query_kernel, dim3((unsigned)B), dim3(GB), 0, st, g);
        hipLaunchKernelGGL(group_tile_box_kernel, dim3((unsigned)((max_ntiles * 16 + 255) / 256), (unsigned)nu), dim3(256), 0, st, g);
        if (bboxes) hipLaunchKernelGGL(group_batch_box_kernel, dim3((unsigned)max_nbatch, (unsigned)nu), dim3(64), 0, st, g);
        if (moments) {
            if (d0 == 1) hipLaunchKernelGGL(group_tile_moments_kernel<1>, dim3((unsigned)((max_ntiles * 16 + 255) / 256), (unsigned)nu), dim3(256), 0, st, g);
            else hipLaunchKernelGGL(group_tile_moments_kernel<2>, dim3((unsigned)((max_ntiles * 16 + 255) / 256), (unsigned)nu), dim3(256), 0, st, g);
        }
        hipLaunchKernelGGL(group_prepass_kernel, dim3((unsigned)((max_nqtiles * 16 + 255) / 256), (unsigned)nu), dim3(256), 0, st, g);
        HIP_CHECK(hipGetLastError());
    }
    {
        KernelTimer kt(ctx, PBN_K_SWEEP);
        GSweepArgs sa{};
        sa.units = (const GSweepUnit*)(arena + o_sweep); sa.wg_unit = (const int32_t*)(arena + o_wgunit); sa.total_wg = total_wg;
        sa.fold = fold ? 1 : 0; sa.wmul = fold ? 0 : 1; sa.count_redo = knob_int("PBN_SWEEP_COUNT_REDO", 0);
        sa.prune_margin = 0.0;   // the units' own margins
        static const int gmasks = PBN_TUNE(PRUNE_GROUP_MASKS, 1);
        sa.group_masks = gmasks;
        sa.far_span = (double)knob_int("PBN_FAR_SPAN", 17);   // fp64 FOLD shapes: far tiles through the fp32 unit (kde_sweep_body: FARP)
        static const int log_chunks = PBN_TUNE(SWEEP_LOG, 0);   // experiments build: one line per grouped sweep launch (synchronises)
        hipEvent_t le0 = nullptr, le1 = nullptr;
        if (log_chunks) { HIP_CHECK(hipEventCreate(&le0)); HIP_CHECK(hipEventCreate(&le1)); HIP_CHECK(hipEventRecord(le0, st)); }
        sa.moments = moments ? 1 : 0;
        sa.w32 = (f16 && f16x2_w32p(d0, KS)) ? 1 : 0;
        launch_sweep_grouped(sa, fdt, KS, st);
        if (moments) { KernelTimer km(ctx, PBN_K_MOMENT); launch_moment_grouped(sa, d0, st); }
        if (log_chunks) {
            HIP_CHECK(hipEventRecord(le1, st));
            HIP_CHECK(hipEventSynchronize(le1));
            float ms = 0.f;
            HIP_CHECK(hipEventElapsedTime(&ms, le0, le1));
            int64_t pairs = 0;
            for (const GUnit& U : units) pairs += (int64_t)U.N * U.nq;
            std::fprintf(stderr, "pbn-group-sweep d=%d %s pools=%d units=%d pairs=%lld ms=%.3f\n", d0, f16 ? "f32" : "f64", np, nu, (long long)pairs, ms);
            (void)hipEventDestroy(le0); (void)hipEventDestroy(le1);
        }
    }
    {
        KernelTimer kt(ctx, PBN_K_FINISH);
        hipLaunchKernelGGL(group_finish_kernel, dim3((unsigned)((max_nq + 255) / 256), (unsigned)nu), dim3(256), 0, st, g);
        hipLaunchKernelGGL(group_reduce_kernel, dim3((unsigned)nu), dim3(256), 0, st, g, dev_out);
        HIP_CHECK(hipGetLastError());
    }
}

// bytes of arena a pool needs (its element arrays + its units' buffers), for chunking
size_t pool_bytes(const GroupBatch& b, const GPool& P) {
    const int d = P.d, KS = (d + 3) / 4, pd = P.kd;
    size_t s = (size_t)P.n * (16 + (size_t)PBN_GROUP_MAX_D * 8 + 1) + (size_t)((P.n + GB - 1) / GB + 1) * (PBN_GROUP_MAX_R * 4 + 4) + sizeof(GPool) + 4096;
    for (int u = 0; u < P.nunits; ++u) {
        const GUnit& U = b.units[P.unit0 + u];
        const size_t nt = (U.N + 15) / 16, nqt = (U.nq + 15) / 16, split_tiles = (size_t)split_tiles_for((int)nt, /*the finer rule: an upper bound for both*/ false), nsplit = std::max<size_t>(1, (nt + split_tiles - 1) / split_tiles);
        s += nt * KS * 1024 + nt * 256 + (nt + nqt) * 2048 + nqt * 64 + (size_t)U.N * d * 8 + nt * 2 * pd * 8 + nqt * KS * 1024 + nqt * 128 + (size_t)U.nq * d * 8 + nqt * 64 +
             nqt * 2 * pd * 8 + nqt * 8 + nqt * 128 + 2 * (nsplit + 1) * nqt * 256 * (d <= 2 ? 2 : 1) + (d <= 2 ? (nt + 64) * (pbn_mom_rec(d <= 1 ? 1 : 2) * 8 + 4) + 512 : 0) +
             (size_t)U.nq / 32 + sizeof(GUnit) + sizeof(GSweepUnit) + (nsplit + 1) * ((nt / nsplit + 127) / 64) * 2 * pd * 8 + 256 +
             (nqt / 2 + 1) * nsplit / 16 + 13 * 256 + 64;   // (workgroup table: one entry per 64 workgroups of nqt / PBN_QG_PRUNE x nsplit)
    }
    return s;
}

}  // namespace

// 8 GB of the 288: a hill-climb's delta-cache update hands over ~10 hybrid candidates at a time, ~1.6 GB of arena each at 1M rows (every row is
// whitened once per unit it trains, every query keeps a partial per training split).  C5 alone (tools/hybrid_batch_probe.sh): 4 GB 7.89 s,
// 8 GB 7.77 s, 16 GB 7.73 s, 32 GB 7.65 s - but the hipMalloc of a 16 GB arena in a process that has used the memory before costs 0.4 s
// (tools/arena_bench_probe.sh: the bench line's 0.62 s cv_weak leg became 1.00 s), so the default stops at 8
size_t kde_group_arena_budget() { return (size_t)std::max(64, knob_int("PBN_GROUP_ARENA_MB", 8192)) << 20; }

size_t kde_group_pool_bytes(const GroupBatch& b, const GPool& P) { return pool_bytes(b, P); }

bool kde_group_applies(int dtype, int d, int64_t n_min, int R) {
    const int on = knob_int("PBN_SCORE_GROUPED", 1);   // read per call: the tests switch it
    // fp64 classic fragments; sets whose boxes cover every dimension (no subsample bound needed); the pruned-sweep shapes
    return on && (dtype == PBN_F64 || use_f16x2(dtype)) && d >= 1 && d <= std::min(PBN_TUNE(PRUNE_BOX_DIMS, 4), 4) && R >= 1 &&
           R <= PBN_GROUP_MAX_R && kde_prune_applies(dtype, d, n_min);
}

void kde_group_run(pbn_ctx* ctx, const pbn_table* t, GroupBatch& b, double* dev_out_sums, double* dev_out_max, bool force_f64) {
    if (b.pools.empty()) return;
    if (!force_f64 && t->dtype != PBN_F64 && !use_f16x2(t->dtype)) throw invalid_error("grouped KDE evaluation: fp64 tables, or fp32 tables on the f16 matrix cores");
    HIP_CHECK(hipSetDevice(ctx->device));
    for (const GPool& P : b.pools) {
        if (P.d < 1 || P.d > PBN_GROUP_MAX_D || P.kd < 1 || P.kd > PBN_PRUNE_PD || P.kd > P.d || P.R < 1 || P.R > PBN_GROUP_MAX_R || P.n < 1)
            throw invalid_error("grouped KDE evaluation: bad pool");
        for (int u = 0; u < P.nunits; ++u) {
            const GUnit& U = b.units[P.unit0 + u];
            if (U.N < 1 || U.nq < 1) throw invalid_error("grouped KDE evaluation: empty unit");
        }
    }
    // pools of one sweep shape together (KS, norm in a K slot or as weights), larger sets first
    std::vector<int> order(b.pools.size());
    for (size_t i = 0; i < order.size(); ++i) order[i] = (int)i;
    const bool f16 = !force_f64 && use_f16x2(t->dtype);
    // (fp64 sets of one and of two variables get chunks of their own: their units take the moment pass, whose records and kernel depend on d)
    // (... and only when every training set of the pool is dense enough: moment_pass_rows)
    const int mom_rows = moment_pass_rows();
    auto dense = [&](const GPool& P) {
        for (int u = 0; u < P.nunits; ++u)
            if (moment_rule_rows(b.units[P.unit0 + u]) < mom_rows) return false;
        return true;
    };
    auto variant = [&](int i) {
        const int d = b.pools[i].d;
        return f16 ? f16x2_mfmas(d) * 2 : ((d + 3) / 4) * 2 + (d % 4 == 0 ? 1 : 0) + (d <= 2 ? 16 * d + (dense(b.pools[i]) ? 64 : 0) : 0);
    };
    std::stable_sort(order.begin(), order.end(), [&](int x, int y) { return variant(x) < variant(y); });
    const size_t budget = kde_group_arena_budget();
    const int max_pools = std::min(256, std::max(1, PBN_TUNE(GROUP_MAX_POOLS, 256)));
    size_t p0 = 0;
    while (p0 < order.size()) {
        size_t p1 = p0, bytes = 0;
        int64_t elems = 0;
        while (p1 < order.size() && variant(order[p1]) == variant(order[p0]) && (int)(p1 - p0) < max_pools) {
            const size_t pb = pool_bytes(b, b.pools[order[p1]]);
            if (p1 > p0 && (bytes + pb > budget || elems + b.pools[order[p1]].n > 0x7ff00000ll)) break;
            bytes += pb;
            elems += b.pools[order[p1]].n;
            ++p1;
        }
        run_chunk(ctx, t, b, order, p0, p1, dev_out_sums, dev_out_max, force_f64);
        p0 = p1;
    }
}

}  // namespace pbn
