// Kernel argument blocks + launchers shared between kde_kernels.hip and the C-ABI glue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "common.hpp"

#define PBN_MAX_D 33       // up to 32 whitened "main" dimensions (fp64: KS <= 8; fp32: 16) + 1 CKDE extra coordinate
#define PBN_W_INLINE_D 17  // whitening matrices up to this order travel inside the kernel arguments, larger ones through device memory

#define PBN_PRUNE_PD 5     // most whitened dimensions the Morton keys and the boxes of the pruned sweeps cover (KdeModel::pdims <= this)

namespace pbn {

// bits per dimension of a 32-bit Morton key over kd dimensions: 10 up to three, 8 for four, 6 for five
// Morton keys of the stand-alone pruned sweeps: bits per key dimension and the edge of a key cell in (base-2) whitened units.  Round 5:
// one or two key dimensions take 16 bits per axis in cells of 1/32 unit (the same +-1024 units of range as the 10 bits x 2.0 units before),
// three and four dimensions cells of 0.5 (+-256 / +-64 units), five of 1.0 (1e6 x 1e5 handles d = 1 ... 5: 10.2 / 7.2 / 7.9 / 11.8 / 20.0 ->
// 8.5 / 5.4 / 6.7 / 11.5 / 19.0 ms, profiles/r5/prune_visits.txt): with 2.0-unit cells (1.7 bandwidths) a cell of a 10^6-row set held hundreds of rows in
// arbitrary order, and a 16-row tile was 16 random rows of it (the grouped evaluation had the same flaw: kde_group.hip group_keys_kernel)
__host__ __device__ inline int prune_key_bits(int kd) { return kd <= 2 ? 16 : (kd == 3 ? 10 : 32 / kd); }
inline double prune_key_cell(int kd) { return kd <= 2 ? 0.03125 : (kd <= 4 ? 0.5 : (kd == 5 ? 1.0 : 2.0)); }   // 4 / 5 dimensions: 8 / 6 bits, +-64 / +-32 units

struct PackArgs {
    const void* base;     // device column-major table
    int64_t ld;           // elements between columns
    int cols[PBN_MAX_D];  // selected columns, in whitening order
    int d;                // number of selected columns
    int dm;               // main (marginal) dimensions written to `pack`; d == dm or dm + 1
    int KS;               // ceil(dm / 4)
    int is_query;
    // the table holds float while the fragments are packed (and swept) in double: fp32 models whose whitened training rows reach so
    // far from the centre that the Gram-form distances would lose them in fp32 (KdeModel::widen); classic fp64 pack only
    int src_f32;
    // classic (non-f16x2) pack, dm % 4 != 0: the first unused K slot of the last MFMA carries the norm - training side
    // -1/2|z|^2, query side 1 - so the sweep's accumulator starts from a per-query constant (no add per value).  A
    // query pack that leaves the slot 0 (CKDE::cdf, UCV) makes the training side's entry inert.
    int fold_norm;
    int write_w;          // training side, classic pack: also write the weights 2^norm at npack + ntiles * 16 (WMUL sweeps)
    int write_r;          // ... and per tile sqrt(max -norm) of its rows (+inf with a padding row) as doubles behind the weights (fp64 fragments;
                          // SweepArgs::tile_r: the guard of the unclamped 2^x of the unpruned sum-only sweeps)
    // source rows: logical row r maps to  l = r < n0 ? row0 + r : row1 + (r - n0)   (two contiguous ranges: a CV training
    // set is "everything before the fold" ++ "everything after it"), and then through the gather list, rows[l], when one is
    // given (hybrid slices: the rows of one discrete configuration, grouped fold by fold - the same two-range trick inside it)
    int64_t row0, n0, row1;
    const int32_t* rows;  // device gather list (nullable)
    const int32_t* perm;  // nullable: logical row r is taken from logical row perm[r] (spatially sorted packs, see SweepArgs::prune)
    int64_t perm_stride;  // > 1: from logical row perm[r * perm_stride] (the stratified subsample of a sorted pack)
    int64_t n;            // valid rows
    int64_t ntiles;       // ceil(n / 16)
    double W[PBN_W_INLINE_D * PBN_W_INLINE_D];  // d x d row-major lower-triangular whitening matrix (kernel argument), d <= PBN_W_INLINE_D
    const double* Wdev;               // the same matrix in device memory when d > PBN_W_INLINE_D (else null)
    double mu[PBN_MAX_D];             // d centring offsets
    void* pack;           // [ntiles][KS][64]
    void* npack;          // [ntiles][16]
    void* xpack;          // [ntiles][64] or null
    void* xnorm;          // f16x2 query side only: [ntiles][16] f32 base of the rewritable CKDE slots
    // CKDE::cdf: u = sum_j wu[j] * (x_j - mu_j) over the d selected columns -> upack (npack layout); classic pack only
    double wu[PBN_MAX_D];
    void* upack;          // nullable
    unsigned char* far_flag;  // nullable; f16x2 query pack: [16 ntiles] 1 for a row with a whitened coordinate beyond the f16 range (clamped in the fragments; kde_far_fix_kernel)
};


// ---- f16x2 fragments of fp32 tables (kde_kernels.hip, "fp32 path on the 16-bit matrix cores"): device helpers shared by the pack kernels ----
typedef _Float16 hf8 __attribute__((ext_vector_type(8)));   // (eight 16-bit pieces of a fragment lane)
typedef _Float16 hpiece;
#define PBN_H_C1 32768.0f      // 2^15
#define PBN_H_C2 32.0f         // 2^5
#define PBN_H_C3 0.015625f     // 2^-6
#define PBN_H_LO 64.0f         // scale of the low coordinate piece (its partner carries 2^-6)
#define PBN_H_MAX 65504.0f

__device__ __forceinline__ hpiece h_piece(float x) {   // f16(x), never subnormal
    const hpiece h = (hpiece)x;
    return __builtin_fabsf((float)h) < 0x1p-14f ? (hpiece)0.0f : h;
}
// z -> (a1, a2 2^6) and the represented value a1 + a2
__device__ __forceinline__ double split2(double z, hpiece& hi, hpiece& lo, bool& clamped) {
    clamped = !(__builtin_fabs(z) <= (double)PBN_H_MAX);
    z = z > (double)PBN_H_MAX ? (double)PBN_H_MAX : (z < -(double)PBN_H_MAX ? -(double)PBN_H_MAX : z);
    if (!(z == z)) z = 0.0;
    hi = h_piece((float)z);
    const double r = z - (double)(float)hi;
    lo = h_piece((float)(r * (double)PBN_H_LO));
    return (double)(float)hi + (double)(float)lo * (1.0 / (double)PBN_H_LO);
}
// x = 2^15 p1 + 2^5 p2 + 2^-6 p3 (|x| clamped to 65504 x 2^15 = 2.1e9: the padding norm -1e30 becomes that, far below every real exponent)
__device__ __forceinline__ void split3s(float x, hpiece& p1, hpiece& p2, hpiece& p3) {
    constexpr float XMAX = PBN_H_MAX * PBN_H_C1;
    x = x > XMAX ? XMAX : (x < -XMAX ? -XMAX : x);
    if (!(x == x)) x = -XMAX;
    p1 = h_piece(x * (1.0f / PBN_H_C1));
    float r = x - (float)p1 * PBN_H_C1;
    p2 = h_piece(r * (1.0f / PBN_H_C2));
    r -= (float)p2 * PBN_H_C2;
    p3 = h_piece(r * (1.0f / PBN_H_C3));
}

// Slots per dimension: the fourth product a2 b2 rides along where the 32-slot blocks the three-product layout needs have room for it
// (1...7, 10...15, 21...23 dimensions): then nothing is dropped and what is left is the input perturbation and the f32 accumulation,
// 2^-24 |z|^2 on an exponent as for the f16x2 fragments of rounds 1-5; with three products (8, 9, 16...20, ... dimensions) <= 2^-22 |z|^2.
__host__ __device__ inline int f16x2_blocks(int dm) { return (3 * dm + 3 + 31) / 32; }
__host__ __device__ inline int f16x2_spd(int dm) { return 4 * dm + 3 <= 32 * f16x2_blocks(dm) ? 4 : 3; }
// f16x2 fragments of one row: slot s = spd k + role for dimension k, then the three pieces of the training norm against the constants; the last
// three slots of the contraction carry the constants on the training side when they are free (the W32 sweep's query offsets);
// slot s lives in MFMA s / 32, lane group (s % 32) / 8, element s % 8.  Roles: training (a1, a1 2^-6, a2 2^6, a2), query (b1, b2 2^6, b1 2^-6, b2).
__device__ __forceinline__ void f16x2_store_row(hf8* pack, int NB, int64_t tile, int idx, int dm, const hpiece* p1, const hpiece* p2, float nv, bool query) {
    hpiece n1, n2, n3;
    split3s(nv, n1, n2, n3);
    const int spd = f16x2_spd(dm);
    const hpiece zero = (hpiece)0.0f, c1 = (hpiece)PBN_H_C1, c2 = (hpiece)PBN_H_C2, c3 = (hpiece)PBN_H_C3;
    auto konst = [&](int t) { return t == 0 ? c1 : (t == 1 ? c2 : c3); };
    auto slot = [&](int s) -> hpiece {
        if (s < spd * dm) {
            const int k = s / spd, role = s % spd;
            const hpiece hi_s = h_piece((float)p1[k] * (1.0f / PBN_H_LO)), lo_u = h_piece((float)p2[k] * (1.0f / PBN_H_LO));
            if (role == 0) return p1[k];
            if (role == 3) return lo_u;
            if (!query) return role == 1 ? hi_s : p2[k];
            return role == 1 ? p2[k] : hi_s;
        }
        const int t = s - spd * dm;
        if (t < 3) return query ? konst(t) : (t == 0 ? n1 : (t == 1 ? n2 : n3));
        if (!query && s >= 32 * NB - 3 && spd * dm + 6 <= 32 * NB) return konst(s - (32 * NB - 3));
        return zero;
    };
    for (int mb = 0; mb < NB; ++mb)
        for (int g = 0; g < 4; ++g) {
            hf8 v;
#pragma unroll
            for (int j = 0; j < 8; ++j) v[j] = slot(mb * 32 + g * 8 + j);
            pack[(tile * NB + mb) * 64 + g * 16 + idx] = v;
        }
}

struct SweepArgs {
    const void* Apack;
    const void* nxpack;
    const void* Axpack;
    const void* Bpack;
    const void* nypack;
    const void* Bxpack;
    const void* Bxnorm;  // f16x2 CKDE: f32 [nqtiles][16]
    int64_t ntiles;
    int64_t nqtiles;
    int64_t tiles_per_split;
    int fold;      // the packs carry the training norms in a free K slot (PackArgs::fold_norm)
    int count_redo; // measurement aid (PBN_SWEEP_COUNT_REDO): count the units of the unchecked pass that redo their split
    int fast;      // fp64 plain sweeps whose result is a SUM over the test rows: 2^f on the fp32 transcendental unit (kde_kernels.hip: exp2_f64_fract<true>)
    int wmul;      // fp64 plain sweeps with d mod 4 == 0: training norms as weights 2^norm behind the norms (PackArgs::write_w)
    int w32;       // fp32 plain unpruned sweeps whose contraction leaves three slots free: the 32x32x16 f16 form (kde_sweep_f16_w32_kernel; f16x2_w32)
    // Tile pruning (low-dimensional fp64 sweeps of the score engine): both sides are packed in Morton order of their
    // whitened coordinates, every 16-row training tile and every 16-row query tile has a bounding box over the first
    // `pdims` whitened dimensions, and qtile_thr holds, per query tile, a lower bound of its queries' largest exponents
    // (from a scan of the training rows next to them in Morton order).  A wave skips a training tile whose box is so far
    // from the box of its queries that every exponent is below that bound - prune_margin: such terms are < 2^-margin of their sums
    // each, i.e. at most N 2^-margin of a sum in total; the margin is 52 (fp64) / 40 (fp32) at 10^6 training rows and follows
    // log2(N / 10^6), so that bound is 2.2e-10 (9.1e-7) of a sum for every N (kde_kernels.hip: prune_margin).
    int prune;
    int pdims;
    double prune_margin;       // base-2 exponent distance below the queries' bound beyond which a tile is skipped (52 / 40)
    const double* tile_box;    // [ntiles][2 * pdims]: lo..., hi...
    const double* qtile_box;   // [nqtiles][2 * pdims]
    const double* qtile_thr;   // [nqtiles]
    const double* qlb;         // [nqtiles * 16] per (sorted) query: lower bound of its largest exponent, -inf = none; nullable
    int nsplit_grid;           // pruned sweeps: number of splits (their grid is one-dimensional; launch_sweep sets this)
    int group_masks;           // pruned plain fp64 sweeps: test every 16-query group against its own box and bound (prune_group_mask)
    double far_span;           // pruned plain fp64 sum-only sweeps (FOLD shapes): > 0 = tiles whose every term lies more than prune_margin - far_span
                               // below the group's sum bound take the fp32 tail path (kde_sweep_body: FARP); 0 = off
    // Tile moments (round 5; grouped sum-only fp64 sweeps of one or two dimensions, kde_group.hip): tile_rad2[t] = squared radius of tile t
    // about its centroid (rounded up), tile_mom = per tile a record of PBN_MOM_REC(d) doubles: centroid, then the coefficients of the
    // order-PBN_MOM_ORDER expansion of the tile's contribution about it.  A (tile, group) pair whose expansion error is below the pruning
    // bound is left to the moment pass (kde_moment_group_kernel) by the sweep; null = off.
    const float* tile_rad2;
    const double* tile_mom;
    // Unpruned sum-only fp64 sweeps (round 6): tile_r[t] = sqrt(max -norm) over the rows of training tile t (PackArgs::write_r) - a chunk of
    // tiles whose largest radius proves every exponent of the wave's queries inside +-1022 takes exp2_magic without its clamp; null = clamp always
    const double* tile_r;
    int box_full;              // pruned sweeps: the boxes cover ALL whitened dimensions (pdims == the model's) - the proof behind prune_open_mask
    // Batch boxes (round 5, pruned fp64 plain sweeps): the bounding box of every 64-tile batch of a split - [split * batches_per_split + k][2 * pdims].
    // A wave classifies 64 BATCHES with one ballot (lane = batch) before it loads and tests the tile boxes of the batches in reach; null = off.
    const double* batch_box;
    int batches_per_split;
    // (Measured and dropped, round 5: a bitmap written by the moment pass - the batches that still hold pairs for the sweep, 24 % of the
    //  (batch, group) masks at 300 000 rows - instead of the sweep's own batch test: sweep 2.03 -> 1.90 s, moment kernel 4.93 -> 5.04 s on C3's first iteration.)
    double* part;  // [nsplit][nqtiles*16][P]
};

// Position along the Hilbert curve of a cell in n = 2 ... 4 dimensions, `bits` bits per axis (Skilling's transpose form: undo the excess rotations
// from the top bit down, Gray-encode, interleave with axis 0 most significant) - consecutive keys are neighbouring cells, so 16 consecutive rows of
// the sorted order form a compact tile where the Z-order jumps at every power-of-two boundary.  Used for the pruned sweeps' row order only:
// nothing numeric depends on it.  X is overwritten.
__host__ __device__ inline uint32_t hilbert_key(uint32_t* X, int n, int bits) {
    const uint32_t M = 1u << (bits - 1);
    for (uint32_t Q = M; Q > 1; Q >>= 1) {
        const uint32_t P = Q - 1;
        for (int i = 0; i < n; ++i) {
            if (X[i] & Q) X[0] ^= P;
            else { const uint32_t t = (X[0] ^ X[i]) & P; X[0] ^= t; X[i] ^= t; }
        }
    }
    for (int i = 1; i < n; ++i) X[i] ^= X[i - 1];
    uint32_t t = 0;
    for (uint32_t Q = M; Q > 1; Q >>= 1)
        if (X[n - 1] & Q) t ^= Q - 1;
    uint32_t key = 0;
    for (int i = 0; i < n; ++i) {
        X[i] ^= t;
        for (int q = 0; q < bits; ++q) key |= ((X[i] >> q) & 1u) << (q * n + (n - 1 - i));
    }
    return key;
}

// Tile-moment expansions: order and record layout.  Per tile NREC = D + ncoef doubles - the centroid, then the coefficients in the order the
// Horner scheme of kde_moment_group_kernel reads them (D = 2: j = 8 ... 0; within j, i = 8 - j ... 0) - stored STRUCTURE-OF-ARRAYS: value k of
// tile t at mom[k * stride + t] (stride = tiles rounded up to 64), so that the 64 lanes of a wave, one tile each, load a value of 64
// consecutive tiles with one coalesced instruction.
#ifndef PBN_MOM_ORDER
#define PBN_MOM_ORDER 8
#endif
// log2((P + 1)!) of the remainder bound
#define PBN_MOM_LOG2_FACT (PBN_MOM_ORDER == 8 ? 18.469133f : PBN_MOM_ORDER == 6 ? 12.299208f : PBN_MOM_ORDER == 5 ? 9.491853f : PBN_MOM_ORDER == 4 ? 6.906891f : PBN_MOM_ORDER == 10 ? 25.250475f : -1.f)
constexpr int pbn_mom_coefs(int d) { return d == 1 ? PBN_MOM_ORDER + 1 : (PBN_MOM_ORDER + 1) * (PBN_MOM_ORDER + 2) / 2; }
constexpr int pbn_mom_rec(int d) { return d + pbn_mom_coefs(d); }   // 10 | 47

struct FinishArgs {
    const double* part;
    int nsplit;
    int64_t nqtiles;
    int64_t nq;
    double lognorm;
    double lognorm_marg;
    double* logl;        // nullable
    double* block_sums;  // nullable
    double* block_sums_marg;  // nullable, COND only: block_sums then holds the joint logl, this one the marginal logl
    const int32_t* scatter;   // nullable: logl[scatter[q]] instead of logl[q] (queries evaluated in Morton order)
};

// query groups (of 16 rows) per wave: 4, except the fp64 CKDE sweep (two accumulator + exp sets per group)
// which keeps 2 to hold 3 waves/SIMD
#ifndef PBN_QG_F64
#define PBN_QG_F64 4
#endif
template <bool F64, bool COND>
struct SweepQG {
    static constexpr int value = (F64 && COND) ? 2 : (F64 ? PBN_QG_F64 : 4);
};
// Pruned fp64 sweeps: 2 groups per wave since round 5 (4 before).  With per-group visit masks a wave visits the UNION of its groups' tiles and pays the
// mask test for every group; two groups halve that union's excess and the register need (fewer spills at four waves per SIMD), at twice the fragment
// loads per pair - cv64 2.32 -> 2.05 s, C3's first iteration 7.76 -> 7.22 s, 24 iterations 26.1 -> 24.8 s, handles d = 2 / 3 4.5 / 5.4 -> 4.2 / 4.9 ms
// (one group per wave: cv64 2.27 s, C3 6.91 / 25.3 s; profiles/r5/waves_probe.txt).  The moment pass follows (same grid mapping).
#ifndef PBN_QG_PRUNE
#define PBN_QG_PRUNE 2
#endif
#ifndef PBN_F16_QG_PRUNE
#define PBN_F16_QG_PRUNE 4   // query groups (tiles of 16 queries) per wave of the pruned fp32 sweeps
#endif
#ifndef PBN_QG_PRUNE_COND
#define PBN_QG_PRUNE_COND 2
#endif
int sweep_qg(int dtype, bool cond, int KS, bool prune = false);
bool sweep_folds_norm(int dtype, bool cond, int KS, int dm);   // see PackArgs::fold_norm
bool sweep_weights_norm(int dtype, bool cond, int KS, int dm); // see SweepArgs::wmul
// spatial sort + bounding boxes + exponent bounds of the pruned sweeps (SweepArgs::prune)
void launch_prune_keys(const PackArgs& a, int dtype, int zd, int kd, double* zrow, uint32_t* keys, int32_t* iota, hipStream_t st);
void launch_tile_boxes(const double* zrow, const int32_t* perm, int64_t n, int zd, int pd, double* box, double* zsorted, hipStream_t st);
// boxes of the 64-tile batches of every split of a pruned sweep (SweepArgs::batch_box): out[(split * batches_per_split + k)][2 * pd]
void launch_batch_boxes(const double* tile_box, int pd, int64_t ntiles, int64_t tiles_per_split, int nsplit, double* out, hipStream_t st);
// subpart (nullable): per query (in sorted order) the (offset, sum) partials of a sweep over a subsample of nsub training rows,
// P doubles per query, the pair to use at [which]: offset + log2(sum) - log2(nsub) is a second lower bound of the query's
// largest exponent
void launch_query_prepass(const double* zq_row, const int32_t* qperm, int64_t nq, const uint32_t* qkeys_sorted, const double* ztrain_sorted,
                          const uint32_t* tkeys_sorted, int64_t n, int zd, int pd, double* qbox, double* qthr, double* qlb, hipStream_t st,
                          const double* subpart = nullptr, int P = 2, int which = 0, double log2_nsub = 0.0, const double* tile_box = nullptr);
void sort_keys(pbn::dev_buf<char>& tmp, const uint32_t* keys_in, uint32_t* keys_out, const int32_t* vals_in, int32_t* vals_out, int64_t n,
               int bits, hipStream_t st);
// 52 (fp64) / 40 (fp32 on the f16 cores) at 10^6 training rows, + log2(n_train / 10^6): a constant bound (2.2e-10 / 9.1e-7 of a sum) on what
// pruning drops; PBN_PRUNE_MARGIN / PBN_PRUNE_MARGIN_F32 override the base, PBN_PRUNE_MARGIN_ADAPT=0 the scaling
double prune_margin(int dtype, int64_t n_train, bool sum_only = false);
bool use_f16x2(int dtype);   // fp32 tables: f16x2 split on the 16-bit matrix cores (default on)
int f16x2_mfmas(int dm);   // number of v_mfma_f32_16x16x32_f16 per (tile, group) for dm whitened dimensions
bool f16x2_w32p(int dm, int NB);  // the same for the pruned plain fp32 sweeps (kde_sweep_f16_w32p_body)
bool f16x2_w32(int dm, int NB);   // the training fragments carry the constants in the last three slots and kde_sweep_f16_w32_kernel applies

// ---- more than 32 whitened dimensions (the reference's kernels loop over any d: kde/KDE.hpp:592-640): a generic, runtime-sized form of
// the pack and of the sweep.  fp64 fragments in the classic order (fp32 tables are packed into doubles), norms added per value, the
// columns / centring offsets / whitening matrix in device memory; no folding, no weights, no pruning: correctness path, any d.
struct WidePackArgs {
    const void* base;      // device column-major table
    int64_t ld;
    const int* cols;       // device: d selected columns, whitening order
    const double* mu;      // device: d centring offsets
    const double* W;       // device: d x d row-major lower whitening matrix
    int d, KS, is_query, src_f32;
    int dm;                // whitened coordinates written to `pack` (<= d; the CKDE::cdf fragments contract over the evidence only: d - 1)
    int ldw;               // row stride of W (>= d: CKDE::sample packs the evidence with the leading block of the joint matrix)
    int64_t row0, n0, row1;
    const int32_t* rows;   // device gather list (nullable)
    int64_t n, ntiles;
    double* pack;          // [ntiles][KS][64]
    double* npack;         // [ntiles][16]
    const double* wu;      // device, nullable: CKDE::cdf - u = sum_j wu[j] (x_j - mu_j) over all d columns -> upack (npack layout)
    double* upack;
};
void launch_pack_wide(const WidePackArgs& a, hipStream_t st);
// plain fp64 sweep with a runtime number of K steps: Apack / nxpack / Bpack / nypack / ntiles / nqtiles / tiles_per_split / part of `a`
void launch_sweep_wide(const SweepArgs& a, int KS, int nsplit, hipStream_t st);

void launch_pack(const PackArgs& a, int dtype, hipStream_t st);
// dev_out[0] = max over the rows described by `a` of |z|^2 (all a.d whitened coordinates, base-2 units), as the bits of a non-negative
// double (atomic max: the caller zeroes it); src_dtype = element type of the table
void launch_max_norm2(const PackArgs& a, int src_dtype, double* dev_out, hipStream_t st);
// f16x2 fragments: the queries the pack flagged as beyond the f16 range (PackArgs::far_flag) are evaluated again in fp64 against the decoded
// training fragments and their partials are overwritten before the finish (exact to fp64; a few ms per 1 000 such queries at 1e6 training rows)
void launch_far_fix(const PackArgs& query_pack, const void* Apack, const void* Axpack, int NB, int64_t n_train, int64_t ntiles, double* part, int nsplit,
                    int64_t nqtiles, bool cond, hipStream_t st);
void launch_sweep(const SweepArgs& a, int dtype, int KS, bool cond, int nsplit, hipStream_t st);
void launch_finish(const FinishArgs& a, bool cond, double* dev_sum_out, hipStream_t st, double* dev_sum_marg_out = nullptr);
// out[i] = a[i] - b[i] (CKDE as joint - marginal when the two come from separate sweeps)
void launch_diff(double* out, const double* a, const double* b, int64_t n, hipStream_t st);

// CKDE::cdf (factors/continuous/CKDE.hpp:509-735): weights from the marginal sweep, normal cdf of the conditional mean.
struct CdfArgs {
    const void* Apack;   // classic fragments of the evidence dimensions
    const void* nxpack;
    const void* utrain;  // [ntiles][16] in C-row order: (x_t - b.e_t) / (sigma_c sqrt 2)
    const void* Bpack;
    const void* nypack;
    const void* uquery;  // [nqtiles][16]
    int64_t ntiles, nqtiles, tiles_per_split;
    double* part;        // [nsplit][nqtiles*16][4] : m, sum w, sum w*cdf, 0
    int KS;              // number of K steps (launch_cdf / launch_ucv fill it); beyond 4 the runtime-sized kernel reads it (fp64 fragments only)
};
void launch_pack_classic(const PackArgs& a, int dtype, hipStream_t st);
void launch_cdf(const CdfArgs& a, int dtype, int KS, int nsplit, hipStream_t st);
// UCV (kde/UCV.cpp:300-360): dev_out2[0] = sum over all ordered pairs (t, q < nq) of w, dev_out2[1] = of sqrt(w).
void launch_ucv(const CdfArgs& a, int dtype, int KS, int nsplit, int64_t nq, double* block_scratch, double* dev_out2, hipStream_t st);
void launch_cdf_finish(const double* part, int nsplit, int64_t nqtiles, int64_t nq, double* dev_out, hipStream_t st);

}  // namespace pbn
