// Grouped KDE evaluation: MANY (training set, test set) units per launch chain.
//
// The score engine evaluates, for a variable set S, the k fold units "sum over the test rows of fold f of the log-density
// of the KDE fitted on the other folds" (learning/scores/cv_likelihood.cpp:18-22 x factors/continuous/CKDE.hpp:256-287 of
// the reference: one fit + one slogl per fold and candidate).  Round 2 ran one chain of ~24 launches per (set, fold): keys,
// a device sort of each side, boxes, two packs, prepass, sweep, finish.  Here the rows of a POOL (all rows the units of a
// set draw from) are ordered ONCE, in Morton order of pool-standardised coordinates; every unit's training / test order is
// a stable compaction of that order (prefix counts per region, no further sort), and every stage runs once for ALL units
// of ALL pools of a batch: ~15 launches per batch of up to 256 pools instead of 24 per unit.
//
// pool    = a row list (a contiguous range of the split-ordered table, or a block of a gather list) cut into R regions
//           (CV folds; hold-out train / test), + the variable set
// unit    = (pool, set of training regions, test region) + its own whitening / normalisation
#pragma once
#include <cstdint>
#include <vector>

#include "common.hpp"

#define PBN_GROUP_MAX_D 8     // variables per set on this path (pruned sweeps: KS <= 2)
#define PBN_GROUP_MAX_R 64    // regions per pool
#define PBN_GROUP_BLOCK 256   // sorted rows per block of the block-wise kernels

namespace pbn {

struct GPool {
    int64_t elem0;        // first element of this pool in the batch-wide element arrays
    int64_t row_base;     // pool position p -> table row  rows ? rows[row_base + p] : row_base + p
    const int32_t* rows;  // device gather list (nullable)
    int32_t n;            // rows in the pool
    int32_t nblk;         // ceil(n / PBN_GROUP_BLOCK) data blocks (+ 1 padding block in the flat block table)
    int32_t blk0;         // first flat block of this pool
    int32_t R;            // regions
    int32_t d;            // variables
    int32_t kd;           // dimensions of the Morton keys and boxes (<= PBN_PRUNE_PD)
    int32_t unit0, nunits;
    int32_t local;        // index of the pool inside the batch (high bits of its sort keys)
    int32_t pad_;
    int32_t rb[PBN_GROUP_MAX_R + 1];          // region r = pool positions [rb[r], rb[r + 1])
    int32_t test_unit[PBN_GROUP_MAX_R];       // unit (index inside the pool) whose test region is r, or -1
    int32_t cols[PBN_GROUP_MAX_D];
    double Wg[PBN_GROUP_MAX_D * PBN_GROUP_MAX_D];   // pool-level standardisation (row-major lower), rows < kd used
    double mug[PBN_GROUP_MAX_D];
};

struct GUnit {
    int32_t pool;
    int32_t test_region;
    uint64_t train_mask;   // bit r: region r trains this unit
    int32_t N, ntiles;     // training rows, 16-row tiles
    int32_t nq, nqtiles;   // test rows, 16-row tiles
    int32_t nsplit, tps;   // training splits of the sweep, tiles per split
    int32_t nwg, sum_slot; // sweep workgroups (rounded up to 64 in the flat table); slot of the unit's sum in the output
    int64_t wg0;           // first flat sweep workgroup
    // byte offsets into the batch arena
    int64_t apack, npack, zs, box, bpack, ny, zq, qpos, qbox, qthr, qlb, part, bsum;
    int64_t bbox;                    // boxes of the 64-tile batches of every split (sweeps: one uniform test per batch before the 64 tile tests)
    int64_t mom, rad2;               // tile-moment records / squared tile radii (0 = the chunk does not take the moment pass)
    int32_t nsplit_fin, pad2_;       // split partials per query the finish merges (2 x nsplit with a moment pass: its partials behind the sweep's)
    double lognorm;
    double W[PBN_GROUP_MAX_D * PBN_GROUP_MAX_D];    // whitening, row-major lower, base-2 units (kde_prepare)
    double mu[PBN_GROUP_MAX_D];
};

// Host description of a batch: filled by the caller (pools with rows / regions / columns / standardisation; units with pool,
// masks, N, nq, W, mu, lognorm, sum_slot), completed by kde_group_run (offsets, splits, tables).
struct GroupBatch {
    std::vector<GPool> pools;
    std::vector<GUnit> units;
};

// Runs the whole chain on ctx->stream for an f64 table and adds nothing to the host: out_sums[unit.sum_slot] receives the
// unit's sum of log-densities (device pointer).  The batch may hold any number of pools; it is cut into chunks that fit
// `arena_budget` bytes of the context's arena.
// dev_out_max (nullable, fp32 tables on f16x2 fragments): dev_out_max[unit.sum_slot] receives |z|^2 of the unit's farthest whitened
// training row, as the bits of a non-negative double (atomic max: the caller zeroes it) - the caller re-evaluates the units that
// kde_wants_widening() flags with force_f64 = true: fp64 fragments and fp64 sweeps on the float columns (KdeModel::widen).
void kde_group_run(pbn_ctx* ctx, const pbn_table* t, GroupBatch& b, double* dev_out_sums, double* dev_out_max = nullptr, bool force_f64 = false);

// Arena bytes one chain may take (PBN_GROUP_ARENA_MB) and the bytes a pool with its units needs: a caller that collects pools over many
// candidates hands them over about an arena-full at a time.
size_t kde_group_arena_budget();
size_t kde_group_pool_bytes(const GroupBatch& b, const GPool& P);

// Whether a set of d variables over training sets of at least n_min rows takes this path.
bool kde_group_applies(int dtype, int d, int64_t n_min, int R);

// per-unit record the grouped sweep kernels read (kde_kernels.hip)
struct GSweepUnit {
    const void* Apack;
    const void* nxpack;
    const void* Bpack;
    const void* nypack;
    const double* tile_box;
    const double* qtile_box;
    const double* qtile_thr;
    const double* qlb;
    double* part;
    const float* tile_rad2;    // moment pass (or null): squared tile radii, tile-moment records, the queries' whitened rows, the pass's partials
    const double* tile_mom;
    const double* zq;
    double* part_mom;
    int64_t wg0;
    int32_t ntiles, nqtiles, tps, nsplit, nwg, pdims;
    float margin;         // the unit's pruning margin (prune_margin(dtype, training rows))
    int32_t nbps;         // 64-tile batches per split (batch_box rows per split)
    const double* batch_box;
    int32_t mom_stride;   // doubles between consecutive values of the tile-moment records (tiles rounded up to 64)
    int32_t box_full;     // the boxes cover all of the unit's whitened dimensions (pool.kd == pool.d; SweepArgs::box_full)
};
struct GSweepArgs {
    const GSweepUnit* units;
    const int32_t* wg_unit;   // [total flat workgroups / 64] -> unit
    int64_t total_wg;
    int fold, wmul, count_redo, group_masks;
    int w32;               // fp32 units: paired kept tiles on 32x32x16 MFMAs (kde_sweep_f16_w32p_group_kernel; f16x2_w32p of the chunk's dimension)
    int moments;           // the units carry tile-moment records and the moment pass runs before the sweep (kde_sweep_group_kernel<..., MOM = true>)
    double prune_margin;   // > 0: one margin for every unit of the launch; 0: the units' own
    double far_span;       // SweepArgs::far_span of every unit of the launch
};
void launch_sweep_grouped(const GSweepArgs& g, int dtype, int KS, hipStream_t st);
// the moment pass of the same units (same flat workgroup table): d = 1 or 2, fp64 sum-only
void launch_moment_grouped(const GSweepArgs& g, int d, hipStream_t st);

}  // namespace pbn
