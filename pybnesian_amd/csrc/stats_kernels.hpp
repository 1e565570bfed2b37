// Argument blocks + launchers for the statistics kernels (stats_kernels.hip).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

namespace pbn {

struct GramCols {
    int cols[64];
};

struct GramArgs {
    const void* base;
    int64_t ld;
    GramCols gc;
    int n_cols;           // <= 64
    int64_t row0;         // contiguous range start (ignored when rows != null)
    const int32_t* rows;  // device gather list or null
    int64_t n;            // number of rows
    int64_t rows_per_block;
    const int32_t* blk;   // nullable device block table [nblocks][4] = (segment, first position, end position, partial slot): block b
                          // takes positions [blk[4b+1], blk[4b+2]) of the row list instead of its rows_per_block share and writes partial
                          // blk[4b+3] (launch_gram_segments); an entry with end <= first is a padding block: it writes nothing
    const void* rowmajor; // nullable, row lists only: a row-major mirror of the launch's columns (build_rowmajor_mirror) - the gathered
                          // rows are then read as whole contiguous rows instead of one element per column
    const double* shift;  // device, pilot means indexed by TABLE column id
    double* partial;      // device, [nblocks][gram_ws(nct)]
    int num_cus;          // of the device (0 = unknown): gram_glds_kernel runs one block per resident slot
    long long* stamps;    // measurement aid (PBN_GRAM_STAMPS): per-block start / end / hardware slot, or null
    int debug_skip;       // measurement aid (PBN_GRAM_DEBUG): 1 = no MFMAs (loads + LDS traffic only), 2 = no global loads
};

struct LgArgs {
    const void* base;
    int64_t ld;
    GramCols gc;      // cols[0] = variable, cols[1..p] = evidence
    int p;
    int64_t row0, n;
    const int32_t* rows;  // nullable device gather list: row r of the evaluation is rows[row0 + r] instead of row0 + r
    double beta[64];  // p+1 coefficients (intercept first)
    double inv_std;   // 1 / sqrt(variance)
    double cte;       // -0.5 log(variance) - 0.5 log(2 pi)
    int want_cdf;        // logl[] receives Phi((y - mean) / sigma) instead (LinearGaussianCPD::cdf)
    double* logl;        // device, nullable
    double* block_sums;  // device, nullable: ceil(n/256) partial sums
};
void launch_lg_logl(const LgArgs& a, int dtype, hipStream_t st);

int gram_ws(int nct);  // doubles per partial: nct(nct+1)/2 tiles of 256 + nct*16 column sums
void launch_pilot(const void* base, int64_t ld, const GramCols& gc, int n_cols, int64_t row0, const int32_t* rows,
                  int64_t n, int dtype, double* shift, hipStream_t st);
void launch_gram(const GramArgs& a, int dtype, int nblocks, double* out, hipStream_t st);
// Segmented form: the rows (a.rows: a gather list, or null = the table's own order) are cut into segments, the blocks of the
// table a.blk each cover a piece of one segment, and out receives one partial layout (gram_ws doubles) PER SEGMENT: segment g
// is the sum of partial SLOTS blk_off[g] .. blk_off[g + 1] - 1 (blk_off: device, [n_seg + 1]) in a fixed association (four runs in slot
// order, then the runs in order).  The launch order of the pieces (the order of a.blk's entries) is free - nblocks entries, every slot
// of a.partial ([slots][gram_ws]) written by exactly one of them.
void launch_gram_segments(const GramArgs& a, int dtype, int nblocks, const int32_t* blk_off, int n_seg, double* out, hipStream_t st);
// Row-major mirror of columns cols[0 .. n_cols) of a column-major table for the row-list Gram (GramArgs::rowmajor): `out` holds
// n rows of W = 16 ceil(n_cols / 16) elements, element (r, NCT (c & 15) + (c >> 4)) = column c of row r (NCT = W / 16: the NCT columns a
// Gram lane holds are adjacent), zeros past n_cols.  rowmajor_mirror_elems(n, n_cols) elements must be allocated.
size_t rowmajor_mirror_elems(int64_t n, int n_cols);
void build_rowmajor_mirror(const void* base, int64_t ld, const GramCols& gc, int n_cols, int64_t n, int dtype, void* out, hipStream_t st);
void launch_take(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* rows, int64_t n,
                 int n_cols, int dtype, hipStream_t st);

}  // namespace pbn
