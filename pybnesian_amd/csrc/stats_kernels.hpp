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
    const double* shift;  // device, pilot means indexed by TABLE column id
    double* partial;      // device, [nblocks][gram_ws(nct)]
};

int gram_ws(int nct);  // doubles per partial: nct(nct+1)/2 tiles of 256 + nct*16 column sums
void launch_pilot(const void* base, int64_t ld, const GramCols& gc, int n_cols, int64_t row0, const int32_t* rows,
                  int64_t n, int dtype, double* shift, hipStream_t st);
void launch_gram(const GramArgs& a, int dtype, int nblocks, double* out, hipStream_t st);
void launch_take(const void* src, int64_t ld_src, void* dst, int64_t ld_dst, const int32_t* rows, int64_t n,
                 int n_cols, int dtype, hipStream_t st);

}  // namespace pbn
