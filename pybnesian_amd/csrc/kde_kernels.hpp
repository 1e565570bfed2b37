// Kernel argument blocks + launchers shared between kde_kernels.hip and the C-ABI glue.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#define PBN_MAX_D 17       // up to 16 whitened "main" dimensions (KS <= 4) + 1 CKDE extra coordinate
#define PBN_SWEEP_QG 4     // query groups (of 16 rows) per wave

namespace pbn {

struct PackArgs {
    const void* base;     // device column-major table
    int64_t ld;           // elements between columns
    int cols[PBN_MAX_D];  // selected columns, in whitening order
    int d;                // number of selected columns
    int dm;               // main (marginal) dimensions written to `pack`; d == dm or dm + 1
    int KS;               // ceil(dm / 4)
    int is_query;
    int64_t row0;         // contiguous source range ... or
    const int32_t* rows;  // ... device gather list (nullable)
    int64_t n;            // valid rows
    int64_t ntiles;       // ceil(n / 16)
    const double* W;      // device, d x d row-major lower-triangular whitening matrix
    const double* mu;     // device, d centring offsets
    void* pack;           // [ntiles][KS][64]
    void* npack;          // [ntiles][16]
    void* xpack;          // [ntiles][64] or null
};

struct SweepArgs {
    const void* Apack;
    const void* nxpack;
    const void* Axpack;
    const void* Bpack;
    const void* nypack;
    const void* Bxpack;
    int64_t ntiles;
    int64_t nqtiles;
    int64_t tiles_per_split;
    double* part;  // [nsplit][nqtiles*16][P]
};

struct FinishArgs {
    const double* part;
    int nsplit;
    int64_t nqtiles;
    int64_t nq;
    double lognorm;
    double lognorm_marg;
    double* logl;        // nullable
    double* block_sums;  // nullable
};

void launch_pack(const PackArgs& a, int dtype, hipStream_t st);
void launch_sweep(const SweepArgs& a, int dtype, int KS, bool cond, int nsplit, hipStream_t st);
void launch_finish(const FinishArgs& a, bool cond, double* dev_sum_out, hipStream_t st);

}  // namespace pbn
