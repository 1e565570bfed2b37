// Non-owning fitted-KDE description shared by the public KDE handles (capi.hip) and the score engine
// (scoring.hip): host-side whitening/normalisation data + pointers to packed training fragments.
#pragma once
#include <cmath>
#include <vector>

#include "common.hpp"
#include "kde_kernels.hpp"

namespace pbn {

struct KdeModel {
    int dtype = PBN_F64;   // element type of the TABLES (training and test)
    // fp32 table, fp64 fragments and sweep (kde_widen): the Gram-form distance z_t.z_q - |z_t|^2/2 - |z_q|^2/2 carries an absolute error
    // of ~2^-24 (|z_t|^2 + |z_q|^2) in fp32 - harmless while the whitened rows stay within tens of bandwidths of the centre, 1e-3...1e-2
    // on a logl when bandwidths are tiny against the spread of the data (diagonal bandwidths of nearly collinear columns, user-set
    // bandwidths), where the reference's fp32 differences (KDE.cl.src:173-226) hold 1e-5.  Such models pack double fragments from the
    // float columns and run the fp64 kernels; fdtype() is the type of the packs and sweeps, dtype stays the tables'.
    bool widen = false;
    int fdtype() const { return widen ? PBN_F64 : dtype; }
    int d = 0;         // number of variables
    int dm = 0;        // dimensions in the main MFMA contraction (d, or d-1 for CKDE)
    int KS = 0;        // ceil(dm / 4)
    bool cond = false;
    int64_t N = 0;
    int64_t ntiles = 0;
    double lognorm = 0.0, lognorm_marg = 0.0;
    std::vector<int> perm;     // whitening order -> position in the caller's column list
    // more than 32 main dimensions: the generic runtime-sized pack / sweep (kde_kernels.hpp "wide"), always in fp64 fragments,
    // plain sweeps only (a CKDE of that size is evaluated as joint - marginal, the reference's own formulation: CKDE.hpp:256-287)
    bool wide = false;
    std::vector<double> W, mu; // d x d row-major lower whitening matrix; d centring offsets (whitening order)
    void* Apack = nullptr;     // device [ntiles][KS][64]
    void* nxpack = nullptr;    // device [ntiles][16] norms, [ntiles][16] weights (classic packs), [ntiles] tile radii as doubles (fp64 fragments)
    bool tile_r = false;       // the tile radii were written (PackArgs::write_r)
    void* Axpack = nullptr;    // device [ntiles][64] (CKDE only)
    // tile pruning (set by kde_pack_train when asked for and the shape qualifies; see SweepArgs::prune): the packs above
    // hold the training rows in Morton order of their whitened coordinates
    bool prune = false;
    int pdims = 0, zdims = 0;
    const double* tile_box = nullptr;      // [ntiles][2 * pdims]
    const double* zsorted = nullptr;       // [N][zdims] whitened rows in packed order
    const uint32_t* keys_sorted = nullptr; // [N]
    // stratified subsample of the sorted training rows (every N / nsub-th), packed like the full set; a sweep over it gives
    // every query a lower bound of its largest exponent that holds in ALL dimensions (models with more than 3 of them)
    int64_t nsub = 0, ntiles_sub = 0;
    void* Asub = nullptr;
    void* nxsub = nullptr;
    void* Axsub = nullptr;
};

// Bytes needed for the three training-side fragment arrays.
struct KdePackBytes { size_t apack, nxpack, axpack; };
KdePackBytes kde_pack_bytes(int dtype, int dm, bool cond, int64_t n);

// Host math of KDE::_fit / ProductKDE::_fit / CKDE::_fit: permutation (evidence first for CKDE), Cholesky,
// whitening matrix, log-normalisation constants.  bw: H (d*d col-major, caller's order) or h (d).
// center: d offsets in the caller's order.  Throws singular_error when H is not PD.
void kde_prepare(KdeModel& m, int dtype, int d, int64_t n, const double* bw, int kind, bool cond, const double* center);

// fp32 tables: |z|^2 of the farthest whitened training row (one pass + one synchronisation), and the switch to fp64 fragments
// (before kde_pack_bytes / kde_pack_train).  kde_wants_widening: 2^-24 max|z|^2 > PBN_F32_WIDEN_AT (default 5e-4, the reference tests'
// own fp32 tolerance per logl; PBN_F32_WIDEN=0 switches the test off).
double kde_max_norm2(pbn_ctx* ctx, const KdeModel& m, const pbn_table* t, const int* cols, int64_t row0, int64_t n0, int64_t row1,
                     const int32_t* dev_rows = nullptr);
bool kde_wants_widening(double max_norm2, int dm);
void kde_widen(KdeModel& m);

// Wide packs (kde_kernels.hpp: WidePackArgs) for the consumers that keep their own fragments (CKDE::cdf / sample, UCV) when they hold more
// than 16 dimensions: uploads the columns (already in whitening order), centring offsets, whitening matrix (d rows of stride ldw) and
// - nullable - the cdf weights through the context's scratch, in stream order, and fills the pointers / sizes of `wa`.
void kde_wide_pack_args(pbn_ctx* ctx, WidePackArgs& wa, const pbn_table* t, const int* cols_whitened, int d, int dm, const double* W, int ldw,
                        const double* mu, const double* wu);

// Whiten + pack training rows (two contiguous ranges: [row0, row0+n0) ++ [row1, row1 + n - n0)).
// dev_rows (nullable): device gather list of m.N row ids, used instead of the ranges.
// prune: ask for the spatially sorted pack + tile boxes of the pruned sweep (fp64, <= 5 marginal dimensions, enough rows;
// otherwise ignored).  The training order inside the pack is then NOT the table order: only for consumers that need
// sums over the training rows (the score engine, logl / slogl of fitted handles); CKDE::sample / cdf keep their own
// table-ordered fragments.  Queries are evaluated in Morton order too and scattered back by the finish kernel.
// dev_max_norm2 (nullable; fp32 models on fp32 fragments only): receives |z|^2 of the farthest whitened training row as the bits of a
// non-negative double (atomic max into a slot the caller zeroed) - the score engine's check-after of kde_wants_widening().
void kde_pack_train(pbn_ctx* ctx, KdeModel& m, const pbn_table* t, const int* cols, int64_t row0, int64_t n0,
                    int64_t row1, const int32_t* dev_rows = nullptr, bool prune = false, double* dev_max_norm2 = nullptr);
// Whether kde_pack_train(prune = true) would build the Morton-ordered pack for a model of `dm` main dimensions and n rows.
bool kde_prune_applies(int dtype, int dm, int64_t n);
// Copies the pruning tables of a pruned pack out of the context arena into `store` (handles that outlive the call).
void kde_prune_persist(pbn_ctx* ctx, KdeModel& m, dev_buf<char>& store);

// pack(queries) -> sweep -> finish on the context stream; dev_logl / dev_sum nullable (device pointers).
// dev_sum_marg (CKDE only, nullable): dev_sum then receives the sum of the JOINT log-densities and dev_sum_marg the sum
// of the marginal ones, instead of their difference.
// precise: a sum is wanted (dev_logl == nullptr) but at the accuracy of the per-row path - the polynomial 2^f, the pruning margin of the
// per-row sweeps, no fp32 tail.  The callers' second evaluation of a sum that came out too close to zero for the sum-only path's
// absolute error budget (kde_sum_needs_precision).
void kde_eval_enqueue(pbn_ctx* ctx, const KdeModel& m, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                      double* dev_logl, double* dev_sum, const int32_t* dev_rows = nullptr, double* dev_sum_marg = nullptr, bool precise = false);
// The sum-only fp64 sweeps carry an ABSOLUTE error per log-density of at most 1.4e-7 (2^f on the fp32 unit) + 1.1e-7 (dropped mass) +
// 8e-8 (fp32 far tail) = 3.3e-7; measured with every rounding made one-sided (tests/test_error_budget_gpu.py): 3e-8.  Relative to a sum
// of n log-densities that is harmless unless the sum is a cancellation to ~0 (a table whose density happens to sit near 1 in its units):
// |sum| < 0.33 n is where the BOUND would reach the north star's bar, 1e-6 of the sum (the measured one-sided error reaches it at 0.03 n) - such
// a sum is evaluated once more at full precision.  (Rounds 4-5 drew the line at 0.66 n, 5e-7 of the sum: ordinary conditional log-likelihoods
// and one-variable terms with sigma ~ 0.13-0.47 sit at |mean logl| < 0.66 and paid two evaluations for a margin nobody asked for.)
inline bool kde_sum_needs_precision(double sum, int64_t n) { return n > 0 && std::fabs(sum) < PBN_TUNE_D(NEAR_ZERO_LOGL, 0.33) * (double)n; }

// Bandwidth selectors on a covariance (kde/NormalReferenceRule.hpp:72-134, kde/ScottsBandwidth.hpp:66-117).
void bandwidth_from_cov(int selector, int kind, const double* cov, int d, int64_t n, int dtype, double* out);
void bandwidth_full_block(int selector, const double* cov, int d, int rule_d, int64_t n, int dtype, double* out);

void check_cols(const pbn_table* t, const int* cols, int d, const char* who);
void check_range(const pbn_table* t, int64_t row0, int64_t n, const char* who);

}  // namespace pbn
