// Internal declarations shared by scoring.hip and hybrid.hip (score engine).
#pragma once
#include <cstdint>
#include <limits>
#include <map>
#include <set>
#include <memory>
#include <vector>

#include "common.hpp"

namespace pbn {
namespace score {

constexpr double MACHINE_TOL = 1.4901161193847656e-08;  // util/math_constants.hpp:30, sqrt(eps(double))
constexpr double LOG_2PI = 1.8378770664093454835606594728112;
constexpr double LOG_PI = 1.1447298858494001741434273513531;
const double INF = std::numeric_limits<double>::infinity();

struct Stats {  // pilot-shifted moments of a row set over all n columns (or a column subset, see users)
    int64_t N = 0;
    std::vector<double> S;  // n   : sum_r (x_rc - shift_c)
    std::vector<double> G;  // n*n : sum_r (x_ri - shift_i)(x_rj - shift_j), col-major symmetric
    void zero(int n) { N = 0; S.assign(n, 0.0); G.assign((size_t)n * n, 0.0); }
    void add(const Stats& o) {
        N += o.N;
        for (size_t i = 0; i < S.size(); ++i) S[i] += o.S[i];
        for (size_t i = 0; i < G.size(); ++i) G[i] += o.G[i];
    }
};

}  // namespace score
}  // namespace pbn

// Rows of the score regions grouped by (configuration of a set of discrete parents, region): see hybrid.hip
struct HybridGrouping {
    int nc = 1, nregions = 1;
    std::vector<int64_t> off;          // [nc * nregions + 1], cell = configuration * nregions + region
    pbn::dev_buf<int32_t> rows;        // device: permuted-table row ids, cell by cell
    pbn::dev_buf<int32_t> piece;       // device [npieces][3]: cell, first, last + 1 (pieces of <= 4096 rows)
    pbn::dev_buf<int32_t> piece_off;   // device [cells + 1]
    int npieces = 0;
    // moments of ALL continuous columns per cell (hybrid.hip group_moments): one gathered, segmented MFMA Gram per grouping - every
    // candidate over this grouping then takes its columns' entries on the host instead of a launch and a synchronisation of its own
    mutable std::vector<pbn::score::Stats> full;
    mutable int full_state = 0;        // 0 not tried, 1 ready, -1 not applicable (more than 64 continuous columns / too many cells)
};

struct pbn_scoredata {
    pbn::ctx_ptr ctx;
    int dtype = PBN_F64;
    int n = 0;  // continuous columns
    int split = PBN_SPLIT_NONE;
    int k = 0;
    int selector = PBN_SEL_NORMAL_REFERENCE;  // bandwidth selector of the CKDEs fitted while scoring
    // A(S, m) of the CKDE likelihood scores, keyed by [region, m, sorted columns...] (see pbn_score_batch)
    std::map<std::vector<int>, double> kde_cache;
    // totals of A(S, m) over the test regions of a score kind, keyed by [kind, m, sorted columns...]: installed by pbn_score_terms_put (values another
    // rank computed) and preferred over kde_cache, so that every rank of a job assembles a candidate from the very same doubles
    std::map<std::vector<int>, double> term_total;
    int64_t kde_sweeps = 0;
    int64_t precise_redos = 0;   // fp64 terms evaluated a second time at per-row accuracy (kde_sum_needs_precision)
    // pbn_scoredata_set_precise: CKDE terms of fp64 tables are evaluated at the accuracy of the per-row path (polynomial 2^f, per-row pruning
    // margin, no fp32 tail, no moment pass), whatever the caches hold - their results replace the cached ones.  What the search's near-tie
    // check asks for (hc.hip); never on by default.
    bool force_precise = false;
    // hybrid likelihood local scores by [kind, node type, variable, sorted parents...] (see pbn_score_batch)
    std::map<std::vector<int>, double> score_memo;
    int64_t memo_hits = 0;
    int64_t cache_resets = 0;   // times kde_cache / score_memo started over (PBN_SCORE_CACHE_ENTRIES)
    bool partial = false;  // moments hold only this rank's row share (pbn_scoredata_create_sharded)
    // one process per GPU (pbn_scoredata_set_comm): pbn_score_batch evaluates this rank's share and completes the batch through the
    // host's all-gather (shard.hip)
    bool has_comm = false;
    pbn_comm comm{};
    const pbn_table* src = nullptr;  // caller's table (borrowed)
    pbn_table* perm_table = nullptr; // owned permuted copy (null for PBN_SPLIT_NONE)
    const pbn_table* table() const { return perm_table ? perm_table : src; }
    std::vector<int32_t> perm;    // permuted row -> source row
    std::vector<int32_t> limits;  // k+1 fold limits inside the CV region
    int64_t n_cv = 0;             // rows of the CV / training region [0, n_cv)
    int64_t n_hold = 0;           // hold-out test rows [n_cv, n_cv + n_hold)
    std::vector<double> shift;    // n pilot shifts
    pbn::dev_buf<double> shift_dev;
    pbn::score::Stats all;               // CV / training region
    std::vector<pbn::score::Stats> fold; // k
    pbn::score::Stats hold;              // hold-out test region
    // World-size-invariant summation (pbn_scoredata_create_sharded): every region is cut into a FIXED number of super-blocks
    // (boundaries a function of the region's length only); a super-block's moments come from one segment of one segmented Gram
    // launch - a function of the segment's rows only - and a region's moments are the sum of its segments in segment order,
    // whichever rank computed which segment.
    std::vector<pbn::score::Stats> seg;  // per segment
    std::vector<int> seg_region;         // 0 .. k-1 folds (or 0 = the CV / training region when k == 0), then the hold-out region
    std::vector<int64_t> seg_r0, seg_r1; // row range of the segment
    // discrete (dictionary) columns, addressed as column ids n .. n + n_disc - 1, in permuted row order
    int n_disc = 0;
    std::vector<std::vector<int32_t>> codes;
    std::vector<int> card;
    pbn::dev_buf<int32_t> rows_dev;  // gather lists (valid rows of BIC / BGe candidates on tables with nulls)
    // by [kind, sorted discrete parents...]; shared: a HybridBatch keeps the groupings its enqueued work reads alive until it has flushed
    std::map<std::vector<int>, std::shared_ptr<HybridGrouping>> groupings;
    // validity of the continuous columns (BIC / BGe on tables with nulls): byte masks, empty = no nulls
    std::vector<std::vector<uint8_t>> valid;
    bool has_nulls = false;
};

namespace pbn {
namespace score {

// Shifted Gram of up to 64 columns over a contiguous row range or a device gather list -> raw S (d), G (d*d).
void gram_raw(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const int32_t* dev_rows,
              const double* shift_dev, double* S, double* G);
void subset_moments(const pbn_scoredata* sd, const Stats& st, const int* cols, int d, double* mu, double* sse);
void stats_minus(const Stats& a, const Stats& b, Stats& out);
// suspect (nullable, p >= 3 only): set when the normal equations cannot be trusted (see lg_fit) - refit with lg_fit_accurate
double lg_fit(int64_t N, int p, const double* mu, const double* S, double* beta, bool* suspect = nullptr);
// lg_accurate.hip: least-squares fit of cols[0] on cols[1..d-1] from double-double moments of the rows
// r < n0 ? row0 + r : row1 + (r - n0), r < n (through dev_rows when given).  d <= 16.
double lg_fit_accurate(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n0, int64_t row1, int64_t n,
                       const int32_t* dev_rows, double* beta);
bool lg_guard_on();   // PBN_LG_GUARD (default 1)
double bic_lg(int64_t N, int p, double variance);
double lg_slogl_from_rows(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const int32_t* dev_rows,
                          const double* beta, double variance);
double lg_slogl_from_moments(const pbn_scoredata* sd, const Stats& test, const int* cols, int p, const double* beta,
                             double variance);
// hybrid.hip: candidates with a discrete variable or discrete parents (synchronous)
// The CKDE slices (configuration c, region u) of a hybrid candidate fall into PBN_HYBRID_PARTS = 64 fixed PARTS, part = (c * regions + u) mod 64,
// and the candidate's score is the sum of its parts in part order - with one process as with many: a job with one process per GPU
// evaluates on rank r only the parts that rank owns and hands out the per-part sums (pbn_score_batch_parts).  Owners: the parts are dealt
// longest-processing-time first on their slices' training x test rows - every rank computes the same dealing from the same counts - so
// that the folds of a large configuration do not pile up on the ranks a fixed pattern would give them; which rank owns a part changes no
// bit of the result.
constexpr int PBN_HYBRID_PARTS = 64;
struct HybridParts {
    int rank, world;            // evaluate the parts dealt to `rank` of `world`
    double* out;                // [PBN_HYBRID_PARTS] per-part sums (0 for the parts not owned)
};
// CKDE candidates of one pbn_score_batch call evaluated together (hybrid.hip: HybridBatch): score_hybrid(..., hb, sink, &deferred) enqueues
// the candidate's sweeps and returns at once with *deferred = true; hybrid_batch_flush() runs the batch's one grouped chain, waits once and
// delivers every pending score through its sink (and its per-part sums through `parts->out`, which must stay valid until then).  Without a
// batch score_hybrid is synchronous.  hybrid_batch_end() frees the batch (waiting for whatever an exception left in flight).
struct HybridSink {
    double* out;                  // where the candidate's score goes
    std::vector<int> memo_key;    // non-empty: also remembered in pbn_scoredata::score_memo under this key
};
// pbn_score_batch without / with the job's communicator (scoring.hip / shard.hip)
int score_batch_local(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents,
                      const double* params, int n_params, double* out);
void score_batch_sharded(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents,
                         const double* params, int n_params, double* out);
struct HybridBatch;
HybridBatch* hybrid_batch_begin(pbn_scoredata* sd);
void hybrid_batch_flush(HybridBatch* hb);
void hybrid_batch_end(HybridBatch* hb) noexcept;
double score_hybrid(pbn_scoredata* sd, int kind, int var, int node_type, const int* parents, int p, const HybridParts* parts = nullptr,
                    HybridBatch* hb = nullptr, const HybridSink* sink = nullptr, bool* deferred = nullptr);

}  // namespace score
}  // namespace pbn
