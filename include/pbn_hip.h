/*
 * pbn_hip.h — C ABI of libpbn_hip.so: the MI355X (gfx950) implementation of PyBNesian's
 * KDE / CKDE log-likelihood and LinearGaussian / BIC / BGe / CV-likelihood scoring hot path.
 *
 * The reference (davenza/PyBNesian) has no C ABI: the hot path sits behind pybind11 classes and an
 * OpenCL singleton (pybnesian/opencl/opencl_config.hpp:123-224).  Every entry point below names the
 * reference routine it replaces (paths relative to /root/reference/pybnesian/).  Plain pointers and
 * sizes only; no torch / Arrow / Eigen types.  All matrices are column-major.  All functions return
 * a pbn_status; pbn_last_error() gives the message of the last failure on the calling thread.
 *
 * Memory: `const void* const* cols` are HOST pointers to contiguous column buffers (exactly what an
 * Arrow primitive array exposes); they are borrowed for the duration of the call.  Pointers named
 * `dev_*` are DEVICE pointers.  Handles own device memory and are freed explicitly.
 */
#ifndef PBN_HIP_H
#define PBN_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum {
    PBN_OK = 0,
    PBN_ERR_INVALID = 1,  /* std::invalid_argument  -> ValueError                                     */
    PBN_ERR_SINGULAR = 2, /* util::singular_covariance_data -> SingularCovarianceData (ValueError)    */
    PBN_ERR_DEVICE = 3    /* OpenCL enqueue failure (opencl_config.hpp:19-27) -> RuntimeError         */
} pbn_status;

typedef enum { PBN_F64 = 0, PBN_F32 = 1 } pbn_dtype; /* arrow::DoubleType / arrow::FloatType */

typedef enum {
    PBN_BW_FULL = 0, /* KDE: d x d bandwidth matrix H (kde/KDE.hpp:451-478)        */
    PBN_BW_DIAG = 1  /* ProductKDE: d variances h (kde/ProductKDE.hpp:153-190)     */
} pbn_bw_kind;

typedef enum {
    PBN_SEL_NORMAL_REFERENCE = 0, /* kde/NormalReferenceRule.hpp:72-134 */
    PBN_SEL_SCOTT = 1             /* kde/ScottsBandwidth.hpp:66-117     */
} pbn_selector;

typedef struct pbn_ctx pbn_ctx;     /* one device + one stream; replaces OpenCLConfig::get() singleton */
typedef struct pbn_table pbn_table; /* device-resident column-major table (DataFrame columns in HBM)  */
typedef struct pbn_kde pbn_kde;     /* fitted KDE / ProductKDE / CKDE device state                     */

const char* pbn_last_error(void);
const char* pbn_version(void);

/* ---- context: replaces opencl/opencl_config.cpp:149-220 (platform/device/queue singleton) ------ */
int pbn_ctx_create(int device, pbn_ctx** out);
/* Drops the creator's reference.  Every handle created on the context (tables, KDE models, score data, MI / kMI engines) holds one
 * of its own: the stream, the arenas and the lock go with the LAST reference, so destroy calls may arrive in any order (a garbage
 * collector finalises the members of a reference cycle in no particular order).  The pointer must not be passed to another
 * pbn_ctx_* function afterwards. */
void pbn_ctx_destroy(pbn_ctx* ctx);
int pbn_ctx_sync(pbn_ctx* ctx);
void* pbn_ctx_stream(pbn_ctx* ctx); /* hipStream_t the kernels are launched on (for event timing)    */
/* Per-kernel-class device timing with HIP events on the context stream (the reference has no profiling
 * queue, opencl_config.cpp:175).  Classes: 0 pack, 1 KDE sweep, 2 finish/reduce, 3 Gram/SSE.
 * pbn_ctx_kernel_time synchronises the stream and returns the accumulated ms and launch count.
 * on = 1: events, and the score engine issues everything on the context's one stream (attribution under a profiler);
 * on = 2: events around the sweep and Gram classes only, nothing else changes (launches on the engine's issue lanes are timed on their
 * lane - the classes' totals are device time and may overlap); 0: off.  Switching resets the totals. */
#define PBN_NUM_KERNEL_CLASSES 8
typedef enum { PBN_K_PACK = 0, PBN_K_SWEEP = 1, PBN_K_FINISH = 2, PBN_K_GRAM = 3, PBN_K_MOMENT = 4 } pbn_kernel_class;   /* MOMENT: the tile-moment pass beside the grouped fp64 sweeps */
int pbn_ctx_set_profiling(pbn_ctx* ctx, int on);
int pbn_ctx_kernel_time(pbn_ctx* ctx, int kernel_class, double* total_ms, int64_t* launches);

/* ---- tables: replaces DataFrame::to_eigen + OpenCLConfig::copy_to_buffer ------------------------
 * (dataset/dataset.hpp:236-338, opencl/opencl_config.hpp:226-239).  `valid` is an optional Arrow
 * validity bitmap already AND-combined over the columns (dataset.cpp:208-235); null rows are
 * compacted away on upload exactly as the reference does (dataset.hpp:92-106). */
int pbn_table_create(pbn_ctx* ctx, const void* const* cols, int n_cols, int64_t n_rows, int dtype,
                     const uint8_t* valid, int64_t valid_offset, pbn_table** out);
/* Borrow an existing device allocation (column c at dev_base + c*ld elements). */
int pbn_table_from_device(pbn_ctx* ctx, void* dev_base, int64_t ld, int n_cols, int64_t n_rows, int dtype,
                          pbn_table** out);
void pbn_table_destroy(pbn_table* t);
int64_t pbn_table_rows(const pbn_table* t);
int pbn_table_cols(const pbn_table* t);
/* Row gather on device (arrow::compute::Take, dataset.hpp:2072-2075): out = t[rows[0..n), :]. */
int pbn_table_take(const pbn_table* t, const int32_t* rows, int64_t n, pbn_table** out);
/* Copy columns back to host (read_from_buffer, opencl_config.hpp:241-250); out is n_rows*n_sel. */
int pbn_table_read(const pbn_table* t, const int* cols, int n_sel, void* out);

/* ---- column statistics: replaces DataFrame::means / cov / sse (dataset.hpp:208-234,340-512) -----
 * One pass, pilot-shifted centred Gram on f64 MFMA.  rows==NULL means the contiguous range
 * [row0, row0+n).  means: d doubles; sse: d*d doubles (sum of centred cross products). */
int pbn_table_sse(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* means,
                  double* sse);

/* ---- bandwidth selectors (host math on a d x d covariance; kde/NormalReferenceRule.hpp:72-134,
 * kde/ScottsBandwidth.hpp:66-117, util/basic_eigen_ops.hpp:136-148).  cov is d*d col-major, out is
 * d*d (PBN_BW_FULL) or d (PBN_BW_DIAG).  PBN_ERR_SINGULAR when n<=d or cov is not PD. */
int pbn_bandwidth(int selector, int kind, const double* cov, int d, int64_t n, int dtype, double* out);

/* ---- KDE / ProductKDE: replaces KDE::_fit / ProductKDE::_fit (kde/KDE.hpp:451-478,
 * kde/ProductKDE.hpp:153-190).  `bandwidth` is H (d*d col-major) or h (d variances), as chosen by
 * `kind`; the training rows are whitened, centred and packed into MFMA fragment order on device.
 * `center` (d doubles, nullable) is any offset near the column means (the distances do not depend on
 * it; it only keeps the whitened coordinates small); NULL lets the library take pilot means.
 * Dimensions: any number of variables, as in the reference (KDE.hpp:592-640 loops over d).  Up to 32 (+ the conditional one of
 * pbn_ckde_fit) run the templated MFMA shapes; beyond, a generic runtime-sized pack + sweep in fp64 fragments (a CKDE of that size as
 * joint - marginal, CKDE.hpp:256-287); pbn_ckde_cdf / pbn_ckde_sample / pbn_ucv_* likewise beyond 16.
 * fp32 tables: every whitened coordinate is cut into two f16 pieces and contracted on the matrix cores with f32 accumulation (as accurate as
 * the f32 Gram form: DESIGN.md 4) unless the whitened training rows reach so far from the centre that the Gram-form distances would lose
 * them (2^-24 max|z|^2 > 5e-4; 2^-22 max|z|^2 > 1e-3 for the dimensions whose fragments drop the product of the low pieces: 8, 9, 16-20) -
 * then fp64 fragments are packed from the float columns.  A query more than 65504 whitened units from the centre is evaluated in fp64. */
int pbn_kde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                const double* bandwidth, int kind, const double* center, pbn_kde** out);
/* CKDE: replaces CKDE::_fit (factors/continuous/CKDE.hpp:182-200).  cols[0] is the variable,
 * cols[1..d) the evidence; H is the joint bandwidth in that order (d*d col-major).  d==1 degrades
 * to a plain KDE exactly as CKDE.hpp:232-241 does. */
int pbn_ckde_fit(pbn_ctx* ctx, const pbn_table* train, const int* cols, int d, int64_t row0, int64_t n,
                 const double* H, const double* center, pbn_kde** out);
void pbn_kde_destroy(pbn_kde* k);
int64_t pbn_kde_num_instances(const pbn_kde* k);
double pbn_kde_lognorm(const pbn_kde* k, int which); /* 0 joint / plain, 1 marginal (CKDE only)      */

/* UCV bandwidth selection (kde/UCV.{hpp,cpp}; sum_ucv_* / ucv_diag kernels, kde/opencl_kernels/KDE.cl.src:470-574).
 * pbn_ucv_score: N * UCV(bandwidth) = UCVScorer::score_unconstrained / score_diagonal (UCV.cpp:226-360); the two
 * pair sums come from one self-sweep of the training rows.  pbn_ucv_bandwidth: UCV::bandwidth / diag_bandwidth
 * (:452-526) - a Nelder-Mead search (the reference uses NLopt's LN_NELDERMEAD, ftol_rel = xtol_rel = 1e-4) from
 * `start`, the normal reference bandwidth; kind = PBN_BW_FULL (d*d col-major) or PBN_BW_DIAG (d variances). */
int pbn_ucv_score(pbn_ctx* ctx, const pbn_table* table, const int* cols, int d, int64_t row0, int64_t n,
                  const double* bandwidth, int kind, double* out);
int pbn_ucv_bandwidth(pbn_ctx* ctx, const pbn_table* table, const int* cols, int d, int64_t row0, int64_t n, int kind,
                      const double* start, double* out, int64_t* n_evals);

/* logl: replaces KDE::_logl / ProductKDE::_logl / CKDE::_logl (KDE.hpp:513-547,
 * ProductKDE.hpp:192-238, CKDE.hpp:202-254).  Writes n doubles to HOST `out` (dev_out: DEVICE). */
int pbn_kde_logl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out);
int pbn_kde_logl_dev(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                     double* dev_out);
/* cdf: replaces CKDE::_cdf / _cdf_univariate / _cdf_multivariate (factors/continuous/CKDE.hpp:509-735 and the
 * conditional_means / normal_cdf kernels, kde/opencl_kernels/KDE.cl.src:376-468): P(X <= x | evidence) of each
 * test row, n doubles to HOST `out`.  Only for handles made by pbn_ckde_fit; one fused sweep instead of the
 * reference's W / mu / cdf / product N x m matrices. */
int pbn_ckde_cdf(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out);
/* sample: replaces CKDE::_sample / _sample_multivariate / _sample_indices_from_weights (CKDE.hpp:289-508) and the
 * normalize_accum_sum_mat_cols / find_random_indices kernels (KDE.cl.src:351-374).  Random numbers come from
 * std::mt19937{seed} through libstdc++'s distributions in the reference's call order; the instance selection (weights
 * of the evidence KDE, prefix sums, bracket search) runs on the device without the N x n matrix.  evidence: n rows of
 * the evidence columns (ev_cols in the factor's evidence order), NULL for a factor without evidence.  stream_n >= n:
 * how many samples the reference call would have been asked for (DiscreteAdaptator hands the whole batch size to every
 * slice factor, DiscreteAdaptator.hpp:442) - 0 means n.  out: n values of the factor's dtype on the HOST.  The
 * training table given to pbn_ckde_fit must still be alive. */
int pbn_ckde_sample(pbn_kde* k, int64_t n, int64_t stream_n, const pbn_table* evidence, const int* ev_cols,
                    uint32_t seed, void* out);
/* LinearGaussianCPD::sample (factors/continuous/LinearGaussianCPD.cpp:317-380), host only: n normal draws around
 * beta[0] plus beta[j+1] * evidence[j][i]; evidence[j] = n HOST values of ev_dtype. */
int pbn_lg_sample(int64_t n, const double* beta, int p, double variance, uint32_t seed, const void* const* evidence,
                  int ev_dtype, double* out);
/* DiscreteFactor::sample_indices (factors/discrete/DiscreteFactor.hpp:144-205), host only: logprob = CPT with the
 * variable fastest (card values per parent configuration, n_entries in total); parent_offset[i] = first entry of row
 * i's configuration (NULL without evidence); out: n category indices. */
int pbn_discrete_sample(int64_t n, const double* logprob, int card, int64_t n_entries, const int32_t* parent_offset,
                        uint32_t seed, int32_t* out);
/* slogl: replaces KDE::_slogl / ProductKDE::_slogl / CKDE::_slogl (KDE.hpp:549-562,
 * ProductKDE.hpp:295-308, CKDE.hpp:256-287): sum of logl over the rows, one scalar read-back. */
int pbn_kde_slogl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out);
/* Enqueue-only form for benchmarking / batching: result lands in DEVICE memory, no sync. */
int pbn_kde_slogl_async(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                        double* dev_out);

/* ---- score engine ---------------------------------------------------------------------------------
 * pbn_scoredata replaces the DataFrame held by a Score plus its CrossValidation / HoldOut adaptors
 * (learning/scores/cv_likelihood.hpp, holdout_likelihood.hpp, validated_likelihood.hpp:14-22;
 * dataset/crossvalidation_adaptator.hpp:15-58, dataset/holdout_adaptator.hpp:17-61): the table (all
 * columns continuous, no nulls) is permuted on device into split order with libstdc++
 * std::shuffle(std::mt19937{seed}) - the reference's exact fold membership - and per-fold moments are
 * cached.  `table` must outlive the handle. */
typedef struct pbn_scoredata pbn_scoredata;
typedef enum {
    PBN_SPLIT_NONE = 0,     /* BIC, BGe                                   */
    PBN_SPLIT_CV = 1,       /* CVLikelihood(df, k, seed)                  */
    PBN_SPLIT_HOLDOUT = 2,  /* HoldoutLikelihood(df, test_ratio, seed)    */
    PBN_SPLIT_VALIDATED = 3 /* ValidatedLikelihood(df, test_ratio, k, seed) */
} pbn_split_kind;
typedef enum {
    PBN_SCORE_BIC = 0,    /* learning/scores/bic.cpp:12-27                                      */
    PBN_SCORE_BGE = 1,    /* learning/scores/bge.hpp:154-234; params = iss_mu, iss_w, total_nodes[, nu x n] */
    PBN_SCORE_CVLIK = 2,  /* learning/scores/cv_likelihood.cpp:11-25                            */
    PBN_SCORE_HOLDOUT = 3 /* learning/scores/holdout_likelihood.cpp:14-23 (ValidatedScore::vlocal_score) */
} pbn_score_kind;
typedef enum { PBN_NODE_LG = 0, PBN_NODE_CKDE = 1, PBN_NODE_DISCRETE = 2 } pbn_node_type; /* LinearGaussianCPDType / CKDEType / DiscreteFactorType */

int pbn_scoredata_create(pbn_ctx* ctx, const pbn_table* table, int split, int k, uint32_t seed, double test_ratio,
                         pbn_scoredata** out);
/* Row-sharded construction for one-process-per-GPU jobs (BGe / BIC / LG-CV row): every rank holds
 * the same table.  Each region (fold, hold-out part) is cut into a fixed number of super-blocks (PBN_MOMENT_SUPERBLOCKS,
 * default 16; boundaries depend on the region's length only); rank r takes the Gram moments of segments
 * [S r / world, S (r + 1) / world) and leaves the others zero; the host adds the ranks' pbn_scoredata_moments buffers
 * (exact: a segment is non-zero on one rank only) and installs them with set != 0 on every rank, which rebuilds the
 * regions' totals in segment order - bit-identical for every world size, world = 1 included.  Until then
 * pbn_score_batch refuses the handle.  The reference has no counterpart (single process, CPU Eigen). */
int pbn_scoredata_create_sharded(pbn_ctx* ctx, const pbn_table* table, int split, int k, uint32_t seed,
                                 double test_ratio, int rank, int world, pbn_scoredata** out);
/* buf: per segment (the super-blocks of the k folds or of the single CV/training region, then of the hold-out part)
 * S[n] then G[n*n]; *len = doubles. */
int pbn_scoredata_moments(pbn_scoredata* sd, double* buf, int64_t* len, int set);
/* Bandwidth selector (PBN_SEL_*) of the CKDEs fitted while scoring: the reference passes it through the scores'
 * construction_args to CKDEType::new_factor (learning/scores/cv_likelihood.hpp:19-27). Default normal reference. */
int pbn_scoredata_set_selector(pbn_scoredata* sd, int selector);
/* CKDE likelihood scores are assembled from A(S, m) = sum over a test region of log KDE(S) under the m-dimensional
 * bandwidth rule; the engine remembers every A it has swept.  entries / sweeps so far. */
int pbn_scoredata_cache_stats(const pbn_scoredata* sd, int64_t* entries, int64_t* sweeps);
void pbn_scoredata_destroy(pbn_scoredata* sd);
/* Dictionary-encoded columns (arrow::DictionaryArray indices, factors/discrete/discrete_indices.cpp): n_disc int32
 * arrays in SOURCE row order + cardinalities.  They get column ids n_cols .. n_cols+n_disc-1 in pbn_score_batch;
 * a continuous column with discrete parents is scored as CLinearGaussianCPD / HCKDE (DiscreteAdaptator.hpp:201-348),
 * a discrete column (node type PBN_NODE_DISCRETE, discrete parents only) as DiscreteFactor. */
int pbn_scoredata_set_discrete(pbn_scoredata* sd, int n_disc, const int32_t* const* codes, const int* cardinality);
/* BIC / BGe on tables with nulls: masks[c] = byte array (1 = valid) of continuous column c in source row order, or
 * NULL when the column has no nulls (valid_rows / combined_bitmap semantics of bic.cpp:12-27, bge.hpp:184-234). */
int pbn_scoredata_set_validity(pbn_scoredata* sd, const uint8_t* const* masks);
/* perm: n_rows ints (source row of every permuted row); limits: k+1 fold limits; all nullable. */
/* The split layout alone, host only (dataset::CrossValidation / HoldOut as stand-alone objects,
 * dataset/crossvalidation_adaptator.hpp:15-58, holdout_adaptator.hpp:17-61): perm[n_rows] = source row of every
 * split-ordered row, limits[k+1] fold limits (nullable), sizes of the CV/training and hold-out regions. */
int pbn_split_layout(int64_t n_rows, int split, int k, uint32_t seed, double test_ratio, int32_t* perm, int32_t* limits,
                     int64_t* n_cv, int64_t* n_hold);
int pbn_scoredata_layout(const pbn_scoredata* sd, int32_t* perm, int32_t* limits, int64_t* n_cv, int64_t* n_hold);
/* MLE<LinearGaussianCPD>::estimate (learning/parameters/mle_LinearGaussianCPD.hpp:195-221) from the cached
 * moments of the training region: beta has p+1 entries (intercept first). */
int pbn_lg_fit(const pbn_scoredata* sd, int var, const int* parents, int p, double* beta, double* variance);
/* The same fit on any row range of a table (one Gram pass): cols[0] is the variable, cols[1..d) the evidence. */
int pbn_lg_fit_table(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, double* beta, double* variance);
/* LinearGaussianCPD::logl / slogl (factors/continuous/LinearGaussianCPD.cpp:92-149,251-292): one streaming
 * pass; out_logl (n host doubles) and out_slogl are nullable. */
int pbn_lg_logl(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const double* beta, double variance,
                double* out_logl, double* out_slogl);
/* LinearGaussianCPD::cdf (factors/continuous/LinearGaussianCPD.cpp:171-249): Phi((y - beta.x) / sigma), n doubles to HOST. */
int pbn_lg_cdf(const pbn_table* t, const int* cols, int d, int64_t row0, int64_t n, const double* beta, double variance,
               double* out);
/* Score::local_score for a batch of candidates (replaces the serial double loop of
 * learning/operators/operators.cpp:100-132,296-347): candidate c scores column var[c] given
 * parents[par_off[c] .. par_off[c+1]) with node type node_type[c] (NULL = all LinearGaussian).
 * This is SURVEY.md 8b's `pbn_lg_score_batch` and `pbn_ckde_slogl_batch` in one call ("one call per batch"): LinearGaussian
 * candidates are host arithmetic on the handle's moments; the CKDE candidates of a likelihood kind are reduced to their unknown
 * (variable set, region) terms, and ALL folds of ALL sets of the batch are evaluated by one launch chain (one ordering of every
 * set's rows, one sweep launch - learning/scores/cv_likelihood.cpp:18-22 x factors/continuous/CKDE.hpp:256-287 fit and
 * evaluate per fold and candidate). */
int pbn_score_batch(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off,
                    const int* parents, const double* params, int n_params, double* out);
/* on != 0: the CKDE terms of fp64 tables are evaluated at the accuracy of the per-row path whatever the handle's caches hold, and replace
 * what they hold (what the search's near-tie check asks for: pbn_hc_config.near_tie_abs; the reference has one precision only,
 * kde/KDE.hpp:592-640).  Off by default. */
int pbn_scoredata_set_precise(pbn_scoredata* sd, int on);

/* The two halves of a CKDE likelihood score as functions of a variable SET: local(v | P) = A({v} u P, d) - A(P, d), d = |P| + 1, with
 * A(S, m) = sum over the split's test regions of sum_q log KDE_S(q) under the bandwidth rule for m dimensions on the region's training
 * rows (CKDE.hpp:186-199: the marginal of the reference is the KDE of the parents with H[1:, 1:]).  A({s, t}, 2) serves s -> t and t -> s,
 * A({s}, 2) every child of s - so a job with one process per GPU deals the TERMS of a delta-cache batch, not its candidates (SURVEY.md
 * section 8e; the reference has no counterpart).  Term i = columns vars[off[i] .. off[i+1]) (continuous, any order) with m[i] = their
 * number (a joint term) or one more (the marginal term of a candidate with these parents).
 *   pbn_score_terms          evaluates the terms (grouped engine, set-function cache) -> out[i] = total over the regions;
 *   pbn_score_terms_put      installs totals computed elsewhere: pbn_score_batch prefers them, and assembles every CKDE candidate as
 *                            (sum of joint terms in region order) - (sum of marginal terms in region order) with or without them, so
 *                            every rank - and the one-process run - produce the same doubles;
 *   pbn_score_terms_missing  missing[i] = 1 where no total is installed. */
int pbn_score_terms(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, double* out);
int pbn_score_terms_put(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, const double* values);
/* One REGION of a term: out[i] = the part of term i's total that region[i] contributes (a CV fold 0 .. k-1 of pbn_scoredata_create; 0 for the
 * hold-out score).  A term's total is its regions added in region order (cv_likelihood.cpp:18-22 adds the folds' slogl in fold order), so an
 * update batch with fewer unknown terms than ranks is dealt (term, fold) by (term, fold), the per-region values gathered, added in region order
 * by every rank and installed with pbn_score_terms_put: the same doubles as pbn_score_terms.  Totals installed earlier are not consulted. */
int pbn_score_term_regions(pbn_scoredata* sd, int kind, int n_items, const int* off, const int* vars, const int* m, const int* region, double* out);
int pbn_score_terms_missing(pbn_scoredata* sd, int kind, int n_terms, const int* off, const int* vars, const int* m, int* missing);
/* CKDE candidates with discrete parents (DiscreteAdaptator.hpp:201-348: one CKDE per configuration of the discrete parents): the
 * slices (configuration c, test region u) of a candidate fall into 64 fixed parts, part = (c * regions + u) mod 64, and
 * pbn_score_batch forms the score as the sum of the parts in part order.  pbn_score_batch_parts evaluates only the parts dealt
 * to rank `part` of `n_parts` (<= 64) - longest-processing-time first on the slices' training x test rows, the same dealing on
 * every rank - and returns the per-part sums, out[c * 64 + p]: added over the ranks (every part is non-zero on one rank only) and
 * then over p in order they give pbn_score_batch's value bit for bit.  The update
 * batches of a restricted hill-climb hold a handful of such candidates - fewer than ranks; their slices are what can be shared. */
int pbn_score_batch_parts(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off,
                          const int* parents, int part, int n_parts, double* out);

/* ---- one process per GPU: the delta-score cache sharded BEHIND the boundary (shards the serial double
 * loops of learning/operators/operators.cpp:100-132,296-347 and the fold loop of learning/scores/cv_likelihood.cpp:5-25; the
 * reference is one process and has no counterpart).  The host supplies ONE collective - an all-gather of doubles - as a function
 * pointer (RCCL `ncclAllGather` on a device staging buffer, MPI_Allgather, torch.distributed ...: INTEGRATION.md shows the RCCL
 * form); the library plans every batch (which rank evaluates which CKDE term, (term, fold) pair, hybrid slice part or whole
 * candidate - a pure function of the batch, identical on every rank), evaluates this rank's share, makes ONE all-gather per batch,
 * installs the gathered terms and assembles every candidate from the same doubles on every rank: the one-process score, bit for
 * bit, for every world size.
 *   all_gather(user, send, count, recv): every rank contributes `count` doubles; recv[r * count .. (r + 1) * count) = rank r's, in
 *   rank order on every rank.  HOST pointers.  Returns 0 on success.  Every rank of the job makes the same sequence of calls. */
typedef int (*pbn_allgather_fn)(void* user, const double* send, int64_t count, double* recv);
typedef struct {
    int rank, world;              /* 0 <= rank < world; world = 1 is allowed (the collective is still made)          */
    pbn_allgather_fn all_gather;
    void* user;
} pbn_comm;
/* Binds the job's communicator to a score handle (copied; NULL unbinds): from then on pbn_score_batch - and therefore every
 * pbn_hc_score_fn that calls it, i.e. OperatorSet::cache_scores / update_scores of a hill-climb - evaluates only this rank's share of
 * the batch's device work and completes it through the collective.  LinearGaussian / discrete candidates (O(p^3) host arithmetic on
 * replicated moments) are computed by every rank. */
int pbn_scoredata_set_comm(pbn_scoredata* sd, const pbn_comm* comm);
/* The exchange of pbn_scoredata_create_sharded in one call: all-gathers the ranks' segment moments, adds them in rank order and
 * installs the totals (see there: bit-identical for every world size). */
int pbn_scoredata_reduce_moments(pbn_scoredata* sd, const pbn_comm* comm);
/* KDE / ProductKDE / CKDE slogl with the TEST rows split over the ranks (the fitted model replicated): rank r evaluates
 * rows [row0 + n r / world, row0 + n (r + 1) / world), the partial sums are gathered and added in rank order. */
int pbn_kde_slogl_sharded(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, const pbn_comm* comm, double* out);
/* The planner and driver behind pbn_scoredata_set_comm, over an abstract engine (what a batch is made of, as function pointers with the
 * signatures of pbn_score_batch / pbn_score_terms* / pbn_score_batch_parts minus the handle): pbn_score_batch binds it to its own handle;
 * hosts with their own evaluator - and the CPU tests of this repository - bind theirs.  `shape` reports, for a score kind, the number of
 * regions a term adds up (CV folds; 1 for hold-out) and the training / test rows of one region: the plan prices a term by the rows a
 * sweep of its dimension meets (pruned sweeps: ~ N^(4/(d+4)) rows per query), not by a fixed table.  terms* / batch_parts may be NULL (then
 * CKDE candidates are dealt whole, by variable set).  shard_all != 0 deals every candidate, whatever its node type. */
typedef struct {
    void* user;
    int n_cont;                   /* column ids below it are continuous, the others dictionary columns */
    int (*shape)(void* user, int kind, int* regions, int64_t* train_rows, int64_t* test_rows);
    int (*batch)(void* user, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents, double* out);
    int (*terms_missing)(void* user, int kind, int n_terms, const int* off, const int* vars, const int* m, int* missing);
    int (*terms)(void* user, int kind, int n_terms, const int* off, const int* vars, const int* m, double* out);
    int (*term_regions)(void* user, int kind, int n_items, const int* off, const int* vars, const int* m, const int* region, double* out);
    int (*terms_put)(void* user, int kind, int n_terms, const int* off, const int* vars, const int* m, const double* values);
    int (*batch_parts)(void* user, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents,
                       int part, int n_parts, double* out);
    /* Optional (NULL: pbn_shard_term_cost on the engine's shape): the engine's own price of one region of a term of `dims` columns - only
     * ratios matter.  pbn_score_batch's engine prices what ITS kernels cost (profiles/r6/term_prices.txt: on fp64 tables whose folds take the
     * tile-moment pass a one-variable term is 2.4x cheaper than the list says).  Must be a pure function of replicated state. */
    double (*term_price)(void* user, int kind, int dims);
} pbn_shard_engine;
int pbn_shard_batch(const pbn_shard_engine* engine, const pbn_comm* comm, int kind, int n_cand, const int* var, const int* node_type,
                    const int* par_off, const int* parents, int shard_all, double* out);
/* The dealing rule itself (host only): items in order of decreasing cost - equal costs by decreasing tie[] (NULL: none), then by
 * index - each to the least loaded rank (lowest rank on ties); load[world] is read and updated when given (loads carried over from
 * earlier dealings), else starts at zero.  pbn_shard_term_cost: the price of sweeping one region of a term of `dims` columns. */
int pbn_shard_deal(int n_items, const double* cost, const uint32_t* tie, int world, double* load, int* owner);
double pbn_shard_term_cost(int dims, int64_t train_rows, int64_t test_rows);

/* ---- greedy hill-climbing (host logic; replaces learning/algorithms/hillclimbing.hpp:62-199 driving
 * learning/operators/operators.{hpp,cpp}).  Nodes are 0..n_nodes-1 in model.nodes() order.  Every step's
 * Score::local_score requests are handed to `score` as ONE batch (same layout as pbn_score_batch);
 * validated != 0 asks for ValidatedScore::vlocal_score.  The callback returns 0 on success. */
typedef enum { PBN_BN_GAUSSIAN = 0, PBN_BN_SEMIPARAMETRIC = 1, PBN_BN_KDE = 2, PBN_BN_CLG = 3 } pbn_bn_type;
typedef int (*pbn_hc_score_fn)(void* user, int validated, int n_cand, const int* var, const int* node_type,
                               const int* par_off, const int* parents, double* out);
typedef struct {
    int n_nodes;
    int bn_type;             /* pbn_bn_type                                                              */
    const int* node_types;   /* n_nodes pbn_node_type of the start model (NULL: LG, or CKDE for PBN_BN_KDE) */
    int n_arcs;
    const int* arcs;         /* start model arcs, (source, target) pairs                                 */
    int n_arc_blacklist;
    const int* arc_blacklist;
    int n_arc_whitelist;
    const int* arc_whitelist;
    int n_type_blacklist;
    const int* type_blacklist; /* (node, node_type) pairs                                                */
    int n_type_whitelist;
    const int* type_whitelist;
    int op_arcs;             /* ArcOperatorSet in the pool                                               */
    int op_node_type;        /* ChangeNodeTypeSet in the pool                                            */
    int arcs_first;          /* pool order: 1 = [arcs, node_type] (validate_options.cpp default)         */
    int max_indegree;        /* 0 = unlimited                                                            */
    int max_iters;
    double epsilon;
    int patience;
    int validated;           /* score is a ValidatedScore                                                */
    /* Callback::call(model, operator, score, iteration) (learning/algorithms/callbacks/callback.hpp; call sites
     * hillclimbing.hpp:127,180,195): iteration 0 with the start model, every iteration after the operator was applied,
     * and once more with the returned model.  op = {kind, source|node, target|type}, kind -1 = no operator.
     * arcs = 2*n_arcs node indices (source, target).  Non-zero return aborts the search.  Nullable. */
    int (*on_iter)(void* user, int iteration, const int* op, double delta, int n_arcs, const int* arcs,
                   const int* node_types);
    void* on_iter_user;
    /* Conditional networks (ConditionalBayesianNetwork; operators.cpp:134-256,365-437, operators.hpp:526-578): the
     * last n_interface ids n_nodes .. n_nodes+n_interface-1 are interface nodes - parents only, never scored, never
     * flipped.  node_types then holds n_nodes+n_interface entries, arc sources may be interface ids, the delta
     * matrix of pbn_hc_get is (n_nodes+n_interface) x n_nodes. */
    int n_interface;
    /* Near ties of likelihood scores (round 6; 0 = off, the reference's behaviour): when the best operator and the runner-up differ by less
     * than near_tie_abs and one of them rests on a CKDE local score, the local scores behind BOTH deltas are asked for again with bit 1 of
     * the callback's `validated` argument set (validated | 2 = "at full precision": polynomial 2^f, per-row pruning margin, no fp32 tail, no
     * moment pass - pbn_scoredata_set_precise) and the operator with the larger precise delta is applied.  The sum-only sweeps carry up to
     * 3.3e-7 per log-density with a mean bias of -1.3e-9: 4 x 3.3e-7 x test rows is the band in which that could decide. */
    double near_tie_abs;
} pbn_hc_config;
typedef struct {
    int iterations;
    int64_t cells_scored;      /* delta cells written by cache_scores + update_scores                    */
    int64_t local_score_evals; /* local_score / vlocal_score evaluations requested                       */
    int trace_capacity;        /* in: capacity of trace (ops) ; trace = 4 ints per applied operator:     */
    int* trace;                /*   kind (0 add, 1 remove, 2 flip, 3 change type), source|node, target|type, 0 */
    double* trace_delta;       /* delta of every applied operator (nullable)                             */
    int trace_len;
    int64_t near_tie_redos;    /* iterations whose two best operators were re-scored at full precision (near_tie_abs)  */
} pbn_hc_stats;
int pbn_hc_estimate(const pbn_hc_config* cfg, pbn_hc_score_fn score, void* user, int* out_arcs, int* out_n_arcs,
                    int* out_node_types, pbn_hc_stats* stats);

/* Stateful form for callers that drive their own loop: OperatorSet::{cache_scores, find_max, find_max_tabu,
 * update_scores} and LocalScoreCache (learning/operators/operators.hpp:295-355).  The handle owns a copy of the
 * model; pbn_hc_set_model re-synchronises it after the caller applied an operator.  find_max: op = {kind, source|node,
 * target|node_type}, kind -1 when no operator is available; tabu = 4 ints per forbidden operator (same encoding). */
typedef struct pbn_hc pbn_hc;
int pbn_hc_create(const pbn_hc_config* cfg, pbn_hc_score_fn score, void* user, pbn_hc** out);
void pbn_hc_destroy(pbn_hc* h);
int pbn_hc_set_model(pbn_hc* h, int n_arcs, const int* arcs, const int* node_types);
int pbn_hc_cache_scores(pbn_hc* h);
int pbn_hc_find_max(pbn_hc* h, int n_tabu, const int* tabu, int* op, double* delta);
int pbn_hc_update_scores(pbn_hc* h, int n, const int* nodes);
/* local: n cached local scores; delta_arcs: n*n col-major (row = source, column = target); delta_types: n. Nullable. */
int pbn_hc_get(pbn_hc* h, double* local, double* delta_arcs, double* delta_types);

/* ---- restriction phase of MMHC ---------------------------------------------------------
 * Conditional-independence test callback: p-value of v1 _||_ v2 | cond (indices into the test's variables);
 * IndependenceTest::pvalue (learning/independences/independence.hpp:17-40).  NaN aborts the search. */
typedef double (*pbn_ci_pvalue_fn)(void* user, int v1, int v2, int n_cond, const int* cond);
/* n_tests independent tests at once: cond_off has n_tests + 1 offsets into cond; NaN in out[i] aborts the search. */
typedef void (*pbn_ci_pvalue_batch_fn)(void* user, int n_tests, const int* v1, const int* v2, const int* cond_off,
                                       const int* cond, double* out);
/* LinearCorrelation (learning/independences/continuous/linearcorrelation.hpp:62-88, .cpp:9-100): covariance of all
 * columns of `table` taken once on the device (Gram kernel), partial correlations from the eigen-decomposition of the
 * (k+2)-variable block, two-sided Student-t p-value.  pbn_lincor_pvalue has the pbn_ci_pvalue_fn signature with
 * user = the handle. */
typedef struct pbn_lincor pbn_lincor;
int pbn_lincor_create(pbn_ctx* ctx, const pbn_table* table, pbn_lincor** out);
int pbn_lincor_from_cov(int n, int64_t rows, const double* cov /* n*n col-major */, pbn_lincor** out); /* host only */
void pbn_lincor_destroy(pbn_lincor* h);
int pbn_lincor_cov(const pbn_lincor* h, double* cov /* n*n */);
double pbn_lincor_pvalue(void* user, int v1, int v2, int n_cond, const int* cond);
/* Hybrid MutualInformation (learning/independences/hybrid/mutual_information.{hpp,cpp}): conditional-Gaussian MI of
 * mixed discrete / continuous variables and its chi-square p-value.  table = the continuous columns (device, may be
 * NULL), codes[j] = n_rows int32 category indices of discrete column j (HOST), cardinality[j] its categories.
 * Variable ids: continuous columns first (table order), then the discrete ones.  One device pass per test gathers the
 * per-configuration counts and moments; pbn_mi_pvalue has the pbn_ci_pvalue_fn signature (user = the handle, indices
 * mapped through pbn_mi_set_order when set). */
typedef struct pbn_mi pbn_mi;
int pbn_mi_create(pbn_ctx* ctx, const pbn_table* table, int64_t n_rows, int n_disc, const int32_t* const* codes,
                  const int* cardinality, int asymptotic_df, pbn_mi** out);
void pbn_mi_destroy(pbn_mi* h);
int pbn_mi_value(pbn_mi* h, int v1, int v2, int n_cond, const int* cond, double* mi, double* df);
double pbn_mi_pvalue(void* user, int v1, int v2, int n_cond, const int* cond);
void pbn_mi_pvalue_batch(void* user, int n_tests, const int* v1, const int* v2, const int* cond_off, const int* cond,
                         double* out); /* pbn_ci_pvalue_batch_fn: one launch for many tests */
int pbn_mi_set_order(pbn_mi* h, int n, const int* ids);
/* ChiSquare::pvalue (learning/independences/discrete/chi_square.cpp:8-139) over the discrete columns of the same
 * handle (pbn_ci_pvalue_fn signature). */
double pbn_chisq_pvalue(void* user, int v1, int v2, int n_cond, const int* cond);
/* Tables with nulls (hybrid/mutual_information.cpp:152-215, the contains_null overloads): a discrete null is code -1 in
 * pbn_mi_create; continuous nulls are NaN cells of the columns flagged here, with the pilot shift to use for them.  A
 * test then counts only the rows valid in all of its variables. */
int pbn_mi_set_continuous_nulls(pbn_mi* h, const unsigned char* flags, const double* shift);
/* LinearCorrelation::pvalue on a table with nulls (continuous/linearcorrelation.cpp:20-122, pvalue_impl): covariance of
 * [v1, v2, cond...] over the rows valid in all of them, then the partial-correlation t-test with valid_rows - 2 - n_cond
 * degrees of freedom.  pbn_ci_pvalue_fn signature, user = a pbn_mi handle over continuous columns. */
double pbn_mi_lincor_pvalue(void* user, int v1, int v2, int n_cond, const int* cond);
/* joint_counts (factors/discrete/discrete_indices.cpp:134-150) of n_vars categorical variables of the handle: out has
 * prod(cardinality) entries, index = sum code_i * stride_i with the first variable fastest.  Serves BDe (learning/scores/
 * bde.cpp:5-50). */
int pbn_mi_counts(pbn_mi* h, int n_vars, const int* vars, double* out);
int pbn_mi_stats(const pbn_mi* h, int64_t* device_passes, int64_t* host_passes);
/* ---- k-nearest-neighbour mutual information (learning/independences/continuous/mutual_information.{hpp,cpp}:
 * KMutualInformation).  cols: n_vars host columns of N values as double.  Ranks as rank_data (:17-52); MI by the KSG /
 * Frenzel-Pompe estimator on the ranks (mi_pair :9-43, mi_triple :45-116, mi_general :118-145) with brute-force
 * neighbour kernels instead of the kd-tree; p-values by `samples` permutations drawn exactly as the reference draws them
 * (:157-190, hpp:128-206; seeded std::mt19937). */
typedef struct pbn_kmi pbn_kmi;
int pbn_kmi_create(pbn_ctx* ctx, const double* const* cols, int n_vars, int64_t N, int k, uint32_t seed, int shuffle_neighbors,
                   int samples, pbn_kmi** out);
void pbn_kmi_destroy(pbn_kmi* h);
int pbn_kmi_value(pbn_kmi* h, int v1, int v2, int n_cond, const int* cond, double* mi);
double pbn_kmi_pvalue(void* user, int v1, int v2, int n_cond, const int* cond);   /* pbn_ci_pvalue_fn */

/* mmpc_all_variables (learning/algorithms/mmpc.cpp:910-966; forward / backward phases :356-644): candidate
 * parents-and-children of every variable.  Lists are pairs of node indices.  symmetric != 0 applies
 * remove_asymmetries (learning/algorithms/mmhc.cpp:12-22).  cpc_off: n+1 offsets into cpc (capacity n*(n-1)). */
int pbn_mmpc_cpcs(int n, pbn_ci_pvalue_fn fn, void* user, double alpha, int n_arc_whitelist, const int* arc_whitelist,
                  int n_edge_blacklist, const int* edge_blacklist, int n_edge_whitelist, const int* edge_whitelist,
                  int symmetric, int* cpc_off, int* cpc, int64_t* n_tests);

/* The same over a conditional graph (mmpc.cpp:875-908, 740-784, 984-993): the last n_interface of the n variables are
 * interface nodes - candidates of the nodes, never of each other. */
int pbn_mmpc_cpcs_conditional(int n, int n_interface, pbn_ci_pvalue_fn fn, void* user, double alpha, int n_arc_whitelist,
                              const int* arc_whitelist, int n_edge_blacklist, const int* edge_blacklist,
                              int n_edge_whitelist, const int* edge_whitelist, int symmetric, int* cpc_off, int* cpc,
                              int64_t* n_tests);

/* With a batched callback for the steps whose tests are mutually independent (marginal pass, forward extensions);
 * results are identical to the unbatched search.  batch_fn may be NULL. */
int pbn_mmpc_cpcs_batched(int n, int n_interface, pbn_ci_pvalue_fn fn, pbn_ci_pvalue_batch_fn batch_fn, void* user,
                          double alpha, int n_arc_whitelist, const int* arc_whitelist, int n_edge_blacklist,
                          const int* edge_blacklist, int n_edge_whitelist, const int* edge_whitelist, int symmetric,
                          int* cpc_off, int* cpc, int64_t* n_tests);

#ifdef __cplusplus
}
#endif
#endif /* PBN_HIP_H */
