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
void pbn_ctx_destroy(pbn_ctx* ctx);
int pbn_ctx_sync(pbn_ctx* ctx);
void* pbn_ctx_stream(pbn_ctx* ctx); /* hipStream_t the kernels are launched on (for event timing)    */
/* Per-kernel-class device timing with HIP events on the context stream (the reference has no profiling
 * queue, opencl_config.cpp:175).  Classes: 0 pack, 1 KDE sweep, 2 finish/reduce, 3 Gram/SSE.
 * pbn_ctx_kernel_time synchronises the stream and returns the accumulated ms and launch count. */
#define PBN_NUM_KERNEL_CLASSES 8
typedef enum { PBN_K_PACK = 0, PBN_K_SWEEP = 1, PBN_K_FINISH = 2, PBN_K_GRAM = 3 } pbn_kernel_class;
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
 * it; it only keeps the whitened coordinates small); NULL lets the library take pilot means. */
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

/* logl: replaces KDE::_logl / ProductKDE::_logl / CKDE::_logl (KDE.hpp:513-547,
 * ProductKDE.hpp:192-238, CKDE.hpp:202-254).  Writes n doubles to HOST `out` (dev_out: DEVICE). */
int pbn_kde_logl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out);
int pbn_kde_logl_dev(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                     double* dev_out);
/* slogl: replaces KDE::_slogl / ProductKDE::_slogl / CKDE::_slogl (KDE.hpp:549-562,
 * ProductKDE.hpp:295-308, CKDE.hpp:256-287): sum of logl over the rows, one scalar read-back. */
int pbn_kde_slogl(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, double* out);
/* Enqueue-only form for benchmarking / batching: result lands in DEVICE memory, no sync. */
int pbn_kde_slogl_async(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n,
                        double* dev_out);

#ifdef __cplusplus
}
#endif
#endif /* PBN_HIP_H */
