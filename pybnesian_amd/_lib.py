"""ctypes binding of libpbn_hip.so (the C ABI in include/pbn_hip.h).

There is deliberately no CPU fallback: if the shared library is missing, or no MI355X is visible when a
device context is requested, the call raises.  Error codes map onto the exception classes the reference
raises for this path: ValueError, SingularCovarianceData(ValueError), RuntimeError.
"""
import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# PBN_LIB: another build of the SAME library (the host-sanitizer build of `make asan`, tools/asan_cpu.sh); never a fallback
LIB_PATH = os.environ.get("PBN_LIB") or os.path.join(_HERE, "libpbn_hip.so")

PBN_OK, PBN_ERR_INVALID, PBN_ERR_SINGULAR, PBN_ERR_DEVICE = 0, 1, 2, 3
PBN_F64, PBN_F32 = 0, 1
PBN_BW_FULL, PBN_BW_DIAG = 0, 1
PBN_SEL_NORMAL_REFERENCE, PBN_SEL_SCOTT = 0, 1
PBN_K_PACK, PBN_K_SWEEP, PBN_K_FINISH, PBN_K_GRAM, PBN_K_MOMENT = 0, 1, 2, 3, 4
PBN_SPLIT_NONE, PBN_SPLIT_CV, PBN_SPLIT_HOLDOUT, PBN_SPLIT_VALIDATED = 0, 1, 2, 3
PBN_SCORE_BIC, PBN_SCORE_BGE, PBN_SCORE_CVLIK, PBN_SCORE_HOLDOUT = 0, 1, 2, 3
PBN_NODE_LG, PBN_NODE_CKDE, PBN_NODE_DISCRETE = 0, 1, 2
PBN_BN_GAUSSIAN, PBN_BN_SEMIPARAMETRIC, PBN_BN_KDE, PBN_BN_CLG = 0, 1, 2, 3


class SingularCovarianceData(ValueError):
    """pybnesian.SingularCovarianceData (pybindings_kde.cpp:114): a ValueError subclass."""


_vp = C.c_void_p
_i64 = C.c_int64
_int = C.c_int
_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol declared in include/pbn_hip.h
SIGNATURES = {
    "pbn_last_error": (C.c_char_p, []),
    "pbn_version": (C.c_char_p, []),
    "pbn_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "pbn_ctx_destroy": (None, [_vp]),
    "pbn_ctx_sync": (_int, [_vp]),
    "pbn_ctx_stream": (_vp, [_vp]),
    "pbn_ctx_set_profiling": (_int, [_vp, _int]),
    "pbn_ctx_kernel_time": (_int, [_vp, _int, _dp, C.POINTER(_i64)]),
    "pbn_table_create": (_int, [_vp, C.POINTER(_vp), _int, _i64, _int, _vp, _i64, C.POINTER(_vp)]),
    "pbn_table_from_device": (_int, [_vp, _vp, _i64, _int, _i64, _int, C.POINTER(_vp)]),
    "pbn_table_destroy": (None, [_vp]),
    "pbn_table_rows": (_i64, [_vp]),
    "pbn_table_cols": (_int, [_vp]),
    "pbn_table_take": (_int, [_vp, _vp, _i64, C.POINTER(_vp)]),
    "pbn_table_read": (_int, [_vp, _ip, _int, _vp]),
    "pbn_table_sse": (_int, [_vp, _ip, _int, _i64, _i64, _dp, _dp]),
    "pbn_bandwidth": (_int, [_int, _int, _dp, _int, _i64, _int, _dp]),
    "pbn_ucv_score": (_int, [_vp, _vp, _ip, _int, _i64, _i64, _dp, _int, _dp]),
    "pbn_ucv_bandwidth": (_int, [_vp, _vp, _ip, _int, _i64, _i64, _int, _dp, _dp, C.POINTER(_i64)]),
    "pbn_kde_fit": (_int, [_vp, _vp, _ip, _int, _i64, _i64, _dp, _int, _dp, C.POINTER(_vp)]),
    "pbn_ckde_fit": (_int, [_vp, _vp, _ip, _int, _i64, _i64, _dp, _dp, C.POINTER(_vp)]),
    "pbn_kde_destroy": (None, [_vp]),
    "pbn_kde_num_instances": (_i64, [_vp]),
    "pbn_kde_lognorm": (C.c_double, [_vp, _int]),
    "pbn_kde_logl": (_int, [_vp, _vp, _ip, _i64, _i64, _dp]),
    "pbn_kde_logl_dev": (_int, [_vp, _vp, _ip, _i64, _i64, _vp]),
    "pbn_ckde_cdf": (_int, [_vp, _vp, _ip, _i64, _i64, _dp]),
    "pbn_ckde_sample": (_int, [_vp, _i64, _i64, _vp, _ip, C.c_uint32, _vp]),
    "pbn_lg_sample": (_int, [_i64, _dp, _int, C.c_double, C.c_uint32, C.POINTER(_vp), _int, _dp]),
    "pbn_discrete_sample": (_int, [_i64, _dp, _int, _i64, _ip, C.c_uint32, _ip]),
    "pbn_kde_slogl": (_int, [_vp, _vp, _ip, _i64, _i64, _dp]),
    "pbn_kde_slogl_async": (_int, [_vp, _vp, _ip, _i64, _i64, _vp]),
    "pbn_scoredata_create": (_int, [_vp, _vp, _int, _int, C.c_uint32, C.c_double, C.POINTER(_vp)]),
    "pbn_scoredata_create_sharded": (_int, [_vp, _vp, _int, _int, C.c_uint32, C.c_double, _int, _int, C.POINTER(_vp)]),
    "pbn_scoredata_moments": (_int, [_vp, _dp, C.POINTER(_i64), _int]),
    "pbn_scoredata_set_selector": (_int, [_vp, _int]),
    "pbn_scoredata_set_precise": (_int, [_vp, _int]),
    "pbn_scoredata_cache_stats": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pbn_scoredata_destroy": (None, [_vp]),
    "pbn_scoredata_set_discrete": (_int, [_vp, _int, C.POINTER(_vp), _ip]),
    "pbn_scoredata_set_validity": (_int, [_vp, C.POINTER(_vp)]),
    "pbn_split_layout": (_int, [_i64, _int, _int, C.c_uint32, C.c_double, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pbn_scoredata_layout": (_int, [_vp, _vp, _vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pbn_lg_fit": (_int, [_vp, _int, _ip, _int, _dp, _dp]),
    "pbn_lg_fit_table": (_int, [_vp, _ip, _int, _i64, _i64, _dp, _dp]),
    "pbn_lg_logl": (_int, [_vp, _ip, _int, _i64, _i64, _dp, C.c_double, _dp, _dp]),
    "pbn_lg_cdf": (_int, [_vp, _ip, _int, _i64, _i64, _dp, C.c_double, _dp]),
    "pbn_score_batch": (_int, [_vp, _int, _int, _ip, _ip, _ip, _ip, _dp, _int, _dp]),
    "pbn_score_batch_parts": (_int, [_vp, _int, _int, _ip, _ip, _ip, _ip, _int, _int, _dp]),
    "pbn_score_terms": (_int, [_vp, _int, _int, _ip, _ip, _ip, _dp]),
    "pbn_score_term_regions": (_int, [_vp, _int, _int, _ip, _ip, _ip, _ip, _dp]),
    "pbn_score_terms_put": (_int, [_vp, _int, _int, _ip, _ip, _ip, _dp]),
    "pbn_score_terms_missing": (_int, [_vp, _int, _int, _ip, _ip, _ip, _ip]),
    "pbn_lincor_create": (_int, [_vp, _vp, C.POINTER(_vp)]),
    "pbn_lincor_from_cov": (_int, [_int, _i64, _dp, C.POINTER(_vp)]),
    "pbn_lincor_destroy": (None, [_vp]),
    "pbn_lincor_cov": (_int, [_vp, _dp]),
    "pbn_lincor_pvalue": (C.c_double, [_vp, _int, _int, _int, _ip]),
    "pbn_mi_create": (_int, [_vp, _vp, _i64, _int, C.POINTER(_vp), _ip, _int, C.POINTER(_vp)]),
    "pbn_mi_destroy": (None, [_vp]),
    "pbn_mi_value": (_int, [_vp, _int, _int, _int, _ip, _dp, _dp]),
    "pbn_mi_pvalue": (C.c_double, [_vp, _int, _int, _int, _ip]),
    "pbn_chisq_pvalue": (C.c_double, [_vp, _int, _int, _int, _ip]),
    "pbn_mi_set_order": (_int, [_vp, _int, _ip]),
    "pbn_mi_stats": (_int, [_vp, C.POINTER(_i64), C.POINTER(_i64)]),
    "pbn_mi_set_continuous_nulls": (_int, [_vp, C.POINTER(C.c_ubyte), _dp]),
    "pbn_mi_lincor_pvalue": (C.c_double, [_vp, _int, _int, _int, _ip]),
    "pbn_mi_counts": (_int, [_vp, _int, _ip, _dp]),
    "pbn_kmi_create": (_int, [_vp, C.POINTER(_vp), _int, _i64, _int, C.c_uint32, _int, _int, C.POINTER(_vp)]),
    "pbn_kmi_destroy": (None, [_vp]),
    "pbn_kmi_value": (_int, [_vp, _int, _int, _int, _ip, _dp]),
    "pbn_kmi_pvalue": (C.c_double, [_vp, _int, _int, _int, _ip]),
    "pbn_mmpc_cpcs": (_int, [_int, _vp, _vp, C.c_double, _int, _ip, _int, _ip, _int, _ip, _int, _ip, _ip, C.POINTER(_i64)]),
    "pbn_mmpc_cpcs_conditional": (_int, [_int, _int, _vp, _vp, C.c_double, _int, _ip, _int, _ip, _int, _ip, _int, _ip, _ip, C.POINTER(_i64)]),
    "pbn_mmpc_cpcs_batched": (_int, [_int, _int, _vp, _vp, _vp, C.c_double, _int, _ip, _int, _ip, _int, _ip, _int, _ip, _ip, C.POINTER(_i64)]),
    "pbn_mi_pvalue_batch": (None, [_vp, _int, _ip, _ip, _ip, _ip, _dp]),
    "pbn_hc_estimate": (_int, [_vp, _vp, _vp, _ip, _ip, _ip, _vp]),
    "pbn_hc_create": (_int, [_vp, _vp, _vp, C.POINTER(_vp)]),
    "pbn_hc_destroy": (None, [_vp]),
    "pbn_hc_set_model": (_int, [_vp, _int, _ip, _ip]),
    "pbn_hc_cache_scores": (_int, [_vp]),
    "pbn_hc_find_max": (_int, [_vp, _int, _ip, _ip, _dp]),
    "pbn_hc_update_scores": (_int, [_vp, _int, _ip]),
    "pbn_hc_get": (_int, [_vp, _dp, _dp, _dp]),
    "pbn_scoredata_set_comm": (_int, [_vp, _vp]),
    "pbn_scoredata_reduce_moments": (_int, [_vp, _vp]),
    "pbn_kde_slogl_sharded": (_int, [_vp, _vp, _ip, _i64, _i64, _vp, _dp]),
    "pbn_shard_batch": (_int, [_vp, _vp, _int, _int, _ip, _ip, _ip, _ip, _int, _dp]),
    "pbn_shard_deal": (_int, [_int, _dp, C.POINTER(C.c_uint32), _int, _dp, _ip]),
    "pbn_shard_term_cost": (C.c_double, [_int, _i64, _i64]),
}

HC_SCORE_FN = C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _ip, _dp)
CI_PVALUE_FN = C.CFUNCTYPE(C.c_double, _vp, _int, _int, _int, _ip)
CI_BATCH_FN = C.CFUNCTYPE(None, _vp, _int, _ip, _ip, _ip, _ip, _dp)
HC_ITER_FN = C.CFUNCTYPE(_int, _vp, _int, _ip, C.c_double, _int, _ip, _ip)
ALLGATHER_FN = C.CFUNCTYPE(_int, _vp, _dp, _i64, _dp)   # pbn_allgather_fn


class Comm(C.Structure):   # pbn_comm
    _fields_ = [("rank", _int), ("world", _int), ("all_gather", ALLGATHER_FN), ("user", _vp)]


class ShardEngine(C.Structure):   # pbn_shard_engine
    _fields_ = [
        ("user", _vp), ("n_cont", _int),
        ("shape", C.CFUNCTYPE(_int, _vp, _int, _ip, C.POINTER(_i64), C.POINTER(_i64))),
        ("batch", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _ip, _dp)),
        ("terms_missing", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _ip)),
        ("terms", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _dp)),
        ("term_regions", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _ip, _dp)),
        ("terms_put", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _dp)),
        ("batch_parts", C.CFUNCTYPE(_int, _vp, _int, _int, _ip, _ip, _ip, _ip, _int, _int, _dp)),
        ("term_price", C.CFUNCTYPE(C.c_double, _vp, _int, _int)),
    ]


class HCConfig(C.Structure):
    _fields_ = [
        ("n_nodes", _int), ("bn_type", _int), ("node_types", _ip), ("n_arcs", _int), ("arcs", _ip),
        ("n_arc_blacklist", _int), ("arc_blacklist", _ip), ("n_arc_whitelist", _int), ("arc_whitelist", _ip),
        ("n_type_blacklist", _int), ("type_blacklist", _ip), ("n_type_whitelist", _int), ("type_whitelist", _ip),
        ("op_arcs", _int), ("op_node_type", _int), ("arcs_first", _int), ("max_indegree", _int), ("max_iters", _int),
        ("epsilon", C.c_double), ("patience", _int), ("validated", _int),
        ("on_iter", HC_ITER_FN), ("on_iter_user", _vp), ("n_interface", _int), ("near_tie_abs", C.c_double),
    ]


class HCStats(C.Structure):
    _fields_ = [
        ("iterations", _int), ("cells_scored", _i64), ("local_score_evals", _i64), ("trace_capacity", _int),
        ("trace", _ip), ("trace_delta", _dp), ("trace_len", _int), ("near_tie_redos", _i64),
    ]


_lib = None
_alive = True


def _shutdown():
    # after this point (interpreter exit) destructors must not call into the HIP runtime any more
    global _alive
    _alive = False


import atexit  # noqa: E402

atexit.register(_shutdown)


def alive():
    return _alive


def load():
    """Load libpbn_hip.so and declare the prototypes.  Raises RuntimeError if it has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError(
            f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc, gfx950).  pybnesian_amd has no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    return lib


def check(status):
    if status == PBN_OK:
        return
    msg = load().pbn_last_error().decode("utf-8", "replace")
    if status == PBN_ERR_SINGULAR:
        raise SingularCovarianceData(msg)
    if status == PBN_ERR_INVALID:
        raise ValueError(msg)
    raise RuntimeError(msg)


def int_array(values):
    arr = (C.c_int * len(values))(*[int(v) for v in values])
    return arr


def dptr(a):
    """double* view of a C-contiguous float64 numpy array."""
    return a.ctypes.data_as(_dp)
