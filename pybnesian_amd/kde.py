"""KDE / ProductKDE and bandwidth selectors — the reference's Python surface for this path
(/root/reference/pybnesian/pybindings/pybindings_kde.cpp:116-393) on top of the HIP C ABI.

Semantics follow kde/KDE.hpp:292-417,451-590 and kde/ProductKDE.hpp:153-308:
  fit   : bandwidth selector on the covariance of the (null-compacted) training rows, then the training
          rows are whitened and packed on device (pbn_kde_fit).
  logl  : per-row log-likelihood, NaN at rows with a null in any variable (KDE.hpp:527-546).
  slogl : sum over valid rows (KDE.hpp:549-562).
Train/test dtype mismatch raises ValueError (kde/KDE.cpp:74-76,92-94).
"""
import ctypes as C

import numpy as np
import pyarrow as pa

from . import _lib
from .dataset import DeviceTable, as_record_batch, default_context, same_type

SingularCovarianceData = _lib.SingularCovarianceData


def _covariance(ctx, rb, variables):
    """(table, mask, N, cov) of the null-compacted rows: DataFrame::cov (dataset.hpp:340-396) on device."""
    table, mask = DeviceTable.from_dataframe(ctx, rb, variables)
    n = table.num_rows
    if n <= 1:
        cov = np.full((len(variables), len(variables)), np.nan)
    else:
        cov = table.cov(variables)
    return table, mask, n, cov


class BandwidthSelector:
    """Subclassable like the reference trampoline (pybindings_kde.cpp:19-110): override
    `bandwidth(df, variables)` and/or `diag_bandwidth(df, variables)`."""

    def bandwidth(self, df, variables):
        raise NotImplementedError

    def diag_bandwidth(self, df, variables):
        raise NotImplementedError

    # internal fast path: selectors implemented by the library reuse the already-uploaded table
    _selector_id = None

    def _from_cov(self, kind, cov, n, dtype):
        d = cov.shape[0]
        out = np.zeros((d, d), order="F") if kind == _lib.PBN_BW_FULL else np.zeros(d)
        cov = np.asfortranarray(cov, dtype=np.float64)
        _lib.check(_lib.load().pbn_bandwidth(self._selector_id, kind, _lib.dptr(cov), d, int(n), dtype, _lib.dptr(out)))
        return out


class _LibrarySelector(BandwidthSelector):
    def _run(self, df, variables, kind):
        variables = list(variables)
        if not variables:
            return np.zeros((0, 0)) if kind == _lib.PBN_BW_FULL else np.zeros(0)
        rb = as_record_batch(df)
        ctx = default_context()
        table, _, n, cov = _covariance(ctx, rb, variables)
        return self._from_cov(kind, cov, n, table.dtype)

    def bandwidth(self, df, variables):
        return self._run(df, variables, _lib.PBN_BW_FULL)

    def diag_bandwidth(self, df, variables):
        return self._run(df, variables, _lib.PBN_BW_DIAG)


class NormalReferenceRule(_LibrarySelector):
    """kde/NormalReferenceRule.hpp:10-134."""

    _selector_id = _lib.PBN_SEL_NORMAL_REFERENCE

    def __str__(self):
        return "NormalReferenceRule"


class ScottsBandwidth(_LibrarySelector):
    """kde/ScottsBandwidth.hpp:10-117."""

    _selector_id = _lib.PBN_SEL_SCOTT

    def __str__(self):
        return "ScottsBandwidth"


class UCV(BandwidthSelector):
    """Unbiased cross-validation selector (kde/UCV.hpp:47-56, UCV.cpp:452-526): minimises N * UCV(H) from the normal
    reference bandwidth.  The pair sums of the objective are one self-sweep on the device per evaluation; the simplex
    search restates NLopt's LN_NELDERMEAD (the optimiser's trajectory is not pinned against a reference build)."""

    def __init__(self):
        self.last_evaluations = 0

    def _run(self, df, variables, kind):
        import ctypes as C

        variables = list(variables)
        if not variables:
            return np.zeros((0, 0)) if kind == _lib.PBN_BW_FULL else np.zeros(0)
        rb = as_record_batch(df)
        ctx = default_context()
        table, _, n, cov = _covariance(ctx, rb, variables)
        start = NormalReferenceRule()._from_cov(kind, cov, n, table.dtype)
        d = len(variables)
        out = np.zeros((d, d), order="F") if kind == _lib.PBN_BW_FULL else np.zeros(d)
        ev = C.c_int64(0)
        _lib.check(_lib.load().pbn_ucv_bandwidth(ctx.handle, table.handle, _lib.int_array(table.index(variables)), d, 0, n, kind,
                                                 _lib.dptr(np.asfortranarray(start)), _lib.dptr(out), C.byref(ev)))
        self.last_evaluations = ev.value
        return out

    def bandwidth(self, df, variables):
        return self._run(df, variables, _lib.PBN_BW_FULL)

    def diag_bandwidth(self, df, variables):
        return self._run(df, variables, _lib.PBN_BW_DIAG)

    def score(self, df, variables, bandwidth):
        """UCVScorer: N * UCV(bandwidth); a d x d matrix or a vector of d variances."""
        import ctypes as C

        variables = list(variables)
        rb = as_record_batch(df)
        ctx = default_context()
        table, _ = DeviceTable.from_dataframe(ctx, rb, variables)
        bw = np.asarray(bandwidth, dtype=np.float64)
        kind = _lib.PBN_BW_FULL if bw.ndim == 2 else _lib.PBN_BW_DIAG
        d = len(variables)
        if bw.shape != ((d, d) if bw.ndim == 2 else (d,)):
            raise ValueError(f"Wrong dimension for bandwidth. it should be a {d}x{d} matrix or a {d} vector.")
        out = C.c_double(0.0)
        _lib.check(_lib.load().pbn_ucv_score(ctx.handle, table.handle, _lib.int_array(table.index(variables)), d, 0, table.num_rows,
                                             _lib.dptr(np.asfortranarray(bw)), kind, C.byref(out)))
        return out.value

    def __str__(self):
        return "UCV"


class _KDEBase:
    _kind = _lib.PBN_BW_FULL
    _name = "KDE"

    def __init__(self, variables, bandwidth_selector=None):
        variables = list(variables)
        if not variables:
            raise ValueError("Cannot create a KDE model with 0 variables")  # KDE.hpp:296-298
        self._variables = variables
        self._selector = bandwidth_selector if bandwidth_selector is not None else NormalReferenceRule()
        self._fitted = False
        self._handle = None
        self._train = None
        self._dtype = None
        self._bandwidth = None
        self._N = 0
        self._on_bandwidth = None  # set by a CKDE that owns this KDE as its joint / marginal member

    # -- reference accessors ---------------------------------------------------------------------
    def variables(self):
        return list(self._variables)

    def num_instances(self):
        self._check_fitted()
        return self._N

    def num_variables(self):
        return len(self._variables)

    def fitted(self):
        return self._fitted

    def data_type(self):
        self._check_fitted()
        return pa.float64() if self._dtype == _lib.PBN_F64 else pa.float32()

    def dataset(self):
        self._check_fitted()
        vals = self._train.read(self._variables)
        return pa.RecordBatch.from_arrays([pa.array(np.ascontiguousarray(vals[:, i])) for i in range(vals.shape[1])],
                                          names=self._variables)

    @property
    def bandwidth(self):
        return self._bandwidth

    @bandwidth.setter
    def bandwidth(self, value):
        value = np.asarray(value, dtype=np.float64)
        d = len(self._variables)
        if self._kind == _lib.PBN_BW_FULL and value.shape != (d, d):
            raise ValueError("The bandwidth matrix must be a square matrix with shape (%d, %d)" % (d, d))
        if self._kind == _lib.PBN_BW_DIAG and value.shape != (d,):
            raise ValueError("The bandwidth vector must have %d elements" % d)
        self._bandwidth = value.copy()
        if self._fitted:
            self._device_fit()
            if self._on_bandwidth is not None:
                self._on_bandwidth()

    def _check_fitted(self):
        if not self._fitted:
            raise ValueError(f"{self._name} factor not fitted.")

    # -- fit ---------------------------------------------------------------------------------------
    def _select_bandwidth(self, rb, table, n, dtype):
        sel = self._selector
        if isinstance(sel, _LibrarySelector):
            d = len(self._variables)
            if n <= (1 if (self._kind == _lib.PBN_BW_DIAG and isinstance(sel, ScottsBandwidth)) else d):
                cov = np.eye(d)  # value unused: the library raises SingularCovarianceData on n <= d
            else:
                cov = table.cov(self._variables)
            return sel._from_cov(self._kind, cov, n, dtype)
        if self._kind == _lib.PBN_BW_FULL:
            return np.asarray(sel.bandwidth(rb, self._variables), dtype=np.float64)
        return np.asarray(sel.diag_bandwidth(rb, self._variables), dtype=np.float64)

    def fit(self, df):
        rb = as_record_batch(df)
        dtype = same_type(rb, self._variables)
        ctx = default_context()
        table, _mask = DeviceTable.from_dataframe(ctx, rb, self._variables)
        n = table.num_rows
        self._bandwidth = self._select_bandwidth(rb, table, n, dtype)
        self._train, self._dtype, self._N = table, dtype, n
        self._train_idx = table.index(self._variables)
        self._device_fit()
        self._fitted = True

    def fit_table(self, table):
        """Fit from a device-resident DeviceTable holding (at least) this model's variables: no host round
        trip of the training rows.  Python-subclassed selectors need a host DataFrame and are rejected."""
        if not isinstance(self._selector, _LibrarySelector):
            raise ValueError("fit_table needs a library bandwidth selector")
        n = table.num_rows
        self._train_idx = table.index(self._variables)
        self._bandwidth = self._select_bandwidth(None, table, n, table.dtype)
        self._train, self._dtype, self._N = table, table.dtype, n
        self._device_fit()
        self._fitted = True

    def _device_fit(self):
        lib = _lib.load()
        if self._handle is not None:
            lib.pbn_kde_destroy(self._handle)
            self._handle = None
        h = C.c_void_p()
        d = len(self._variables)
        bw = np.asfortranarray(self._bandwidth, dtype=np.float64)
        _lib.check(lib.pbn_kde_fit(self._train.ctx.handle, self._train.handle, _lib.int_array(self._train_idx), d, 0, self._N,
                                   _lib.dptr(bw), self._kind, None, C.byref(h)))
        self._handle = h

    # -- evaluation ------------------------------------------------------------------------------------
    def _upload_test(self, df):
        self._check_fitted()
        rb = as_record_batch(df)
        dtype = same_type(rb, self._variables)
        if dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        table, mask = DeviceTable.from_dataframe(self._train.ctx, rb, self._variables)
        return rb, table, mask

    def logl(self, df):
        rb, table, mask = self._upload_test(df)
        m = table.num_rows
        vals = np.empty(m, dtype=np.float64)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_kde_logl(self._handle, table.handle, _lib.int_array(table.index(self._variables)), 0, m, _lib.dptr(vals)))
        if mask is None:
            return vals
        out = np.full(rb.num_rows, np.nan)
        out[mask] = vals
        return out

    def slogl(self, df):
        _, table, _ = self._upload_test(df)
        res = C.c_double(0.0)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_kde_slogl(self._handle, table.handle, _lib.int_array(table.index(self._variables)), 0, table.num_rows, C.byref(res)))
        return res.value

    # device-resident entry used by bench.py / scores: no host round trip of the test rows
    def slogl_table(self, table, names=None, row0=0, n=None):
        self._check_fitted()
        if table.dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        idx = table.index(self._variables if names is None else names)
        n = table.num_rows - row0 if n is None else n
        res = C.c_double(0.0)
        _lib.check(_lib.load().pbn_kde_slogl(self._handle, table.handle, _lib.int_array(idx), row0, n, C.byref(res)))
        return res.value

    def logl_table(self, table, names=None, row0=0, n=None):
        """Per-row logl of rows [row0, row0 + n) of a device-resident table (the bench's parity block, the full-size tests)."""
        self._check_fitted()
        if table.dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        idx = table.index(self._variables if names is None else names)
        n = table.num_rows - row0 if n is None else n
        vals = np.empty(n, dtype=np.float64)
        _lib.check(_lib.load().pbn_kde_logl(self._handle, table.handle, _lib.int_array(idx), row0, n, _lib.dptr(vals)))
        return vals

    def slogl_table_async(self, table, dev_out_ptr, names=None, row0=0, n=None):
        """Enqueue one slogl on the context stream; the scalar lands at DEVICE address dev_out_ptr."""
        self._check_fitted()
        if table.dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        idx = table.index(self._variables if names is None else names)
        n = table.num_rows - row0 if n is None else n
        _lib.check(_lib.load().pbn_kde_slogl_async(self._handle, table.handle, _lib.int_array(idx), row0, n, C.c_void_p(dev_out_ptr)))

    # -- pickle: the tuple layout of KDE::__getstate__ / ProductKDE::__getstate__ (kde/KDE.hpp:642-666, ProductKDE.hpp:310-334),
    #    (variables, fitted, bandwidth selector, bandwidth, training data, lognorm_const, N, arrow type id)
    # with the reference's placeholders for an unfitted model (empty bandwidth / data, -1.0, -1, -1).  KDE: bandwidth is the
    # d x d matrix, training data ONE flat column-major vector of N * d values in the data's dtype; ProductKDE: bandwidth the
    # d vector, training data a list of d column vectors (the reference forgets to append its columns to that list - its
    # pickled ProductKDE cannot be restored, ProductKDE.hpp:323-326 - here the list is filled).  The training matrix is
    # read back from HBM.
    _ARROW_TYPE_ID = {_lib.PBN_F64: 12, _lib.PBN_F32: 11}   # arrow::Type::DOUBLE / FLOAT, as static_cast<int>(type->id())

    def __getstate__(self):
        d = len(self._variables)
        if not self._fitted:
            empty_bw = np.zeros((0, 0)) if self._kind == _lib.PBN_BW_FULL else np.zeros(0)
            empty_data = np.zeros(0) if self._kind == _lib.PBN_BW_FULL else []
            return (list(self._variables), False, self._selector, empty_bw, empty_data, -1.0, -1, -1)
        np_t = np.float64 if self._dtype == _lib.PBN_F64 else np.float32
        vals = np.asarray(self._train.read([self._train.names[i] for i in self._train_idx]), dtype=np_t)
        if self._kind == _lib.PBN_BW_FULL:
            training = np.asfortranarray(vals).reshape(-1, order="F")
        else:
            training = [np.ascontiguousarray(vals[:, i]) for i in range(d)]
        return (list(self._variables), True, self._selector, np.array(self._bandwidth), training,
                float(_lib.load().pbn_kde_lognorm(self._handle, 0)), int(self._N), self._ARROW_TYPE_ID[self._dtype])

    def __setstate__(self, state):
        if isinstance(state, dict):      # states written before the reference layout was adopted
            state = (state["variables"], state["fitted"], state["selector"], state.get("bandwidth"),
                     state.get("training"), -1.0, state.get("N", -1), self._ARROW_TYPE_ID.get(state.get("dtype"), -1))
        if len(state) != 8:
            raise RuntimeError(f"Not valid {self._name}.")          # KDE.cpp:121
        variables, fitted, selector, bw, training, _lognorm, n, type_id = state
        self.__init__(variables, selector)
        if fitted:
            d, n = len(self._variables), int(n)
            if type_id not in (11, 12):
                raise RuntimeError(f"Not valid data type in {self._name}.")
            np_t = np.float64 if type_id == 12 else np.float32
            if self._kind == _lib.PBN_BW_FULL:
                vals = np.asarray(training, dtype=np_t).reshape(n, d, order="F")
                cols = {v: pa.array(np.ascontiguousarray(vals[:, i])) for i, v in enumerate(self._variables)}
            else:
                cols = {v: pa.array(np.ascontiguousarray(np.asarray(training[i], dtype=np_t))) for i, v in enumerate(self._variables)}
            rb = pa.RecordBatch.from_pydict(cols)
            table, _ = DeviceTable.from_dataframe(default_context(), rb, self._variables, drop_null=False)
            self._train, self._dtype, self._N = table, (_lib.PBN_F64 if type_id == 12 else _lib.PBN_F32), n
            self._train_idx = list(range(d))
            self._bandwidth = np.array(bw, dtype=np.float64)
            self._device_fit()          # lognorm_const is recomputed from the bandwidth (the reference trusts the stored one)
            self._fitted = True

    def __del__(self):
        try:
            if not _lib.alive():
                return
            if getattr(self, "_handle", None) is not None:
                _lib.load().pbn_kde_destroy(self._handle)
                self._handle = None
        except Exception:
            pass

    def __str__(self):
        return f"{self._name}({', '.join(self._variables)})"


class KDE(_KDEBase):
    """Full-bandwidth Gaussian KDE (kde/KDE.hpp:292-417)."""

    _kind = _lib.PBN_BW_FULL
    _name = "KDE"


class ProductKDE(_KDEBase):
    """Diagonal-bandwidth ("product") Gaussian KDE (kde/ProductKDE.hpp:18-150)."""

    _kind = _lib.PBN_BW_DIAG
    _name = "ProductKDE"
