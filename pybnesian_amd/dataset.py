"""Host-side DataFrame handling + device tables.

Mirrors the part of the reference's `dataset::DataFrame` that the hot path needs
(/root/reference/pybnesian/dataset/dataset.hpp:1953-2085, dataset.cpp:53-56,208-271): inputs are
`pyarrow.RecordBatch` or `pandas.DataFrame`; continuous columns must all be float64 or all float32;
nulls are Arrow validity bits and rows with a null in any requested column are compacted away before
upload (dataset.hpp:92-106).  The Arrow column buffers are handed to the C ABI as plain pointers.
"""
import ctypes as C

import numpy as np
import pyarrow as pa

from . import _lib

try:  # pandas is optional at import time
    import pandas as pd
except Exception:  # pragma: no cover
    pd = None


def as_record_batch(df):
    if isinstance(df, pa.RecordBatch):
        return df
    if isinstance(df, pa.Table):
        batches = df.combine_chunks().to_batches()
        if len(batches) != 1:
            raise ValueError("Expected a single-chunk table.")
        return batches[0]
    if pd is not None and isinstance(df, pd.DataFrame):
        # dataset.cpp:53-56: RecordBatch.from_pandas(df, None, False)
        # small frames convert on the calling thread: starting pyarrow's conversion pool costs ~0.7 ms, more than a whole fit of a
        # 5-node network over 10k rows
        return pa.RecordBatch.from_pandas(df, preserve_index=False, nthreads=1 if df.shape[0] * max(df.shape[1], 1) < (1 << 22) else None)
    if isinstance(df, dict):
        return pa.RecordBatch.from_pydict({k: pa.array(v) for k, v in df.items()})
    raise TypeError("Expected a pandas.DataFrame or pyarrow.RecordBatch.")


def _column(rb, name):
    idx = rb.schema.get_field_index(name)
    if idx < 0:
        raise KeyError(f"Column {name} not present in DataFrame.")
    return rb.column(idx)


def column_values(arr):
    """Zero-copy numpy view of a primitive Arrow array's data buffer (nulls hold arbitrary values)."""
    if pa.types.is_float64(arr.type):
        dt = np.float64
    elif pa.types.is_float32(arr.type):
        dt = np.float32
    else:
        raise ValueError("Wrong data type. [double] or [float] data is expected.")
    buf = arr.buffers()[1]
    n = len(arr)
    if n == 0 or buf is None:
        return np.empty(0, dtype=dt)
    v = np.frombuffer(buf, dtype=dt, count=arr.offset + n)
    return v[arr.offset:]


def validity_mask(arr):
    """Boolean numpy mask (True = valid) or None when the array has no nulls."""
    if arr.null_count == 0:
        return None
    buf = arr.buffers()[0]
    bits = np.unpackbits(np.frombuffer(buf, dtype=np.uint8), bitorder="little")
    return bits[arr.offset: arr.offset + len(arr)].astype(bool)


def same_type(rb, variables):
    """DataFrame::same_type (dataset.cpp:253-271)."""
    types = {_column(rb, v).type for v in variables}
    if len(types) != 1:
        raise ValueError("All the variables must have the same data type.")
    t = types.pop()
    if pa.types.is_float64(t):
        return _lib.PBN_F64
    if pa.types.is_float32(t):
        return _lib.PBN_F32
    raise ValueError("Wrong data type. [double] or [float] data is expected.")


def combined_mask(rb, variables):
    """AND of the validity bitmaps of `variables` (dataset.cpp:208-235); None if no nulls."""
    mask = None
    for v in variables:
        m = validity_mask(_column(rb, v))
        if m is not None:
            mask = m if mask is None else (mask & m)
    return mask


class Context:
    """One device + one stream; the analogue of the reference's OpenCLConfig singleton
    (opencl/opencl_config.cpp:149-220), but explicit so that several GPUs can be driven."""

    def __init__(self, device=0):
        lib = _lib.load()
        h = C.c_void_p()
        _lib.check(lib.pbn_ctx_create(int(device), C.byref(h)))
        self.handle = h
        self.device = int(device)

    def sync(self):
        _lib.check(_lib.load().pbn_ctx_sync(self.handle))

    @property
    def stream(self):
        return _lib.load().pbn_ctx_stream(self.handle)

    def set_profiling(self, on=True):
        """True / 1: per-class kernel timing with the score engine on one stream; 2: timing only, behaviour unchanged; False: off."""
        _lib.check(_lib.load().pbn_ctx_set_profiling(self.handle, int(on)))

    def kernel_time(self, kernel_class):
        """(total_ms, launches) of a kernel class (0 pack, 1 sweep, 2 finish, 3 gram) since profiling was enabled."""
        ms, n = C.c_double(0.0), C.c_int64(0)
        _lib.check(_lib.load().pbn_ctx_kernel_time(self.handle, int(kernel_class), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def __del__(self):
        try:
            if not _lib.alive():
                return
            if getattr(self, "handle", None):
                _lib.load().pbn_ctx_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


_default_ctx = {}


def default_context(device=None):
    import os

    if device is None:
        device = int(os.environ.get("PBN_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class DeviceTable:
    """Column-major table resident in HBM (pbn_table)."""

    def __init__(self, ctx, handle, names, dtype, keepalive=None):
        self.ctx = ctx
        self.handle = handle
        self.names = list(names)
        self.dtype = dtype
        self._keepalive = keepalive

    @classmethod
    def from_dataframe(cls, ctx, df, variables, drop_null=True):
        """Upload `variables` of df.  Returns (table, mask) where mask is the boolean validity mask
        used for compaction (None when there were no nulls).  Inside a `shared_upload(ctx, df)` scope the table of that scope is
        returned instead when it holds every requested column - address its columns with `table.index(variables)`."""
        rb = as_record_batch(df)
        variables = list(variables)
        shared = getattr(_shared, "entry", None)
        if shared is not None and rb is shared[0] and ctx is shared[1] and shared[2] is not None and all(v in shared[3] for v in variables):
            return shared[2], None
        dtype = same_type(rb, variables)
        mask = combined_mask(rb, variables) if drop_null else None
        arrays = [column_values(_column(rb, v)) for v in variables]
        n = rb.num_rows
        ptrs = (C.c_void_p * len(arrays))(*[a.ctypes.data if a.size else None for a in arrays])
        bitmap = None
        bm_ptr = None
        if mask is not None:
            bitmap = np.packbits(mask, bitorder="little")
            bm_ptr = bitmap.ctypes.data
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_table_create(ctx.handle, ptrs, len(arrays), n, dtype, bm_ptr, 0, C.byref(h)))
        return cls(ctx, h, variables, dtype), mask

    @classmethod
    def from_device_pointer(cls, ctx, ptr, ld, names, n_rows, dtype, keepalive=None):
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_table_from_device(ctx.handle, C.c_void_p(ptr), ld, len(names), n_rows, dtype, C.byref(h)))
        return cls(ctx, h, names, dtype, keepalive)

    @property
    def num_rows(self):
        return int(_lib.load().pbn_table_rows(self.handle))

    def index(self, names):
        return [self.names.index(n) for n in names]

    def take(self, rows):
        rows = np.ascontiguousarray(rows, dtype=np.int32)
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_table_take(self.handle, rows.ctypes.data, rows.size, C.byref(h)))
        return DeviceTable(self.ctx, h, self.names, self.dtype)

    def read(self, names=None):
        names = self.names if names is None else list(names)
        n = self.num_rows
        out = np.empty((len(names), n), dtype=np.float64 if self.dtype == _lib.PBN_F64 else np.float32)
        _lib.check(_lib.load().pbn_table_read(self.handle, _lib.int_array(self.index(names)), len(names), out.ctypes.data))
        return out.T  # (n, len(names)) column-major view

    def sse(self, names, row0=0, n=None):
        """(means, sse) over rows [row0, row0+n) of the named columns: DataFrame::means / ::sse."""
        idx = self.index(names)
        n = self.num_rows - row0 if n is None else n
        d = len(idx)
        means = np.zeros(d)
        sse = np.zeros((d, d), order="F")
        _lib.check(_lib.load().pbn_table_sse(self.handle, _lib.int_array(idx), d, row0, n, _lib.dptr(means), _lib.dptr(sse)))
        return means, sse

    def cov(self, names, row0=0, n=None):
        n_ = self.num_rows - row0 if n is None else n
        means, sse = self.sse(names, row0, n)
        return sse / (n_ - 1)

    def __del__(self):
        try:
            if not _lib.alive():
                return
            if getattr(self, "handle", None):
                _lib.load().pbn_table_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


# ---- dataset::CrossValidation / HoldOut (dataset/crossvalidation_adaptator.hpp, holdout_adaptator.hpp) -----------------
def _valid_rows(rb, include_null):
    """Row ids the reference splits: all rows, or (default) the rows without a null in any column."""
    import numpy as np

    n = rb.num_rows
    if include_null:
        return np.arange(n, dtype=np.int64)
    mask = None
    for i in range(rb.num_columns):
        m = validity_mask(rb.column(i))
        if m is not None:
            mask = m if mask is None else (mask & m)
    return np.arange(n, dtype=np.int64) if mask is None else np.nonzero(mask)[0].astype(np.int64)


def _layout(n, split, k, seed, ratio):
    import ctypes as C

    import numpy as np

    from . import _lib

    perm = np.zeros(max(n, 1), dtype=np.int32)
    limits = np.zeros(max(k, 0) + 1, dtype=np.int32)
    n_cv, n_hold = C.c_int64(0), C.c_int64(0)
    _lib.check(_lib.load().pbn_split_layout(n, split, int(k), C.c_uint32(int(seed)), float(ratio), perm.ctypes.data,
                                            limits.ctypes.data if k > 1 else None, C.byref(n_cv), C.byref(n_hold)))
    return perm[:n], limits, n_cv.value, n_hold.value


_shared = __import__("threading").local()


@__import__("contextlib").contextmanager
def shared_upload(ctx, df, columns=None):
    """One upload for many factors: inside the scope, `DeviceTable.from_dataframe(ctx, df, variables)` hands out ONE table
    holding the null-free floating-point columns of `df` (all of one type; only those named in `columns` when given - a model's
    own nodes: a 4-node network on a 500-column frame must not upload, and keep alive, the other 496) instead of uploading
    `variables` again - the model-level
    fit / logl / slogl of a network make one PCIe pass over the table instead of one per factor (a 64-node network over 2M rows
    uploaded ~3 GB for a 1 GB table).  Columns with nulls, dictionary columns and mixed float types keep the per-factor upload (their
    row sets differ from factor to factor).  `df` must be the very RecordBatch the factors are given."""
    import pyarrow as pa

    rb = as_record_batch(df)
    prev = getattr(_shared, "entry", None)
    wanted = None if columns is None else set(columns)
    cols = [f.name for i, f in enumerate(rb.schema) if (pa.types.is_float64(f.type) or pa.types.is_float32(f.type)) and rb.column(i).null_count == 0
            and (wanted is None or f.name in wanted)]
    table = None
    if len(cols) >= 2 and len({rb.schema.field(c).type for c in cols}) == 1 and rb.num_rows > 0:
        table, _ = DeviceTable.from_dataframe(ctx, rb, cols, drop_null=False)
    _shared.entry = (rb, ctx, table, frozenset(cols))
    try:
        yield table
    finally:
        _shared.entry = prev


class CrossValidation:
    """pbn.CrossValidation(df, k=10, seed=None, include_null=False): iterating yields (train, test) tables; `.indices()`
    yields the (train_indices, test_indices) of generate_cv_pair_indices; `.fold(i)`, `.loc(columns)`."""

    def __init__(self, df, k=10, seed=None, include_null=False):
        import random

        from . import _lib

        self._rb = as_record_batch(df)
        self._k = int(k)
        self._seed = random.SystemRandom().randrange(0, 2 ** 32) if seed is None else int(seed)
        self._include_null = bool(include_null)
        self._rows = _valid_rows(self._rb, include_null)
        self._perm, self._limits, _, _ = _layout(len(self._rows), _lib.PBN_SPLIT_CV, self._k, self._seed, 0.0)

    def indices(self):
        src = self._rows[self._perm]
        for f in range(self._k):
            lo, hi = int(self._limits[f]), int(self._limits[f + 1])
            yield (__import__("numpy").concatenate([src[:lo], src[hi:]]), src[lo:hi].copy())

    def _take(self, idx):
        import pyarrow as pa

        return self._rb.take(pa.array(idx))

    def fold(self, i):
        if not 0 <= i < self._k:
            raise IndexError("fold index out of range")
        tr, te = list(self.indices())[i]
        return self._take(tr), self._take(te)

    def loc(self, columns):
        columns = [columns] if isinstance(columns, (str, int)) else list(columns)
        names = [c if isinstance(c, str) else self._rb.schema.names[c] for c in columns]
        out = CrossValidation.__new__(CrossValidation)
        out.__dict__.update(self.__dict__)
        out._rb = self._rb.select(names)
        return out

    def __iter__(self):
        for tr, te in self.indices():
            yield self._take(tr), self._take(te)


class HoldOut:
    """pbn.HoldOut(df, test_ratio=0.2, seed=None, include_null=False) with training_data() / test_data()."""

    def __init__(self, df, test_ratio=0.2, seed=None, include_null=False):
        import random

        from . import _lib

        self._rb = as_record_batch(df)
        self._seed = random.SystemRandom().randrange(0, 2 ** 32) if seed is None else int(seed)
        rows = _valid_rows(self._rb, include_null)
        perm, _, n_train, n_test = _layout(len(rows), _lib.PBN_SPLIT_HOLDOUT, 0, self._seed, test_ratio)
        src = rows[perm]
        self._train, self._test = src[:n_train], src[n_train: n_train + n_test]

    def training_data(self):
        import pyarrow as pa

        return self._rb.take(pa.array(self._train))

    def test_data(self):
        import pyarrow as pa

        return self._rb.take(pa.array(self._test))
