"""Conditional-independence tests and the MMPC restriction phase.

`IndependenceTest` is the reference's abstract test (learning/independences/independence.hpp:17-77), subclassable
from Python; `LinearCorrelation` (learning/independences/continuous/linearcorrelation.hpp) keeps the covariance of all
continuous columns, taken once on the device, and answers every p-value from it on the host.
"""
import ctypes as C

import numpy as np

from . import _lib
from .dataset import DeviceTable, as_record_batch, default_context


class IndependenceTest:
    """Subclass and implement pvalue(x, y, z=None), num_variables(), variable_names(), name(i), has_variables(v)."""

    def pvalue(self, x, y, z=None):
        raise NotImplementedError("Tried to call pure virtual function \"IndependenceTest::pvalue\"")

    def variable_names(self):
        raise NotImplementedError("Tried to call pure virtual function \"IndependenceTest::variable_names\"")

    def num_variables(self):
        return len(self.variable_names())

    def name(self, index):
        return self.variable_names()[index]

    def has_variables(self, variables):
        names = set(self.variable_names())
        variables = [variables] if isinstance(variables, str) else variables
        return all(v in names for v in variables)

    # -- engine hook: (callback, user pointer, keep-alive) for pbn_mmpc_cpcs over the node order `nodes` ------------------
    def _ci_callback(self, nodes):
        errors = []

        def fn(_user, v1, v2, n_cond, cond):
            try:
                z = [nodes[cond[i]] for i in range(n_cond)]
                if not z:
                    return float(self.pvalue(nodes[v1], nodes[v2]))
                return float(self.pvalue(nodes[v1], nodes[v2], z[0] if len(z) == 1 else z))
            except Exception as ex:  # surfaced by the caller after the C call returns
                errors.append(ex)
                return float("nan")

        cb = _lib.CI_PVALUE_FN(fn)
        return cb, None, (cb, errors), errors


class LinearCorrelation(IndependenceTest):
    """pbn.LinearCorrelation(df): partial-correlation t-test on the continuous columns."""

    def __init__(self, df, ctx=None):
        import pyarrow as pa

        rb = as_record_batch(df)
        cont = [f.name for f in rb.schema if pa.types.is_floating(f.type)]
        if len(cont) < 2:
            raise ValueError("DataFrame does not contain enough continuous columns.")
        self._all_names = [f.name for f in rb.schema]
        self._names = cont
        self._index = {n: i for i, n in enumerate(cont)}
        ctx = ctx or default_context()
        self._handle = None
        self._per_test = None
        if any(rb.column(rb.schema.get_field_index(c)).null_count for c in cont):
            # nulls: no cached covariance - every test takes the covariance of ITS variables over the rows valid in all
            # of them (continuous/linearcorrelation.cpp:20-122, the pvalue_impl branch), one device pass per test
            self._per_test = MutualInformation(rb.select(cont), True, ctx)
            return
        table, _ = DeviceTable.from_dataframe(ctx, rb, cont, drop_null=False)
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_lincor_create(ctx.handle, table.handle, C.byref(h)))
        self._handle = h

    @classmethod
    def from_covariance(cls, names, cov, num_rows):
        """A test over a covariance matrix already at hand (host only, no device work)."""
        self = cls.__new__(cls)
        self._per_test = None
        self._all_names = self._names = list(names)
        self._index = {n: i for i, n in enumerate(self._names)}
        cov = np.asfortranarray(cov, dtype=np.float64)
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_lincor_from_cov(len(self._names), int(num_rows), _lib.dptr(cov), C.byref(h)))
        self._handle = h
        return self

    def _idx(self, name):
        if name not in self._index:
            raise ValueError(f"Continuous variable {name} not present in LinearCorrelation.")
        return self._index[name]

    def pvalue(self, x, y, z=None):
        cond = [] if z is None else ([z] if isinstance(z, str) else list(z))
        arr = _lib.int_array([self._idx(c) for c in cond] or [0])
        lib = _lib.load()
        if self._per_test is not None:
            _lib.check(lib.pbn_mi_set_order(self._per_test._handle, 0, None))
            p = lib.pbn_mi_lincor_pvalue(self._per_test._handle, self._idx(x), self._idx(y), len(cond), arr)
            if np.isnan(p):
                raise ValueError("LinearCorrelation: " + lib.pbn_last_error().decode())
            return p
        p = lib.pbn_lincor_pvalue(self._handle, self._idx(x), self._idx(y), len(cond), arr)
        if np.isnan(p):
            raise ValueError("LinearCorrelation: bad variable index")
        return p

    def covariance(self):
        if self._per_test is not None:
            raise ValueError("LinearCorrelation over a table with nulls keeps no covariance: every test uses its own valid rows.")
        n = len(self._names)
        cov = np.zeros((n, n), order="F")
        _lib.check(_lib.load().pbn_lincor_cov(self._handle, _lib.dptr(cov)))
        return cov

    def variable_names(self):
        return list(self._all_names)

    def _ci_callback(self, nodes):
        if self._per_test is not None:
            lib = _lib.load()
            _lib.check(lib.pbn_mi_set_order(self._per_test._handle, len(nodes), _lib.int_array([self._idx(n) for n in nodes])))
            return C.cast(lib.pbn_mi_lincor_pvalue, C.c_void_p), self._per_test._handle, self, []
        # native fast path: the C function itself is the callback, no Python frame per test
        if all(n in self._index for n in nodes) and [self._index[n] for n in nodes] == list(range(len(nodes))) and len(nodes) == len(self._names):
            lib = _lib.load()
            return C.cast(lib.pbn_lincor_pvalue, C.c_void_p), self._handle, self, []
        return super()._ci_callback(nodes)

    def __del__(self):
        try:
            if _lib.alive() and getattr(self, "_handle", None):
                _lib.load().pbn_lincor_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class MutualInformation(IndependenceTest):
    """pbn.MutualInformation(df, asymptotic_df=True): conditional-Gaussian mutual information for tables that mix
    float and dictionary columns (learning/independences/hybrid/mutual_information.hpp)."""

    def __init__(self, df, asymptotic_df=True, ctx=None):
        import pyarrow as pa

        rb = as_record_batch(df)
        self._all_names = [f.name for f in rb.schema]
        cont, disc = [], []
        for f in rb.schema:
            if pa.types.is_floating(f.type):
                cont.append(f.name)
            elif pa.types.is_dictionary(f.type):
                disc.append(f.name)
            else:
                raise ValueError(f"Wrong data type ({f.type}) for column {f.name}.")
        ctx = ctx or default_context()
        # nulls stay in the tables (NaN cells, category code -1): every test runs over the rows valid in all of ITS
        # variables, like the reference's contains_null overloads (hybrid/mutual_information.cpp:152-215)
        self._table = DeviceTable.from_dataframe(ctx, rb, cont, drop_null=False)[0] if cont else None
        self._codes, card = [], []
        for d in disc:
            col = rb.column(rb.schema.get_field_index(d))
            if isinstance(col, pa.ChunkedArray):
                col = col.combine_chunks()
            idx = col.indices
            if col.null_count:
                import pyarrow.compute as pc

                idx = pc.if_else(col.is_valid(), idx, pa.scalar(-1, type=idx.type))
            self._codes.append(np.ascontiguousarray(idx.to_numpy(zero_copy_only=False), dtype=np.int32))
            card.append(len(col.dictionary))
        self._id = {n: i for i, n in enumerate(cont + disc)}
        ptrs = (C.c_void_p * max(1, len(disc)))(*[c.ctypes.data for c in self._codes])
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_mi_create(ctx.handle, self._table.handle if self._table is not None else None, rb.num_rows,
                                             len(disc), ptrs, _lib.int_array(card or [0]), int(bool(asymptotic_df)), C.byref(h)))
        self._handle = h
        flags = np.array([1 if rb.column(rb.schema.get_field_index(c)).null_count else 0 for c in cont], dtype=np.uint8)
        if flags.any():
            shift = np.zeros(len(cont))
            for i, c in enumerate(cont):
                if flags[i]:
                    vals = rb.column(rb.schema.get_field_index(c)).to_numpy(zero_copy_only=False).astype(np.float64)
                    shift[i] = np.nanmean(vals) if np.isfinite(vals).any() else 0.0
            _lib.check(_lib.load().pbn_mi_set_continuous_nulls(h, flags.ctypes.data_as(C.POINTER(C.c_ubyte)), _lib.dptr(shift)))

    def _var(self, name):
        if name not in self._id:
            raise ValueError(f"Variable {name} not present in MutualInformation.")
        return self._id[name]

    def _args(self, x, y, z):
        cond = [] if z is None else ([z] if isinstance(z, str) else list(z))
        return self._var(x), self._var(y), len(cond), _lib.int_array([self._var(c) for c in cond] or [0])

    def mi(self, x, y, z=None):
        v = C.c_double(0.0)
        a = self._args(x, y, z)
        _lib.check(_lib.load().pbn_mi_value(self._handle, a[0], a[1], a[2], a[3], C.byref(v), None))
        return v.value

    def degrees_of_freedom(self, x, y, z=None):
        v = C.c_double(0.0)
        a = self._args(x, y, z)
        _lib.check(_lib.load().pbn_mi_value(self._handle, a[0], a[1], a[2], a[3], None, C.byref(v)))
        return v.value

    def pvalue(self, x, y, z=None):
        lib = _lib.load()
        _lib.check(lib.pbn_mi_set_order(self._handle, 0, None))
        a = self._args(x, y, z)
        p = lib.pbn_mi_pvalue(self._handle, a[0], a[1], a[2], a[3])
        if np.isnan(p):
            raise ValueError("MutualInformation: the test is not defined for these variables (" + lib.pbn_last_error().decode() + ")")
        return p

    def passes(self):
        d, h = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().pbn_mi_stats(self._handle, C.byref(d), C.byref(h)))
        return d.value, h.value

    def variable_names(self):
        return list(self._all_names)

    def _ci_callback(self, nodes):
        lib = _lib.load()
        _lib.check(lib.pbn_mi_set_order(self._handle, len(nodes), _lib.int_array([self._var(n) for n in nodes])))
        return C.cast(lib.pbn_mi_pvalue, C.c_void_p), self._handle, self, []

    def _ci_batch_callback(self):
        """Batched native callback (same handle / index order as _ci_callback): independent tests share a launch."""
        return C.cast(_lib.load().pbn_mi_pvalue_batch, C.c_void_p)

    def __del__(self):
        try:
            if _lib.alive() and getattr(self, "_handle", None):
                _lib.load().pbn_mi_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class KMutualInformation(IndependenceTest):
    """pbn.KMutualInformation(df, k, seed=None, shuffle_neighbors=5, samples=1000): k-nearest-neighbour (conditional)
    mutual information on the rank-transformed continuous columns with permutation p-values
    (learning/independences/continuous/mutual_information.{hpp,cpp}).  Neighbour distances and ball counts are device
    kernels over all pairs of rows; the permutations are drawn as the reference draws them."""

    def __init__(self, df, k, seed=None, shuffle_neighbors=5, samples=1000, ctx=None):
        import pyarrow as pa

        from .factors import _random_seed

        rb = as_record_batch(df)
        types = {str(f.type) for f in rb.schema}
        if not all(pa.types.is_floating(f.type) for f in rb.schema) or len(types) != 1:
            raise ValueError("Wrong data type in KMutualInformation.")
        if any(rb.column(i).null_count for i in range(rb.num_columns)):
            raise ValueError("KMutualInformation needs columns without nulls.")
        self._names = [f.name for f in rb.schema]
        self._index = {n: i for i, n in enumerate(self._names)}
        self._seed = _random_seed() if seed is None else int(seed)
        ctx = ctx or default_context()
        self._ctx = ctx
        cols = [np.ascontiguousarray(rb.column(i).to_numpy(zero_copy_only=False), dtype=np.float64) for i in range(rb.num_columns)]
        ptrs = (C.c_void_p * len(cols))(*[c.ctypes.data for c in cols])
        h = C.c_void_p()
        _lib.check(_lib.load().pbn_kmi_create(ctx.handle, ptrs, len(cols), rb.num_rows, int(k), C.c_uint32(self._seed & 0xFFFFFFFF),
                                              int(shuffle_neighbors), int(samples), C.byref(h)))
        self._handle = h

    def _idx(self, name):
        if name not in self._index:
            raise ValueError(f"Variable {name} not present in KMutualInformation.")
        return self._index[name]

    def _args(self, x, y, z):
        cond = [] if z is None else ([z] if isinstance(z, str) else list(z))
        return self._idx(x), self._idx(y), len(cond), _lib.int_array([self._idx(c) for c in cond] or [0])

    def mi(self, x, y, z=None):
        v = C.c_double(0.0)
        a = self._args(x, y, z)
        _lib.check(_lib.load().pbn_kmi_value(self._handle, a[0], a[1], a[2], a[3], C.byref(v)))
        return v.value

    def pvalue(self, x, y, z=None):
        lib = _lib.load()
        a = self._args(x, y, z)
        p = lib.pbn_kmi_pvalue(self._handle, a[0], a[1], a[2], a[3])
        if np.isnan(p):
            raise ValueError("KMutualInformation: " + lib.pbn_last_error().decode())
        return p

    def variable_names(self):
        return list(self._names)

    def __del__(self):
        try:
            if _lib.alive() and getattr(self, "_handle", None):
                _lib.load().pbn_kmi_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class ChiSquare(MutualInformation):
    """pbn.ChiSquare(df): Pearson's chi-square test on the categorical columns (learning/independences/discrete/
    chi_square.hpp); shares the device counting pass of MutualInformation."""

    def __init__(self, df, ctx=None):
        import pyarrow as pa

        rb = as_record_batch(df)
        disc = [f.name for f in rb.schema if pa.types.is_dictionary(f.type)]
        if len(disc) < 2:
            raise ValueError("DataFrame does not contain enough categorical columns.")
        super().__init__(rb.select(disc), True, ctx)
        self._all_names = [f.name for f in rb.schema]

    def pvalue(self, x, y, z=None):
        lib = _lib.load()
        _lib.check(lib.pbn_mi_set_order(self._handle, 0, None))
        a = self._args(x, y, z)
        p = lib.pbn_chisq_pvalue(self._handle, a[0], a[1], a[2], a[3])
        if np.isnan(p):
            raise ValueError("ChiSquare: " + lib.pbn_last_error().decode())
        return p

    def mi(self, x, y, z=None):
        raise AttributeError("ChiSquare has no mi()")

    def _ci_batch_callback(self):
        return None

    def _ci_callback(self, nodes):
        lib = _lib.load()
        _lib.check(lib.pbn_mi_set_order(self._handle, len(nodes), _lib.int_array([self._var(n) for n in nodes])))
        return C.cast(lib.pbn_chisq_pvalue, C.c_void_p), self._handle, self, []


def validate_restrictions(nodes, arc_blacklist=(), arc_whitelist=(), edge_blacklist=(), edge_whitelist=()):
    """util::validate_restrictions (util/validate_whitelists.hpp:72-146) over node indices: returns
    (arc_blacklist, arc_whitelist, edge_blacklist, edge_whitelist) as ordered lists of index pairs."""
    idx = {n: i for i, n in enumerate(nodes)}
    for lst in (arc_blacklist, arc_whitelist, edge_blacklist, edge_whitelist):
        for a, b in lst:
            for v in (a, b):
                if v not in idx:
                    raise ValueError(f"Node {v} not present in the graph.")
    und = lambda a, b: (min(a, b), max(a, b))
    e_bl = {und(idx[a], idx[b]): None for a, b in edge_blacklist}
    e_wl = {}
    for a, b in edge_whitelist:
        if und(idx[a], idx[b]) in e_bl:
            raise ValueError(f"Edge {a} -- {b} in blacklist and whitelist")
        e_wl[und(idx[a], idx[b])] = None
    a_wl = {}
    for a, b in arc_whitelist:
        s, t = idx[a], idx[b]
        if und(s, t) in e_bl:
            raise ValueError(f"Edge blacklist {a} -- {b} is incompatible with arc whitelist{a} -> {b}")
        e_wl.pop(und(s, t), None)
        a_wl[(s, t)] = None
    a_bl = {}
    for a, b in arc_blacklist:
        s, t = idx[a], idx[b]
        if (s, t) in a_wl:
            raise ValueError(f"Arc {a} -> {b} in blacklist and whitelist")
        if und(s, t) in e_wl:
            a_wl[(t, s)] = None
            del e_wl[und(s, t)]
        if und(s, t) not in e_bl:
            a_bl[(s, t)] = None
    for (s, t) in list(a_bl):
        if (s, t) in a_bl and (t, s) in a_bl:
            e_bl[und(s, t)] = None
            del a_bl[(s, t)]
            del a_bl[(t, s)]
    return list(a_bl), list(a_wl), list(e_bl), list(e_wl)


def mmpc_cpcs(test, nodes, alpha=0.05, arc_whitelist=(), edge_blacklist=(), edge_whitelist=(), symmetric=True, interface_nodes=()):
    """mmpc_all_variables (+ remove_asymmetries when `symmetric`): list of candidate parents-and-children (node names)
    per node, and the number of independence tests evaluated.  Index-pair restriction lists as validate_restrictions
    returns them.  With `interface_nodes` (conditional graph) the result covers nodes + interface_nodes, in that order."""
    nodes = list(nodes) + list(interface_nodes)
    n = len(nodes)
    fn, user, keep, errors = test._ci_callback(list(nodes))
    flat = lambda prs: _lib.int_array([v for p in prs for v in p] or [0])
    off = (C.c_int * (n + 1))()
    out = (C.c_int * max(1, n * (n - 1)))()
    ntests = C.c_int64(0)
    batch = getattr(test, "_ci_batch_callback", lambda: None)()
    from .distributed import sharded_ci_batch

    batch, keep_batch = sharded_ci_batch(fn, batch, user, errors)
    rc = _lib.load().pbn_mmpc_cpcs_batched(n, len(interface_nodes), fn, batch, user, float(alpha), len(arc_whitelist),
                                           flat(arc_whitelist), len(edge_blacklist), flat(edge_blacklist), len(edge_whitelist),
                                           flat(edge_whitelist), int(bool(symmetric)), off, out, C.byref(ntests))
    if errors:
        raise errors[0]
    _lib.check(rc)
    del keep, keep_batch
    return [[nodes[out[j]] for j in range(off[i], off[i + 1])] for i in range(n)], ntests.value
