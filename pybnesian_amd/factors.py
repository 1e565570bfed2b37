"""Continuous factors of the hot path: CKDE and LinearGaussianCPD — the reference's Python surface
(/root/reference/pybnesian/pybindings/pybindings_factors.cpp:378-640) on top of the HIP C ABI.

CKDE (factors/continuous/CKDE.hpp:182-287): ratio of KDEs.  The reference runs two sweeps (joint, then
marginal) and subtracts; here one fused sweep yields both logsumexps (csrc/kde_kernels.hip, COND=true).
"""
import ctypes as C

import numpy as np
import pyarrow as pa

from . import _lib
from .dataset import DeviceTable, as_record_batch, default_context, same_type
from .kde import NormalReferenceRule, _LibrarySelector


class Factor:
    """The subset of factors::Factor (factors/factors.hpp:113-189) that the hot path exposes."""

    def __init__(self, variable, evidence):
        self._variable = variable
        self._evidence = list(evidence)
        self._fitted = False

    def variable(self):
        return self._variable

    def evidence(self):
        return list(self._evidence)

    def fitted(self):
        return self._fitted

    def _check_fitted(self, name):
        if not self._fitted:
            raise ValueError(f"{name} factor not fitted.")

    def save(self, filename):
        """Factor::save (factors/factors.hpp:150-152, util/pickle.hpp): pickle to `filename`.pickle."""
        import pickle

        with open(filename if filename.endswith(".pickle") else filename + ".pickle", "wb") as f:
            pickle.dump(self, f, protocol=2)

    # A Python-derived factor is pickled as the reference pickles it (pybindings_factors.cpp:160-233): variable and
    # evidence from the base class plus whatever __getstate_extra__() returns, handed back to __setstate_extra__().
    # Without those two methods the instance dictionary travels as usual.
    def __getstate__(self):
        extra = getattr(self, "__getstate_extra__", None)
        if extra is None:
            return dict(self.__dict__)
        return {"__factor_base__": (self._variable, list(self._evidence)), "__factor_extra__": extra()}

    def __setstate__(self, state):
        if "__factor_base__" in state:
            Factor.__init__(self, *state["__factor_base__"])
            self.__setstate_extra__(state["__factor_extra__"])
        else:
            self.__dict__.update(state)


def _random_seed():
    import random

    return random.SystemRandom().randrange(0, 2 ** 32)  # std::random_device{}()


def _check_sample_args(n, evidence, evidence_values):
    if n < 0:
        raise ValueError("n should be a non-negative number")
    if not evidence:
        return None
    if evidence_values is None:
        raise ValueError("Evidence values not present for sampling.")
    rb = as_record_batch(evidence_values)
    if any(rb.schema.get_field_index(e) < 0 for e in evidence):
        raise ValueError("Evidence values not present for sampling.")
    return rb


class CKDE(Factor):
    """Conditional KDE: logl = logl_joint([variable] + evidence) - logl_marg(evidence)."""

    def __init__(self, variable, evidence, bandwidth_selector=None):
        super().__init__(variable, evidence)
        self._variables = [variable] + list(evidence)
        self._selector = bandwidth_selector if bandwidth_selector is not None else NormalReferenceRule()
        self._handle = None
        self._train = None
        self._dtype = None
        self._bandwidth = None
        self._N = 0
        self._kde_joint_obj = None
        self._kde_marg_obj = None
        self._split = False

    def type(self):
        from .models import CKDEType

        return CKDEType()

    def data_type(self):
        self._check_fitted("CKDE")
        return pa.float64() if self._dtype == _lib.PBN_F64 else pa.float32()

    def num_instances(self):
        self._check_fitted("CKDE")
        return self._N

    @property
    def bandwidth(self):
        """Joint bandwidth matrix in [variable, evidence...] order (kde_joint().bandwidth)."""
        return self._bandwidth

    def _member_kde(self, variables, H):
        from .kde import KDE

        k = KDE(variables)
        k._train, k._dtype, k._N = self._train, self._dtype, self._N
        k._train_idx = self._train.index(variables)
        k._bandwidth = np.array(H)
        k._device_fit()
        k._fitted = True
        k._on_bandwidth = self._member_bandwidth_changed
        return k

    def _member_bandwidth_changed(self):
        """The reference's CKDE owns two independent KDE members (CKDE.hpp:262-263) and hands out references to them,
        so assigning `cpd.kde_joint().bandwidth` changes what the factor evaluates.  From then on logl / slogl are the
        difference of the two members' device sweeps (the fused sweep needs marg H = joint H[1:,1:]); cdf and sample
        keep reading the joint bandwidth as the reference does (CKDE.hpp:289-385, 509-558)."""
        self._split = True
        if self._kde_joint_obj is not None and not np.array_equal(self._kde_joint_obj.bandwidth, self._bandwidth):
            self._refit_handle(np.asfortranarray(self._kde_joint_obj.bandwidth, dtype=np.float64))

    def _refit_handle(self, H):
        lib = _lib.load()
        d = len(self._variables)
        h = C.c_void_p()
        _lib.check(lib.pbn_ckde_fit(self._train.ctx.handle, self._train.handle, _lib.int_array(self._train.index(self._variables)), d, 0,
                                    self._N, _lib.dptr(H), None, C.byref(h)))
        if self._handle is not None:
            lib.pbn_kde_destroy(self._handle)
        self._handle, self._bandwidth = h, H

    def kde_joint(self):
        """KDE over [variable] + evidence sharing this factor's training rows (`CKDE.kde_joint`); the same object on
        every call, as the reference returns a reference to its member."""
        self._check_fitted("CKDE")
        if self._kde_joint_obj is None:
            self._kde_joint_obj = self._member_kde(self._variables, self._bandwidth)
        return self._kde_joint_obj

    def kde_marg(self):
        """KDE over the evidence with the bandwidth block H[1:, 1:] (CKDE.hpp:186-199); same object on every call."""
        self._check_fitted("CKDE")
        if not self._evidence:
            raise ValueError("CKDE without evidence has no marginal KDE.")
        if self._kde_marg_obj is None:
            self._kde_marg_obj = self._member_kde(self._evidence, self._bandwidth[1:, 1:])
        return self._kde_marg_obj

    def fit(self, df):
        rb = as_record_batch(df)
        dtype = same_type(rb, self._variables)
        ctx = default_context()
        table, _ = DeviceTable.from_dataframe(ctx, rb, self._variables)
        self._fit_table(table, table.index(self._variables), rb)

    def fit_table(self, table):
        self._fit_table(table, table.index(self._variables), None)

    def _fit_table(self, table, idx, rb):
        d = len(self._variables)
        n = table.num_rows
        names = [table.names[i] for i in idx]
        sel = self._selector
        means = None
        if isinstance(sel, _LibrarySelector):
            if n <= d:
                cov = np.eye(d)
            else:
                means, sse = table.sse(names)
                cov = sse / (n - 1)
            H = sel._from_cov(_lib.PBN_BW_FULL, cov, n, table.dtype)
        else:
            if rb is None:
                raise ValueError("fit_table needs a library bandwidth selector")
            H = np.asarray(sel.bandwidth(rb, self._variables), dtype=np.float64)
        H = np.asfortranarray(H, dtype=np.float64)
        lib = _lib.load()
        if self._handle is not None:
            lib.pbn_kde_destroy(self._handle)
            self._handle = None
        h = C.c_void_p()
        cptr = _lib.dptr(np.ascontiguousarray(means)) if means is not None else None
        _lib.check(lib.pbn_ckde_fit(table.ctx.handle, table.handle, _lib.int_array(idx), d, 0, n, _lib.dptr(H), cptr, C.byref(h)))
        self._handle, self._train, self._dtype, self._bandwidth, self._N = h, table, table.dtype, H, n
        self._kde_joint_obj, self._kde_marg_obj, self._split = None, None, False
        self._fitted = True

    # pickle: CKDE::__getstate__ (factors/continuous/CKDE.hpp:737-745): (variable, evidence, fitted, joint KDE tuple) with the
    # joint tuple in KDE::__getstate__'s layout (kde/KDE.hpp:642-666) and () when unfitted; the marginal is rebuilt from the
    # bottom-right block of the joint bandwidth (CKDE.cpp:190-214).  A marginal bandwidth the user overrode through
    # kde_marg().bandwidth - which the reference's pickle silently loses - rides as an optional fifth entry.
    def __getstate__(self):
        if not self._fitted:
            return (self._variable, list(self._evidence), False, ())
        np_t = np.float64 if self._dtype == _lib.PBN_F64 else np.float32
        vals = np.asarray(self._train.read(self._variables), dtype=np_t)
        joint = (list(self._variables), True, self._selector, np.array(self._bandwidth), np.asfortranarray(vals).reshape(-1, order="F"),
                 float(_lib.load().pbn_kde_lognorm(self._handle, 0)), int(self._N), 12 if self._dtype == _lib.PBN_F64 else 11)
        state = (self._variable, list(self._evidence), True, joint)
        if self._split and self._evidence:
            state += (np.array(self.kde_marg().bandwidth),)
        return state

    def __setstate__(self, state):
        if isinstance(state, dict):      # states written before the reference layout was adopted
            joint = ()
            if state["fitted"]:
                joint = ([state["variable"]] + list(state["evidence"]), True, state["selector"], state["bandwidth"], state["training"], -1.0,
                         state["N"], 12 if state["dtype"] == _lib.PBN_F64 else 11)
            extra = (state["marg_bandwidth"],) if "marg_bandwidth" in state else ()
            legacy_selector = state.get("selector")   # an UNFITTED legacy state has no joint tuple to carry its selector
            state = (state["variable"], state["evidence"], state["fitted"], joint) + extra
        else:
            legacy_selector = None
        if len(state) not in (4, 5):
            raise RuntimeError("Not valid CKDE.")                   # CKDE.cpp:177
        variable, evidence, fitted, joint = state[:4]
        self.__init__(variable, evidence, joint[2] if fitted else legacy_selector)
        if fitted:
            if len(joint) != 8:
                raise RuntimeError("Not valid KDE.")
            d, n, type_id = len(self._variables), int(joint[6]), joint[7]
            np_t = np.float64 if type_id == 12 else np.float32
            vals = np.asarray(joint[4], dtype=np_t).reshape(n, d, order="F")
            rb = pa.RecordBatch.from_pydict({v: pa.array(np.ascontiguousarray(vals[:, i])) for i, v in enumerate(self._variables)})
            table, _ = DeviceTable.from_dataframe(default_context(), rb, self._variables, drop_null=False)
            H = np.asfortranarray(joint[3], dtype=np.float64)
            h = C.c_void_p()
            _lib.check(_lib.load().pbn_ckde_fit(table.ctx.handle, table.handle, _lib.int_array(range(d)), d, 0, n, _lib.dptr(H), None, C.byref(h)))
            self._handle, self._train, self._dtype, self._bandwidth, self._N = h, table, table.dtype, H, n
            self._fitted = True
            if len(state) == 5:
                self.kde_marg().bandwidth = state[4]

    def _upload_test(self, df):
        self._check_fitted("CKDE")
        rb = as_record_batch(df)
        dtype = same_type(rb, self._variables)
        if dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        table, mask = DeviceTable.from_dataframe(self._train.ctx, rb, self._variables)
        return rb, table, mask

    def logl(self, df):
        if self._split:
            self._upload_test(df)  # same argument checks as the fused path
            lj = self.kde_joint().logl(df)
            return lj - self.kde_marg().logl(df) if self._evidence else lj
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

    def cdf(self, df):
        """CKDE.cdf (factors/continuous/CKDE.hpp:126, 509-558): P(X <= x | evidence) per row, NaN at null rows."""
        rb, table, mask = self._upload_test(df)
        m = table.num_rows
        vals = np.empty(m, dtype=np.float64)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_ckde_cdf(self._handle, table.handle, _lib.int_array(table.index(self._variables)), 0, m, _lib.dptr(vals)))
        if mask is None:
            return vals
        out = np.full(rb.num_rows, np.nan)
        out[mask] = vals
        return out

    def sample(self, n, evidence_values=None, seed=None, _stream_n=0):
        """CKDE.sample(n, evidence_values, seed) (factors/continuous/CKDE.hpp:289-385): pyarrow array of the factor's
        data type.  Instance selection on the device, random numbers as the reference draws them (pbn_ckde_sample)."""
        self._check_fitted("CKDE")
        rb = _check_sample_args(n, self._evidence, evidence_values)
        seed = _random_seed() if seed is None else int(seed)
        npdt = np.float64 if self._dtype == _lib.PBN_F64 else np.float32
        out = np.empty(n, dtype=npdt)
        table, cols = None, None
        if rb is not None:
            if rb.num_rows < n:
                raise ValueError(f"Evidence values do not have {n} rows to sample.")
            if same_type(rb, self._evidence) != self._dtype:
                raise ValueError("Data type of training and test datasets is different.")
            table, mask = DeviceTable.from_dataframe(self._train.ctx, rb, self._evidence)
            if mask is not None:
                raise ValueError("Evidence values contain null rows in the evidence variables.")
            cols = _lib.int_array(table.index(self._evidence))
        _lib.check(_lib.load().pbn_ckde_sample(self._handle, n, int(_stream_n), table.handle if table is not None else None, cols,
                                               C.c_uint32(seed), out.ctypes.data_as(C.c_void_p)))
        return pa.array(out)

    def slogl(self, df):
        if self._split:
            return float(np.nansum(self.logl(df)))
        _, table, _ = self._upload_test(df)
        res = C.c_double(0.0)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_kde_slogl(self._handle, table.handle, _lib.int_array(table.index(self._variables)), 0, table.num_rows, C.byref(res)))
        return res.value

    def slogl_table(self, table, row0=0, n=None):
        self._check_fitted("CKDE")
        if table.dtype != self._dtype:
            raise ValueError("Data type of training and test datasets is different.")
        if self._split:
            lj = self.kde_joint().slogl_table(table, None, row0, n)
            return lj - self.kde_marg().slogl_table(table, None, row0, n) if self._evidence else lj
        idx = table.index(self._variables)
        n = table.num_rows - row0 if n is None else n
        res = C.c_double(0.0)
        _lib.check(_lib.load().pbn_kde_slogl(self._handle, table.handle, _lib.int_array(idx), row0, n, C.byref(res)))
        return res.value

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
        if self._evidence:
            return f"[CKDE] P({self._variable} | {', '.join(self._evidence)})"
        return f"[CKDE] P({self._variable})"


class LinearGaussianCPD(Factor):
    """factors/continuous/LinearGaussianCPD.{hpp,cpp}: y ~ N(beta0 + sum_i beta_i x_i, variance).
    fit = MLE (learning/parameters/mle_LinearGaussianCPD.hpp) from one device Gram pass; logl / slogl = one
    streaming device pass."""

    def __init__(self, variable, evidence, beta=None, variance=None):
        super().__init__(variable, evidence)
        self._variables = [variable] + list(evidence)
        if (beta is None) != (variance is None):
            raise ValueError("beta and variance must be given together.")
        if beta is not None:
            beta = np.asarray(beta, dtype=np.float64)
            if beta.size != len(evidence) + 1:
                raise ValueError("Wrong number of beta parameters. Beta vector length: %d, Evidence length: %d" % (beta.size, len(evidence)))
            if variance <= 0:
                raise ValueError("Variance must be a positive value.")
            self.beta, self.variance, self._fitted = beta, float(variance), True
        else:
            self.beta, self.variance = None, None

    def type(self):
        from .models import LinearGaussianCPDType

        return LinearGaussianCPDType()

    def fit(self, df):
        rb = as_record_batch(df)
        same_type(rb, self._variables)
        table, _ = DeviceTable.from_dataframe(default_context(), rb, self._variables)
        self.fit_table(table)

    def fit_table(self, table, row0=0, n=None):
        idx = table.index(self._variables)
        n = table.num_rows - row0 if n is None else n
        beta = np.zeros(len(idx))
        var = C.c_double(0.0)
        _lib.check(_lib.load().pbn_lg_fit_table(table.handle, _lib.int_array(idx), len(idx), row0, n, _lib.dptr(beta), C.byref(var)))
        self.beta, self.variance, self._fitted = beta, var.value, True

    def _eval(self, df, want_logl):
        self._check_fitted("LinearGaussianCPD")
        rb = as_record_batch(df)
        same_type(rb, self._variables)
        table, mask = DeviceTable.from_dataframe(default_context(), rb, self._variables)
        m = table.num_rows
        vals = np.empty(m) if want_logl else None
        s = C.c_double(0.0)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_lg_logl(table.handle, _lib.int_array(table.index(self._variables)), d, 0, m, _lib.dptr(np.ascontiguousarray(self.beta)),
                                           float(self.variance), _lib.dptr(vals) if want_logl else None, C.byref(s)))
        return rb, mask, vals, s.value

    def logl(self, df):
        rb, mask, vals, _ = self._eval(df, True)
        if mask is None:
            return vals
        out = np.full(rb.num_rows, np.nan)
        out[mask] = vals
        return out

    def slogl(self, df):
        return self._eval(df, False)[3]

    def data_type(self):
        return pa.float64()  # LinearGaussianCPD.hpp: always double

    def cdf(self, df):
        """LinearGaussianCPD.cdf (LinearGaussianCPD.cpp:171-249): Phi((y - beta.x) / sigma) per row, NaN at null rows."""
        self._check_fitted("LinearGaussianCPD")
        rb = as_record_batch(df)
        same_type(rb, self._variables)
        table, mask = DeviceTable.from_dataframe(default_context(), rb, self._variables)
        m = table.num_rows
        vals = np.empty(m)
        d = len(self._variables)
        _lib.check(_lib.load().pbn_lg_cdf(table.handle, _lib.int_array(table.index(self._variables)), d, 0, m, _lib.dptr(np.ascontiguousarray(self.beta)),
                                          float(self.variance), _lib.dptr(vals)))
        if mask is None:
            return vals
        out = np.full(rb.num_rows, np.nan)
        out[mask] = vals
        return out

    def sample(self, n, evidence_values=None, seed=None, _stream_n=0):
        """LinearGaussianCPD.sample (LinearGaussianCPD.cpp:317-380): float64 pyarrow array (pbn_lg_sample)."""
        self._check_fitted("LinearGaussianCPD")
        rb = _check_sample_args(n, self._evidence, evidence_values)
        seed = _random_seed() if seed is None else int(seed)
        out = np.empty(n)
        keep, ptrs, ev_dtype = [], None, _lib.PBN_F64
        if rb is not None:
            if rb.num_rows < n:
                raise ValueError(f"Evidence values do not have {n} rows to sample.")
            ev_dtype = same_type(rb, self._evidence)
            npdt = np.float64 if ev_dtype == _lib.PBN_F64 else np.float32
            for e in self._evidence:
                col = rb.column(rb.schema.get_field_index(e))
                keep.append(np.ascontiguousarray(col.to_numpy(zero_copy_only=False), dtype=npdt))
            ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data for k in keep])
        _lib.check(_lib.load().pbn_lg_sample(n, _lib.dptr(np.ascontiguousarray(self.beta, dtype=np.float64)), len(self._evidence),
                                             float(self.variance), C.c_uint32(seed), ptrs, ev_dtype, _lib.dptr(out)))
        return pa.array(out)

    def __str__(self):
        return f"[LinearGaussianCPD] P({self._variable} | {', '.join(self._evidence)})"


class LinearGaussianParams:
    """LinearGaussianCPD::ParamsClass (pybindings_parameters.cpp:43-62)."""

    def __init__(self, beta, variance):
        self.beta, self.variance = np.asarray(beta, dtype=np.float64), float(variance)


class DiscreteFactorParams:
    """DiscreteFactor::ParamsClass (pybindings_parameters.cpp:93-135): `logprob` with one axis per variable - the
    factor's variable first, then its evidence - and `cardinality`."""

    def __init__(self, logprob):
        self.logprob = np.asarray(logprob, dtype=np.float64)

    @property
    def cardinality(self):
        return np.asarray(self.logprob.shape, dtype=np.int32)


class MLELinearGaussianCPD:
    """MLE<LinearGaussianCPD> (learning/parameters/mle_LinearGaussianCPD.cpp:5-45): one device Gram pass + closed forms."""

    def estimate(self, df, variable, evidence):
        cpd = LinearGaussianCPD(variable, list(evidence))
        cpd.fit(df)
        return LinearGaussianParams(cpd.beta, cpd.variance)


class MLEDiscreteFactor:
    """MLE<DiscreteFactor> (learning/parameters/mle_DiscreteFactor.cpp:5-41)."""

    def estimate(self, df, variable, evidence):
        f = DiscreteFactor(variable, list(evidence))
        f.fit(df)
        return DiscreteFactorParams(f._logprob.reshape(f._cards, order="F"))


def MLE(factor_type):
    """pbn.MLE(factor_type) (pybindings_parameters.cpp:31-40, mle_base.cpp): the estimator of LinearGaussianCPDType or
    DiscreteFactorType; any other type raises "MLE not available"."""
    from .models import DiscreteFactorType, LinearGaussianCPDType

    if factor_type == LinearGaussianCPDType():
        return MLELinearGaussianCPD()
    if factor_type == DiscreteFactorType():
        return MLEDiscreteFactor()
    raise ValueError(f"MLE not available for NodeType {factor_type}.")


# ---- hybrid factors (factors/discrete/) --------------------------------------------------------------------------
def _dictionary_column(rb, name):
    import pyarrow as pa

    idx = rb.schema.get_field_index(name)
    if idx < 0:
        raise KeyError(f"Column {name} not present in DataFrame.")
    col = rb.column(idx)
    if not pa.types.is_dictionary(col.type):
        raise ValueError(f"Variable {name} is not categorical.")
    return col


class DiscreteFactor(Factor):
    """Multinomial CPT (factors/discrete/DiscreteFactor.cpp:34-171, learning/parameters/mle_DiscreteFactor.cpp:5-41).
    Integer counting on the host: logP = log(count) - log(sum over the parent
    configuration), uniform for unseen configurations."""

    def __init__(self, variable, evidence):
        super().__init__(variable, evidence)
        self._logprob = None

    def type(self):
        from .models import DiscreteFactorType

        return DiscreteFactorType()

    def _indices(self, rb):
        cols = [_dictionary_column(rb, v) for v in [self._variable] + self._evidence]
        if self._fitted:
            for c, cats, name in zip(cols, self._categories, [self._variable] + self._evidence):
                if c.dictionary.to_pylist() != cats:
                    raise ValueError(f"Variable {name} does not contain the same categories.")  # discrete_indices.cpp:206-227
        valid = np.ones(rb.num_rows, dtype=bool)
        idx = np.zeros(rb.num_rows, dtype=np.int64)
        stride = 1
        cards = []
        for c in cols:
            codes = c.indices.to_numpy(zero_copy_only=False)
            if c.null_count:
                m = np.asarray(c.is_valid().to_numpy(zero_copy_only=False), dtype=bool)
                valid &= m
                codes = np.where(m, codes, 0)
            idx += codes.astype(np.int64) * stride
            cards.append(len(c.dictionary))
            stride *= cards[-1]
        return cols, idx, valid, cards

    def fit(self, df):
        rb = as_record_batch(df)
        self._fitted = False
        cols, idx, valid, cards = self._indices(rb)
        self._categories = [c.dictionary.to_pylist() for c in cols]
        self._cards = cards
        counts = np.bincount(idx[valid], minlength=int(np.prod(cards))).astype(np.float64).reshape(-1, cards[0])
        sums = counts.sum(axis=1, keepdims=True)
        with np.errstate(divide="ignore", invalid="ignore"):
            lp = np.where(sums > 0, np.log(counts) - np.log(np.where(sums > 0, sums, 1.0)), np.log(1.0 / cards[0]))
        self._logprob = lp.reshape(-1)
        self._fitted = True

    def logl(self, df):
        self._check_fitted("DiscreteFactor")
        _, idx, valid, _ = self._indices(as_record_batch(df))
        out = np.full(idx.shape[0], np.nan)
        out[valid] = self._logprob[idx[valid]]
        return out

    def slogl(self, df):
        return float(np.nansum(self.logl(df)))

    def data_type(self):
        self._check_fitted("DiscreteFactor")
        return pa.dictionary(self._index_type(), pa.string())

    def _index_type(self):
        """Smallest signed index type that holds the variable's categories, as pyarrow / pandas encode them
        (DiscreteFactor_test.py:11-35: 128 categories -> int8, 129 -> int16)."""
        card = self._cards[0]
        return pa.int8() if card <= 128 else pa.int16() if card <= 32768 else pa.int32()

    def sample(self, n, evidence_values=None, seed=None, _stream_n=0):
        """DiscreteFactor.sample (DiscreteFactor.cpp:173-208, .hpp:144-205): dictionary array (pbn_discrete_sample)."""
        self._check_fitted("DiscreteFactor")
        rb = _check_sample_args(n, self._evidence, evidence_values)
        seed = _random_seed() if seed is None else int(seed)
        card = self._cards[0]
        offs = None
        if rb is not None:
            if rb.num_rows != n:
                raise ValueError(f"Evidence values do not have {n} rows to sample.")
            off = np.zeros(n, dtype=np.int64)
            stride = card
            for e, cats in zip(self._evidence, self._categories[1:]):
                c = _dictionary_column(rb, e)
                if c.null_count:
                    raise ValueError("Evidence values contain null rows in the evidence variables.")
                if c.dictionary.to_pylist() != cats:
                    raise ValueError(f"Variable {e} does not contain the same categories.")
                off += c.indices.to_numpy(zero_copy_only=False).astype(np.int64) * stride
                stride *= len(cats)
            offs = np.ascontiguousarray(off, dtype=np.int32)
        out = np.zeros(n, dtype=np.int32)
        lp = np.ascontiguousarray(self._logprob, dtype=np.float64)
        _lib.check(_lib.load().pbn_discrete_sample(n, _lib.dptr(lp), card, lp.size,
                                                   offs.ctypes.data_as(C.POINTER(C.c_int)) if offs is not None else None,
                                                   C.c_uint32(seed), out.ctypes.data_as(C.POINTER(C.c_int))))
        return pa.DictionaryArray.from_arrays(pa.array(out, type=pa.int32()).cast(self._index_type()), pa.array(self._categories[0]))


class Assignment:
    """factors/assignment.hpp:154-260: values assigned to a set of variables (category names or numbers), the key of
    `conditional_factor`.  `Assignment({"a": "a1", "b": 2.5})`."""

    def __init__(self, assignments):
        self._values = dict(assignments)
        for k, v in self._values.items():
            if not isinstance(v, (str, int, float)):
                raise TypeError(f"Assignment value for {k} must be a string or a number.")

    def value(self, variable):
        if variable not in self._values:
            raise IndexError(f"Variable {variable} not found in Assignment.")
        return self._values[variable]

    def has_variables(self, variables):
        variables = [variables] if isinstance(variables, str) else list(variables)
        return all(v in self._values for v in variables)

    def empty(self):
        return not self._values

    def size(self):
        return len(self._values)

    def insert(self, variable, value):
        self._values[variable] = value

    def remove(self, variable):
        self._values.pop(variable, None)

    def __iter__(self):
        return iter(self._values.items())

    def __eq__(self, other):
        return isinstance(other, Assignment) and self._values == other._values

    def __hash__(self):
        return hash(frozenset(self._values.items()))

    def __str__(self):
        return "[" + ", ".join(f"{k} = {v}" for k, v in self._values.items()) + "]"

    __repr__ = __str__


class _DiscreteAdaptator(Factor):
    """DiscreteAdaptator<Base, Fitter> (factors/discrete/DiscreteAdaptator.hpp:201-348): one base factor per
    configuration of the discrete evidence; the per-slice work runs on the device through the base factor."""

    _name = "DiscreteAdaptator"

    def __init__(self, variable, evidence):
        super().__init__(variable, evidence)
        self._factors = []

    def _base(self, continuous_evidence):
        raise NotImplementedError

    def _fit_base(self, factor, df):
        raise NotImplementedError

    def _split(self, rb):
        import pyarrow as pa

        disc, cont = [], []
        for e in self._evidence:
            t = rb.schema.field(e).type
            if pa.types.is_dictionary(t):
                disc.append(e)
            elif pa.types.is_floating(t):
                cont.append(e)
            else:
                raise ValueError(f"Non valid data type for variable {e}. Only \"dictionary\", \"double\" and \"float\" data types are allowed.")
        return disc, cont

    def _config(self, rb):
        idx = np.zeros(rb.num_rows, dtype=np.int64)
        valid = np.ones(rb.num_rows, dtype=bool)
        stride = 1
        for name, cats in zip(self._disc, self._categories):
            c = _dictionary_column(rb, name)
            if c.dictionary.to_pylist() != cats:
                raise ValueError(f"Variable {name} does not contain the same categories.")
            codes = c.indices.to_numpy(zero_copy_only=False)
            if c.null_count:
                m = np.asarray(c.is_valid().to_numpy(zero_copy_only=False), dtype=bool)
                valid &= m
                codes = np.where(m, codes, 0)
            idx += codes.astype(np.int64) * stride
            stride *= len(cats)
        return idx, valid

    def fit(self, df):
        import pyarrow as pa

        rb = as_record_batch(df)
        self._disc, self._cont = self._split(rb)
        self._categories = [_dictionary_column(rb, d).dictionary.to_pylist() for d in self._disc]
        self._factors = []
        if not self._disc:
            f = self._base(self._cont)
            f.fit(rb)
            self._factors = [f]
        else:
            ncfg = int(np.prod([len(c) for c in self._categories]))
            idx, valid = self._config(rb)
            for c in range(ncfg):
                rows = np.nonzero(valid & (idx == c))[0]
                if rows.size == 0:
                    self._factors.append(None)
                    continue
                f = self._base(self._cont)
                ok = self._fit_base(f, rb.take(pa.array(rows.astype(np.int32))))
                self._factors.append(f if ok else None)
        self._fitted = True

    def conditional_factor(self, assignment):
        """DiscreteAdaptator::conditional_factor (DiscreteAdaptator.hpp:350-356): the base factor fitted on the rows of one
        configuration of the discrete evidence (None when the configuration was not observed)."""
        self._check_fitted(self._name)
        if not self._disc:
            return self._factors[0]
        index, stride = 0, 1
        for name, cats in zip(self._disc, self._categories):
            if not assignment.has_variables(name):
                raise ValueError(f"Discrete variable {name} not found in Assignment.")
            value = assignment.value(name)
            if value not in cats:
                raise ValueError(f"Category {value} not found for variable {name}.")
            index += cats.index(value) * stride
            stride *= len(cats)
        return self._factors[index]

    def logl(self, df):
        import pyarrow as pa

        self._check_fitted(self._name)
        rb = as_record_batch(df)
        if not self._disc:
            return self._factors[0].logl(rb)
        idx, valid = self._config(rb)
        out = np.full(rb.num_rows, np.nan)
        for c, f in enumerate(self._factors):
            rows = np.nonzero(valid & (idx == c))[0]
            if rows.size and f is not None:
                out[rows] = f.logl(rb.take(pa.array(rows.astype(np.int32))))
        return out

    def slogl(self, df):
        import pyarrow as pa

        self._check_fitted(self._name)
        rb = as_record_batch(df)
        if not self._disc:
            return self._factors[0].slogl(rb)
        idx, valid = self._config(rb)
        total = 0.0
        for c, f in enumerate(self._factors):
            rows = np.nonzero(valid & (idx == c))[0]
            if rows.size and f is not None:
                total += f.slogl(rb.take(pa.array(rows.astype(np.int32))))
        return total


    def sample(self, n, evidence_values=None, seed=None, _stream_n=0):
        """DiscreteAdaptator::sample (DiscreteAdaptator.hpp:427-520): slice i is sampled by its factor with seed + i;
        every slice factor is asked for the whole batch size, so its random stream is the one of an n-sample call."""
        self._check_fitted(self._name)
        rb = _check_sample_args(n, self._evidence, evidence_values)
        seed = _random_seed() if seed is None else int(seed)
        if not self._disc:
            return self._factors[0].sample(n, rb, seed)
        if rb.num_rows != n:
            raise ValueError(f"Evidence values do not have {n} rows to sample.")
        idx, valid = self._config(rb)
        if not valid.all():
            raise ValueError("Evidence values contain null rows in the evidence variables.")
        first = next((f for f in self._factors if f is not None), None)
        npdt = np.float32 if (first is not None and first.data_type() == pa.float32()) else np.float64
        out = np.full(n, np.nan, dtype=npdt)
        for c, f in enumerate(self._factors):
            rows = np.nonzero(idx == c)[0]
            if rows.size == 0 or f is None:
                continue
            sub = rb.take(pa.array(rows.astype(np.int32)))
            out[rows] = f.sample(int(rows.size), sub, (seed + c) & 0xFFFFFFFF, _stream_n=n).to_numpy(zero_copy_only=False)
        return pa.array(out)


class CLinearGaussianCPD(_DiscreteAdaptator):
    """CLinearGaussianCPD = DiscreteAdaptator<LinearGaussianCPD, LinearGaussianFitter> (LinearGaussianCPD.hpp:120-140)."""

    _name = "CLinearGaussianCPD"

    def type(self):
        from .models import LinearGaussianCPDType

        return LinearGaussianCPDType()

    def _base(self, continuous_evidence):
        return LinearGaussianCPD(self._variable, continuous_evidence)

    def _fit_base(self, factor, df):
        factor.fit(df)
        return not (factor.variance < 1.4901161193847656e-08 or np.isinf(factor.variance))


class HCKDE(_DiscreteAdaptator):
    """HCKDE = DiscreteAdaptator<CKDE, CKDEFitter> (CKDE.hpp:745-770): SingularCovarianceData drops the slice's factor."""

    _name = "HCKDE"

    def type(self):
        from .models import CKDEType

        return CKDEType()

    def _base(self, continuous_evidence):
        return CKDE(self._variable, continuous_evidence)

    def _fit_base(self, factor, df):
        try:
            factor.fit(df)
            return True
        except _lib.SingularCovarianceData:
            return False
