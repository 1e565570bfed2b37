"""Scores of the hot path — BIC, BGe, CVLikelihood, HoldoutLikelihood, ValidatedLikelihood — with the
reference's Python surface (/root/reference/pybnesian/pybindings/pybindings_learning/
pybindings_scores.cpp:20-180,460-660) on top of the batched HIP score engine (pbn_score_batch).

The table is uploaded once; splits are generated on the library side with libstdc++'s std::shuffle so that
fold membership equals the reference's CrossValidation / HoldOut (dataset/crossvalidation_adaptator.hpp,
holdout_adaptator.hpp).  Tables with nulls are not accepted by the device engine yet.
"""
import ctypes as C
import random

import numpy as np

from . import _lib
from .dataset import DeviceTable, as_record_batch, default_context
from .models import (CKDEType, CLGNetworkType, DiscreteFactorType, GaussianNetworkType, KDENetworkType, LinearGaussianCPDType,
                     SemiparametricBNType, UnknownFactorType)

_TYPE_CODE = {LinearGaussianCPDType(): _lib.PBN_NODE_LG, CKDEType(): _lib.PBN_NODE_CKDE, DiscreteFactorType(): _lib.PBN_NODE_DISCRETE}


def _random_seed():
    return random.SystemRandom().randrange(0, 2 ** 32)  # std::random_device{}()


class Score:
    """Abstract score, subclassable from Python exactly as the reference's trampoline allows
    (pybindings_scores.cpp:281-392, docs/source/extending.rst): a subclass implements
    `local_score(model, variable, evidence)`, `has_variables(vars)`, `compatible_bn(model)` and optionally
    `local_score_node_type(model, variable_type, variable, evidence)`, `data()`, `score(model)`.  The hill-climbing
    engine calls such a score candidate by candidate through its batch callback (no device work involved)."""

    def local_score(self, model, variable, evidence=None):
        raise NotImplementedError("Tried to call pure virtual function \"Score::local_score\"")

    def local_score_node_type(self, model, variable_type, variable, evidence):
        raise NotImplementedError("Tried to call pure virtual function \"Score::local_score_node_type\"")

    def has_variables(self, variables):
        raise NotImplementedError("Tried to call pure virtual function \"Score::has_variables\"")

    def compatible_bn(self, model):
        raise NotImplementedError("Tried to call pure virtual function \"Score::compatible_bn\"")

    def score(self, model):
        """Score::score (scores.hpp:17-24): sum of the local scores."""
        return float(sum(self.local_score(model, n, model.parents(n)) for n in model.nodes()))

    def data(self):
        return None


class ValidatedScore(Score):
    """Score with a validation counterpart (scores.hpp:103-141, trampoline pybindings_scores.cpp:394-440)."""

    validated = True

    def vlocal_score(self, model, variable, evidence=None):
        raise NotImplementedError("Tried to call pure virtual function \"ValidatedScore::vlocal_score\"")

    def vlocal_score_node_type(self, model, variable_type, variable, evidence):
        raise NotImplementedError("Tried to call pure virtual function \"ValidatedScore::vlocal_score_node_type\"")

    def vscore(self, model):
        return float(sum(self.vlocal_score(model, n, model.parents(n)) for n in model.nodes()))


class Args:
    """factors/arguments.hpp:16-24: wrapper that marks a tuple as *args."""

    def __init__(self, *args):
        self.args = tuple(args)


class Kwargs:
    """factors/arguments.hpp:26-34: wrapper that marks a dict as **kwargs."""

    def __init__(self, **kwargs):
        self.kwargs = dict(kwargs)


class Arguments:
    """factors/arguments.hpp:36-140: construction arguments per (name, factor type) / name / factor type, most
    specific first."""

    def __init__(self, dict_arguments=None):
        from .models import FactorType

        self._name_type, self._name, self._type = {}, {}, {}
        for key, value in (dict_arguments or {}).items():
            av = self._process(value)
            if isinstance(key, str):
                self._name[key] = av
            elif isinstance(key, FactorType):
                self._type[key] = av
            elif isinstance(key, tuple) and len(key) == 2 and isinstance(key[0], str) and isinstance(key[1], FactorType):
                self._name_type[key] = av
            else:
                raise ValueError("Key value is not of type str, FactorType or 2-tuple (str, FactorType).")

    @staticmethod
    def _process(value):
        if isinstance(value, Args):
            return value.args, {}
        if isinstance(value, Kwargs):
            return (), value.kwargs
        if isinstance(value, dict):
            return (), dict(value)
        if isinstance(value, tuple):
            if len(value) == 2 and isinstance(value[0], Args) and isinstance(value[1], Kwargs):
                return value[0].args, value[1].kwargs
            return tuple(value), {}
        raise ValueError("The provided arguments must be a 2-tuple (Args, Kwargs), an Args/tuple or a Kwargs/dict.")

    def args(self, node, node_type):
        for table, key in ((self._name_type, (node, node_type)), (self._name, node), (self._type, node_type)):
            if key in table:
                return table[key]
        return (), {}

    def empty(self):
        return not (self._name_type or self._name or self._type)

    def __repr__(self):
        return "Arguments"


def _engine_selector(construction_args):
    """What the device engine can honour of an `Arguments`: one bandwidth selector for every CKDE it fits while
    scoring (NormalReferenceRule, the default, or ScottsBandwidth), given for CKDEType()."""
    from .kde import NormalReferenceRule, ScottsBandwidth

    if construction_args is None or construction_args.empty():
        return _lib.PBN_SEL_NORMAL_REFERENCE
    if construction_args._name or construction_args._name_type or list(construction_args._type) != [CKDEType()]:
        raise ValueError("The device score engine only accepts construction arguments keyed by CKDEType().")
    args, kwargs = construction_args._type[CKDEType()]
    sel = args[0] if len(args) == 1 and not kwargs else kwargs.get("bandwidth_selector") if (not args and list(kwargs) == ["bandwidth_selector"]) else None
    if type(sel) is NormalReferenceRule:
        return _lib.PBN_SEL_NORMAL_REFERENCE
    if type(sel) is ScottsBandwidth:
        return _lib.PBN_SEL_SCOTT
    raise ValueError("The device score engine only accepts NormalReferenceRule() or ScottsBandwidth() as the CKDE bandwidth selector.")


class _DeviceScore(Score):
    """Score evaluated by the batched HIP engine (pbn_score_batch)."""

    _kind = None
    _split = _lib.PBN_SPLIT_NONE
    _allowed_types = (LinearGaussianCPDType(), DiscreteFactorType())

    def __init__(self, df, split_args=(0, 0, 0.0), ctx=None, table=None):
        self._ctx = ctx or (table.ctx if table is not None else default_context())
        self._disc_names, self._disc_codes, self._disc_card, self._categories = [], [], [], {}
        if table is None:
            import pyarrow as pa

            rb = as_record_batch(df)
            self._masks = None
            if any(rb.column(i).null_count for i in range(rb.num_columns)):
                rb = self._handle_nulls(rb)
            cont = []
            for f in rb.schema:
                if pa.types.is_dictionary(f.type):
                    arr = rb.column(rb.schema.get_field_index(f.name))
                    self._disc_names.append(f.name)
                    self._disc_codes.append(np.ascontiguousarray(arr.indices.to_numpy(zero_copy_only=False), dtype=np.int32))
                    self._disc_card.append(len(arr.dictionary))
                    self._categories[f.name] = arr.dictionary.to_pylist()
                else:
                    cont.append(f.name)
            if not cont:
                raise ValueError("The device score engine needs at least one continuous column.")
            table, _ = DeviceTable.from_dataframe(self._ctx, rb, cont, drop_null=False)
            self._df = rb
            self._cont_names = cont
        else:
            self._df = None
        self._table = table
        self._names = list(table.names) + self._disc_names
        self._col = {n: i for i, n in enumerate(self._names)}
        k, seed, ratio = split_args
        h = C.c_void_p()
        from .distributed import comm

        cm = comm()
        self._comm = cm
        if cm is None:
            _lib.check(_lib.load().pbn_scoredata_create(self._ctx.handle, table.handle, self._split, int(k), C.c_uint32(int(seed)),
                                                        float(ratio), C.byref(h)))
        else:
            # one process per GPU: each rank takes the Gram of its share of the rows, one exchange of the moments; from then on
            # pbn_score_batch on this handle evaluates this rank's share of every batch and completes it through the job's all-gather
            _lib.check(_lib.load().pbn_scoredata_create_sharded(self._ctx.handle, table.handle, self._split, int(k),
                                                                C.c_uint32(int(seed)), float(ratio), cm.struct.rank,
                                                                cm.struct.world, C.byref(h)))
            self._handle = h   # owned from here on: a failing collective below must not leak it (__del__ destroys it)
            cm.check(_lib.load().pbn_scoredata_reduce_moments(h, cm.ref()))
            _lib.check(_lib.load().pbn_scoredata_set_comm(h, cm.ref()))
        self._handle = h
        if getattr(self, "_masks", None):
            keep = [np.ascontiguousarray(self._masks[c].astype(np.uint8)) if c in self._masks else None for c in self._cont_names]
            ptrs = (C.c_void_p * len(keep))(*[k.ctypes.data if k is not None else None for k in keep])
            _lib.check(_lib.load().pbn_scoredata_set_validity(h, ptrs))
        if self._disc_names:
            ptrs = (C.c_void_p * len(self._disc_codes))(*[c.ctypes.data for c in self._disc_codes])
            _lib.check(_lib.load().pbn_scoredata_set_discrete(h, len(self._disc_codes), ptrs, _lib.int_array(self._disc_card)))
        self._params = np.zeros(0)

    def _handle_nulls(self, rb):
        """Likelihood scores: CrossValidation / HoldOut keep only the rows that are valid in EVERY column
        (crossvalidation_adaptator.hpp:24-37).  BIC / BGe: per-candidate valid rows (bic.cpp:12-27) - the values
        under null slots are zeroed for the upload and the validity masks go to the engine."""
        import pyarrow as pa
        import pyarrow.compute as pc

        from .dataset import validity_mask

        if self._split != _lib.PBN_SPLIT_NONE:
            mask = None
            for i in range(rb.num_columns):
                m = validity_mask(rb.column(i))
                if m is not None:
                    mask = m if mask is None else (mask & m)
            return rb.filter(pa.array(mask))
        self._masks = {}
        arrays = []
        for i, f in enumerate(rb.schema):
            col = rb.column(i)
            m = validity_mask(col)
            if m is not None:
                if pa.types.is_dictionary(f.type):
                    raise ValueError("Discrete columns with nulls are not supported by the device score engine.")
                self._masks[f.name] = m
                col = pc.fill_null(col, 0.0)
            arrays.append(col)
        return pa.RecordBatch.from_arrays(arrays, schema=rb.schema)

    # -- reference surface -------------------------------------------------------------------------------
    def data(self):
        return self._df

    def is_discrete(self, variable):
        return variable in self._categories

    def has_variables(self, variables):
        if isinstance(variables, str):
            variables = [variables]
        return all(v in self._col for v in variables)

    def compatible_bn(self, model):
        return self.has_variables(model.joint_nodes() if hasattr(model, "joint_nodes") else model.nodes())

    def local_score(self, model, variable, evidence=None):
        evidence = model.parents(variable) if evidence is None else list(evidence)
        return self._batch(model, [(variable, model.node_type(variable), evidence)], self._kind)[0]

    def local_score_node_type(self, model, variable_type, variable, evidence):
        return self._batch(model, [(variable, variable_type, list(evidence))], self._kind)[0]

    def score(self, model):
        """Score::score (scores.hpp:17-24): sum of the local scores."""
        cands = [(n, model.node_type(n), model.parents(n)) for n in model.nodes()]
        return float(np.sum(self._batch(model, cands, self._kind)))

    # -- batched engine entry ---------------------------------------------------------------------------------
    def _encode(self, cands):
        var, ntype, off, par = [], [], [0], []
        for v, t, ev in cands:
            if t == UnknownFactorType():   # Score::local_score on an unknown node type: underlying_node_type(data, node)
                t = DiscreteFactorType() if self.is_discrete(v) else LinearGaussianCPDType()
            if t not in self._allowed_types:
                raise ValueError(f"Node type \"{t}\" not valid for score {type(self).__name__}")
            var.append(self._col[v])
            ntype.append(_TYPE_CODE[t])
            par.extend(self._col[e] for e in ev)
            off.append(len(par))
        return var, ntype, off, par

    def _batch(self, model, cands, kind):
        var, ntype, off, par = self._encode(cands)
        return self._batch_raw(model, var, ntype, off, par, kind)

    def _batch_raw(self, model, var, ntype, off, par, kind):
        n = len(var)
        out = np.zeros(n)
        if n == 0:
            return out
        params = self._batch_params(model)
        _lib.check(_lib.load().pbn_score_batch(self._handle, kind, n, _lib.int_array(var), _lib.int_array(ntype),
                                               _lib.int_array(off), _lib.int_array(par if par else [0]),
                                               _lib.dptr(params) if params.size else None, int(params.size), _lib.dptr(out)))
        return out

    def _batch_parts(self, model, var, ntype, off, par, kind, part, n_parts):
        """Per-part sums (len(var) x 64) of hybrid CKDE candidates on the parts dealt to rank `part` of `n_parts` - longest processing
        time first on the slices' training x test rows, the same dealing on every rank (hybrid.hip): pbn_score_batch_parts."""
        n = len(var)
        out = np.zeros((n, 64))
        if n:
            _lib.check(_lib.load().pbn_score_batch_parts(self._handle, kind, n, _lib.int_array(var), _lib.int_array(ntype), _lib.int_array(off),
                                                         _lib.int_array(par if par else [0]), int(part), int(n_parts), _lib.dptr(out)))
        return out

    def _term_regions(self, kind):
        """Regions a term of this score adds up: the CV folds, or the one hold-out region."""
        return int(getattr(self, "_k", 1)) if kind == _lib.PBN_SCORE_CVLIK else 1

    def _terms(self, what, kind, terms, values=None, regions=None):
        """The CKDE likelihood terms of the engine (pbn_score_terms*): terms = [(m, column, column, ...)] over continuous column ids.
        what = "eval" -> totals, "missing" -> flags, "put" -> install `values`, "eval_regions" -> item i = region regions[i] of terms[i]."""
        if not terms:
            return np.zeros(0) if what in ("eval", "eval_regions") else []
        off, vs, ms = [0], [], []
        for t in terms:
            ms.append(int(t[0]))
            vs.extend(int(v) for v in t[1:])
            off.append(len(vs))
        lib, n = _lib.load(), len(terms)
        args = (self._handle, kind, n, _lib.int_array(off), _lib.int_array(vs), _lib.int_array(ms))
        if what == "eval":
            out = np.zeros(n)
            _lib.check(lib.pbn_score_terms(*args, _lib.dptr(out)))
            return out
        if what == "eval_regions":
            out = np.zeros(n)
            _lib.check(lib.pbn_score_term_regions(*args, _lib.int_array([int(r) for r in regions]), _lib.dptr(out)))
            return out
        if what == "missing":
            flags = (C.c_int * n)()
            _lib.check(lib.pbn_score_terms_missing(*args, flags))
            return list(flags)
        vals = np.ascontiguousarray(values, dtype=np.float64)
        _lib.check(lib.pbn_score_terms_put(*args, _lib.dptr(vals)))
        return None

    def _batch_params(self, model):
        return self._params

    def kde_cache_stats(self):
        """(entries, sweeps) of the engine's set-function cache for CKDE likelihood scores."""
        e, w = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().pbn_scoredata_cache_stats(self._handle, C.byref(e), C.byref(w)))
        return e.value, w.value

    # -- split layout ------------------------------------------------------------------------------------------
    def _layout(self):
        """(perm, limits, n_cv, n_hold): source row of every split-ordered row, fold limits, region sizes."""
        n = self._table.num_rows
        perm = np.zeros(n, dtype=np.int32)
        limits = np.zeros(max(getattr(self, "_k", 0), 0) + 1, dtype=np.int32)
        n_cv, n_hold = C.c_int64(0), C.c_int64(0)
        _lib.check(_lib.load().pbn_scoredata_layout(self._handle, perm.ctypes.data, limits.ctypes.data if limits.size > 1 else None,
                                                    C.byref(n_cv), C.byref(n_hold)))
        return perm, limits, n_cv.value, n_hold.value

    def _take(self, rows):
        import pyarrow as pa

        if self._df is None:
            raise ValueError("This score was built from a device table; no host DataFrame is attached.")
        return self._df.take(pa.array(np.asarray(rows, dtype=np.int32)))

    def __del__(self):
        try:
            if not _lib.alive():
                return
            if getattr(self, "_handle", None):
                _lib.load().pbn_scoredata_destroy(self._handle)
                self._handle = None
        except Exception:
            pass


class BIC(_DeviceScore):
    """learning/scores/bic.cpp:12-27,108-142 (LinearGaussianCPD nodes)."""

    _kind = _lib.PBN_SCORE_BIC

    def __init__(self, df, ctx=None, table=None):
        super().__init__(df, ctx=ctx, table=table)

    def mle_lg(self, variable, evidence):
        """MLE<LinearGaussianCPD>::estimate on this score's table: (beta, variance)."""
        beta = np.zeros(len(evidence) + 1)
        var = C.c_double(0.0)
        _lib.check(_lib.load().pbn_lg_fit(self._handle, self._col[variable], _lib.int_array([self._col[e] for e in evidence] or [0]),
                                          len(evidence), _lib.dptr(beta), C.byref(var)))
        return beta, var.value


class BGe(_DeviceScore):
    """learning/scores/bge.hpp:14-234.  BGe(df, iss_mu=1, iss_w=None, nu=None)."""

    _kind = _lib.PBN_SCORE_BGE

    def __init__(self, df, iss_mu=1.0, iss_w=None, nu=None, ctx=None, table=None):
        super().__init__(df, ctx=ctx, table=table)
        ncols = len(self._names)
        if iss_w is not None and iss_w <= ncols - 1:
            raise ValueError(f"Imaginary sample size for Wishart prior must be greater than  num_columns - 1 ({ncols - 1}).")
        if nu is not None and len(nu) != ncols:
            raise ValueError(f"\"nu\" argument contains {len(nu)} elements, but DataFrame \"df\" contains {ncols} columns.")
        self._iss_mu = float(iss_mu)
        self._iss_w = float(ncols + 2 if iss_w is None else iss_w)
        # nu is given per DataFrame column, in DataFrame order (bge.hpp:36-49 indexes it by m_df.index(variable)); the engine
        # numbers the continuous columns on their own (dictionary columns follow them): hand it the continuous entries
        # in ITS order
        if nu is None:
            self._nu = None
        else:
            nu = np.asarray(nu, dtype=np.float64)
            order = list(self._df.schema.names) if self._df is not None else list(self._names)
            pos = {name: i for i, name in enumerate(order)}
            self._nu = np.asarray([nu[pos[name]] for name in self._table.names], dtype=np.float64)

    def _batch_params(self, model):
        head = [self._iss_mu, self._iss_w, float(model.num_nodes())]
        if self._nu is None:
            return np.asarray(head)
        return np.concatenate([head, self._nu])


class _LikelihoodScore(_DeviceScore):
    _allowed_types = (LinearGaussianCPDType(), CKDEType(), DiscreteFactorType())

    def _set_selector(self, construction_args):
        """`construction_args` (cv_likelihood.hpp:19-27 -> FactorType::new_factor args): see _engine_selector."""
        self._construction_args = construction_args if construction_args is not None else Arguments()
        _lib.check(_lib.load().pbn_scoredata_set_selector(self._handle, _engine_selector(construction_args)))


class CVLikelihood(_LikelihoodScore):
    """learning/scores/cv_likelihood.cpp:5-25.  CVLikelihood(df, k=10, seed=None)."""

    _kind = _lib.PBN_SCORE_CVLIK
    _split = _lib.PBN_SPLIT_CV

    def __init__(self, df, k=10, seed=None, construction_args=None, ctx=None, table=None):
        self._k = int(k)
        self._seed = _random_seed() if seed is None else int(seed)
        super().__init__(df, (self._k, self._seed, 0.0), ctx=ctx, table=table)
        self._set_selector(construction_args)

    def fold_layout(self):
        """(perm, limits): source row of every permuted row and the k+1 fold limits."""
        perm, limits, _, _ = self._layout()
        return perm, limits

    @property
    def cv(self):
        """The CrossValidation object of the reference (`CVLikelihood.cv`, pybindings_scores.cpp:553-557)."""
        return CrossValidationView(self)


class HoldoutLikelihood(_LikelihoodScore):
    """learning/scores/holdout_likelihood.cpp:8-23.  HoldoutLikelihood(df, test_ratio=0.2, seed=None)."""

    _kind = _lib.PBN_SCORE_HOLDOUT
    _split = _lib.PBN_SPLIT_HOLDOUT

    def __init__(self, df, test_ratio=0.2, seed=None, construction_args=None, ctx=None, table=None):
        self._seed = _random_seed() if seed is None else int(seed)
        super().__init__(df, (0, self._seed, float(test_ratio)), ctx=ctx, table=table)
        self._set_selector(construction_args)

    def training_data(self):
        perm, _, n_cv, _ = self._layout()
        return self._take(perm[:n_cv])

    def test_data(self):
        perm, _, n_cv, n_hold = self._layout()
        return self._take(perm[n_cv: n_cv + n_hold])


class ValidatedLikelihood(_LikelihoodScore, ValidatedScore):
    """learning/scores/validated_likelihood.hpp:12-75: local_score = CV over the hold-out training part,
    vlocal_score = hold-out likelihood; the same seed drives both splits."""

    _kind = _lib.PBN_SCORE_CVLIK
    _split = _lib.PBN_SPLIT_VALIDATED
    validated = True

    def __init__(self, df, test_ratio=0.2, k=10, seed=None, construction_args=None, ctx=None, table=None):
        self._seed = _random_seed() if seed is None else int(seed)
        self._k = int(k)
        super().__init__(df, (int(k), self._seed, float(test_ratio)), ctx=ctx, table=table)
        self._set_selector(construction_args)

    def training_data(self):
        """Hold-out training part in the hold-out shuffle order is not kept: rows are returned in CV order."""
        perm, _, n_cv, _ = self._layout()
        return self._take(perm[:n_cv])

    def validation_data(self):
        perm, _, n_cv, n_hold = self._layout()
        return self._take(perm[n_cv: n_cv + n_hold])

    @property
    def cv_lik(self):
        return _ScoreView(self, _lib.PBN_SCORE_CVLIK)

    @property
    def holdout_lik(self):
        return _ScoreView(self, _lib.PBN_SCORE_HOLDOUT)

    def vlocal_score(self, model, variable, evidence=None):
        evidence = model.parents(variable) if evidence is None else list(evidence)
        return self._batch(model, [(variable, model.node_type(variable), evidence)], _lib.PBN_SCORE_HOLDOUT)[0]

    def vlocal_score_node_type(self, model, variable_type, variable, evidence):
        return self._batch(model, [(variable, variable_type, list(evidence))], _lib.PBN_SCORE_HOLDOUT)[0]

    def vscore(self, model):
        cands = [(n, model.node_type(n), model.parents(n)) for n in model.nodes()]
        return float(np.sum(self._batch(model, cands, _lib.PBN_SCORE_HOLDOUT)))


class _ScoreView:
    """`ValidatedLikelihood.cv_lik` / `.holdout_lik`: the same engine handle evaluated with one fixed score kind."""

    def __init__(self, parent, kind):
        self._parent, self._kind = parent, kind

    def local_score(self, model, variable, evidence=None):
        evidence = model.parents(variable) if evidence is None else list(evidence)
        return self._parent._batch(model, [(variable, model.node_type(variable), evidence)], self._kind)[0]

    def local_score_node_type(self, model, variable_type, variable, evidence):
        return self._parent._batch(model, [(variable, variable_type, list(evidence))], self._kind)[0]

    def score(self, model):
        cands = [(n, model.node_type(n), model.parents(n)) for n in model.nodes()]
        return float(np.sum(self._parent._batch(model, cands, self._kind)))


class BDe(Score):
    """learning/scores/bde.{hpp,cpp}: Bayesian Dirichlet equivalent score of discrete networks, BDe(df, iss=1).  The joint
    counts come from the device (the cached row groupings of pbn_mi, `pbn_mi_counts`); the log-gamma sums over the counts
    are host arithmetic on at most prod(cardinality) numbers."""

    def __init__(self, df, iss=1.0, ctx=None):
        import pyarrow as pa

        from .dataset import as_record_batch
        from .independences import MutualInformation

        rb = as_record_batch(df)
        self._rb = rb
        self._iss = float(iss)
        self._names = [f.name for f in rb.schema]
        disc = [f.name for f in rb.schema if pa.types.is_dictionary(f.type)]
        self._counts = MutualInformation(rb.select(disc), True, ctx) if disc else None
        self._card = {d: len(rb.column(rb.schema.get_field_index(d)).dictionary) for d in disc}

    def __str__(self):
        return "BDe"

    def has_variables(self, variables):
        variables = [variables] if isinstance(variables, str) else list(variables)
        return all(v in self._names for v in variables)

    def compatible_bn(self, model):
        from .models import DiscreteFactorType

        t = model.type()
        nodes = model.joint_nodes() if model.interface_nodes() else model.nodes()
        return bool(t.is_homogeneous()) and t.default_node_type() == DiscreteFactorType() and self.has_variables(nodes)

    def data(self):
        return self._rb

    def _joint_counts(self, variables):
        for v in variables:
            if v not in self._card:
                raise ValueError(f"Variable {v} is not categorical.")
        lib = _lib.load()
        h = self._counts._handle
        _lib.check(lib.pbn_mi_set_order(h, 0, None))
        out = np.zeros(int(np.prod([self._card[v] for v in variables])))
        _lib.check(lib.pbn_mi_counts(h, len(variables), _lib.int_array([self._counts._var(v) for v in variables]), _lib.dptr(out)))
        return out

    def _bde(self, variable, parents):
        from math import lgamma

        counts = self._joint_counts([variable] + list(parents))
        card0 = self._card[variable]
        total = counts.size
        alpha = self._iss / total
        res = -total * lgamma(alpha) + float(sum(lgamma(m + alpha) for m in counts))
        if not parents:   # bde.cpp:5-21
            return res + lgamma(self._iss) - lgamma(self._iss + float(counts.sum()))
        sums = counts.reshape(-1, card0).sum(axis=1)   # bde.cpp:23-50: one term per parent configuration
        sum_alpha = alpha * card0
        return res + float(sum(lgamma(sum_alpha) - lgamma(sum_alpha + s) for s in sums))

    def local_score(self, model, variable, evidence=None):
        from .models import DiscreteFactorType

        evidence = model.parents(variable) if evidence is None else list(evidence)
        if model.node_type(variable) != DiscreteFactorType():
            raise ValueError(f"Bayesian network type \"{model.type()}\" not valid for score BDe")
        return self._bde(variable, evidence)

    def local_score_node_type(self, model, variable_type, variable, evidence):
        from .models import DiscreteFactorType

        if variable_type != DiscreteFactorType():
            raise ValueError(f"Node type \"{variable_type}\" not valid for score BDe")
        return self._bde(variable, list(evidence))


class CrossValidationView:
    """dataset::CrossValidation as seen from Python (pybindings_dataset.cpp): iterating yields (train, test) tables;
    `.indices()` yields (train_indices, test_indices) exactly as generate_cv_pair_indices (crossvalidation_adaptator.cpp)."""

    def __init__(self, score):
        self._score = score

    def indices(self):
        perm, limits, n_cv, _ = self._score._layout()
        for f in range(len(limits) - 1):
            yield (np.concatenate([perm[: limits[f]], perm[limits[f + 1]: n_cv]]), perm[limits[f]: limits[f + 1]].copy())

    def fold(self, f):
        tr, te = list(self.indices())[f]
        return self._score._take(tr), self._score._take(te)

    def __iter__(self):
        for tr, te in self.indices():
            yield self._score._take(tr), self._score._take(te)


def default_score(bn_type, df, seed=None, num_folds=10, test_holdout_ratio=0.2):
    """util/validate_options.cpp:16-58: Gaussian -> BIC; semiparametric / KDE networks -> ValidatedLikelihood."""
    if isinstance(bn_type, (GaussianNetworkType, CLGNetworkType)):
        return BIC(df)
    if isinstance(bn_type, (SemiparametricBNType, KDENetworkType)):
        return ValidatedLikelihood(df, test_holdout_ratio, num_folds, seed)
    raise ValueError("Default score not defined for this Bayesian network type.")
