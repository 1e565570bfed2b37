"""Operator sets and GreedyHillClimbing — the reference's Python surface
(/root/reference/pybnesian/pybindings/pybindings_learning/pybindings_operators.cpp:747-900,
pybindings_algorithms.cpp:75-235) over the C++ search loop of libpbn_hip (csrc/hc.hip).

The search loop, delta cache, find_max and DAG legality live in C++ (pbn_hc_estimate); every step's
local_score requests arrive here as ONE batch and are answered by the score's device engine.  When
torch.distributed is initialised (one process per GPU, RCCL over xGMI) the batch is sharded over the
ranks and the scores are exchanged with one all_gather per batch (pybnesian_amd/distributed.py); every rank
then takes the identical, deterministic decision.
"""
import ctypes as C
import os

import numpy as np

from . import _lib
from .models import (UnknownFactorType, BayesianNetwork, CKDEType, CLGNetworkType, DiscreteFactorType, GaussianNetwork, GaussianNetworkType,
                     KDENetworkType, LinearGaussianCPDType, SemiparametricBNType)
from .scores import Score, default_score

_BN_CODE = {GaussianNetworkType: _lib.PBN_BN_GAUSSIAN, SemiparametricBNType: _lib.PBN_BN_SEMIPARAMETRIC,
            KDENetworkType: _lib.PBN_BN_KDE, CLGNetworkType: _lib.PBN_BN_CLG}
_NODE_CODE = {LinearGaussianCPDType(): _lib.PBN_NODE_LG, CKDEType(): _lib.PBN_NODE_CKDE,
              DiscreteFactorType(): _lib.PBN_NODE_DISCRETE}
_NODE_FROM_CODE = {v: k for k, v in _NODE_CODE.items()}
_NODE_CODE[UnknownFactorType()] = _lib.PBN_NODE_LG   # set_unknown_node_types: a continuous column defaults to LinearGaussianCPD


def _bn_code(bn_type):
    """Engine family of a network type.  The reference's own types map one to one; a user-defined BayesianNetworkType is
    run as the built-in family with the same homogeneity / default factor, its can_have_arc() answers entering the
    engine as extra blacklist entries (see _type_blacklist)."""
    code = _BN_CODE.get(type(bn_type))
    if code is not None:
        return code
    if bn_type.is_homogeneous():
        d = bn_type.default_node_type()
        if d == LinearGaussianCPDType():
            return _lib.PBN_BN_GAUSSIAN
        if d == CKDEType():
            return _lib.PBN_BN_KDE
        if d == DiscreteFactorType():
            return _lib.PBN_BN_GAUSSIAN   # every arc between discrete nodes is legal: the unrestricted family; the score decides
        raise ValueError(f"The hill-climbing engine has no factor family for the default node type {d} of {bn_type}.")
    return _lib.PBN_BN_SEMIPARAMETRIC


def _type_blacklist(model):
    """Arcs a user-defined network type rejects through can_have_arc(model, source, target) (BayesianNetwork.hpp:273):
    asked once per ordered pair on the start model.  The built-in types' rules live in the engine itself."""
    if type(model.type()) in _BN_CODE:
        return []
    joint = model.nodes() + (model.interface_nodes() if hasattr(model, "interface_nodes") else [])
    return [(s, t) for s in joint for t in model.nodes() if s != t and not model.type().can_have_arc(model, s, t)]


class Operator:
    _kind = -1

    def __init__(self, delta):
        self._delta = delta

    def delta(self):
        return self._delta

    def _key(self):
        raise NotImplementedError

    def __eq__(self, other):  # operators.hpp:108-119: operator type + endpoints (+ node type), delta ignored
        return isinstance(other, Operator) and self._key() == other._key()

    def __hash__(self):
        return hash(self._key())


class ArcOperator(Operator):
    """operators.hpp:122-140: an operator on one arc; AddArc, RemoveArc and FlipArc derive from it."""

    def __init__(self, source, target, delta):
        super().__init__(delta)
        self._source, self._target = source, target

    def _key(self):
        return (type(self).__name__, self._source, self._target)

    def source(self):
        return self._source

    def target(self):
        return self._target

    def nodes_changed(self, model=None):
        return [self._target]


class AddArc(ArcOperator):
    _kind = 0

    def opposite(self, model=None):
        return RemoveArc(self._source, self._target, -self._delta)

    def apply(self, model):
        model.add_arc(self._source, self._target)

    def __repr__(self):
        return f"AddArc({self._source} -> {self._target}; {self._delta})"


class RemoveArc(ArcOperator):
    _kind = 1

    def opposite(self, model=None):
        return AddArc(self._source, self._target, -self._delta)

    def apply(self, model):
        model.remove_arc(self._source, self._target)

    def __repr__(self):
        return f"RemoveArc({self._source} -> {self._target}; {self._delta})"


class FlipArc(ArcOperator):
    _kind = 2

    def opposite(self, model=None):
        return FlipArc(self._target, self._source, -self._delta)

    def nodes_changed(self, model=None):
        return [self._source, self._target]

    def apply(self, model):
        model.flip_arc(self._source, self._target)

    def __repr__(self):
        return f"FlipArc({self._source} -> {self._target}; {self._delta})"


class ChangeNodeType(Operator):
    _kind = 3

    def _key(self):
        return ("ChangeNodeType", self._node, self._type)

    def opposite(self, model):
        """operators.hpp:214-216: ChangeNodeType(node, model.node_type(node), -delta)."""
        return ChangeNodeType(self._node, model.node_type(self._node), -self._delta)

    def nodes_changed(self, model=None):
        return [self._node]

    def __init__(self, node, node_type, delta):
        super().__init__(delta)
        self._node, self._type = node, node_type

    def node(self):
        return self._node

    def node_type(self):
        return self._type

    def apply(self, model):
        model.set_node_type(self._node, self._type)

    def __repr__(self):
        return f"ChangeNodeType({self._node} -> {self._type}; {self._delta})"


class OperatorTabuSet:
    """learning/operators/operators.hpp:258-293."""

    def __init__(self):
        self._ops = set()

    def insert(self, op):
        self._ops.add(op)

    def contains(self, op):
        return op in self._ops

    def clear(self):
        self._ops.clear()

    def empty(self):
        return not self._ops


class LocalScoreCache:
    """LocalScoreCache (learning/operators/operators.hpp:295-338): one local score per node.  `LocalScoreCache()` /
    `LocalScoreCache(model)` own their values and fill them through the score (one batch for device scores); the
    object returned by `OperatorSet.local_score_cache()` is a view of the engine's cache."""

    def __init__(self, model=None):
        self._binding = None
        self._values, self._idx = None, {}
        if isinstance(model, _EngineBinding):
            self._binding = model
        elif model is not None:
            self._reset(model)

    def _reset(self, model):
        self._idx = {n: i for i, n in enumerate(model.nodes())}
        self._values = np.zeros(len(self._idx))

    def _fill(self, model, score, validated):
        from .scores import _DeviceScore

        if self._binding is not None:
            raise ValueError("This LocalScoreCache is a view of an operator set's cache.")
        if self._values is None or len(self._idx) != model.num_nodes():
            self._reset(model)
        nodes = model.nodes()
        if isinstance(score, _DeviceScore):
            cands = [(n, model.node_type(n), model.parents(n)) for n in nodes]
            vals = score._batch(model, cands, _lib.PBN_SCORE_HOLDOUT if validated else score._kind)
        else:
            fn = score.vlocal_score if validated else score.local_score
            vals = [fn(model, n, model.parents(n)) for n in nodes]
        for n, v in zip(nodes, vals):
            self._values[self._idx[n]] = v

    def cache_local_scores(self, model, score):
        self._fill(model, score, False)

    def cache_vlocal_scores(self, model, score):
        self._fill(model, score, True)

    def _update(self, model, score, variable, validated):
        if self._binding is not None:
            raise ValueError("This LocalScoreCache is a view of an operator set's cache.")
        fn = score.vlocal_score if validated else score.local_score
        self._values[self._idx[variable]] = fn(model, variable, model.parents(variable))

    def update_local_score(self, model, score, variable):
        self._update(model, score, variable, False)

    def update_vlocal_score(self, model, score, variable):
        self._update(model, score, variable, True)

    def local_score(self, model, variable):
        if self._binding is not None:
            return float(self._binding.local_scores()[self._binding.idx[variable]])
        return float(self._values[self._idx[variable]])

    def sum(self):
        if self._binding is not None:
            return float(np.sum(self._binding.local_scores()))
        return float(np.sum(self._values)) if self._values is not None else 0.0


class Callback:
    """learning/algorithms/callbacks/callback.hpp: subclass and implement call(model, operator, score, iteration).
    Called with iteration 0 and operator None before the search, after every applied operator, and once more with
    operator None on the returned model (hillclimbing.hpp:127,180,195)."""

    def call(self, model, operator, score, iteration):
        raise NotImplementedError("Tried to call pure virtual function \"Callback::call\"")


class SaveModel(Callback):
    """learning/algorithms/callbacks/save_model.hpp: saves the model of every iteration as `folder/000123.pickle`."""

    def __init__(self, folder_name):
        self._folder = folder_name

    def call(self, model, operator, score, iteration):
        import os

        model.save(os.path.join(self._folder, "%06d" % iteration), False)


class _EngineBinding:
    """One pbn_hc handle bound to (operator sets, score, model node order)."""

    def __init__(self, sets, score, model, arc_blacklist=(), arc_whitelist=(), type_blacklist=(), type_whitelist=(),
                 max_indegree=0, max_iters=2 ** 31 - 1, epsilon=0.0, patience=0):
        from .distributed import sharded_batch
        from .scores import ValidatedScore, _DeviceScore

        device_score = isinstance(score, _DeviceScore) or hasattr(score, "_batch_raw")  # batched engine protocol
        # joint index space of the engine: the nodes, then the interface nodes of a conditional network
        self.n_nodes = len(model.nodes())
        self.nodes = model.nodes() + (model.interface_nodes() if hasattr(model, "interface_nodes") else [])
        self.idx = {n: i for i, n in enumerate(self.nodes)}
        self.score, self.model_type = score, model.type()
        n = self.n_nodes
        col_of_node = [score._col[v] for v in self.nodes] if device_score else list(range(len(self.nodes)))
        idx = self.idx

        def node_index(v):
            if v not in idx:
                raise ValueError(f"Node {v} not present in the graph.")   # util::check_arc_list (validate_whitelists.hpp:24-33)
            return idx[v]

        def pairs(lst, second=node_index):
            flat = []
            for a, b in lst:
                flat += [node_index(a), second(b)]
            return flat

        for st in sets:
            if isinstance(st, ArcOperatorSet):
                max_indegree = max_indegree or st.max_indegree
                arc_blacklist = list(arc_blacklist) + [a for a in st.blacklist if a not in arc_blacklist]
                arc_whitelist = list(arc_whitelist) + [a for a in st.whitelist if a not in arc_whitelist]
            else:
                type_blacklist = list(type_blacklist) + [a for a in st.type_blacklist if a not in type_blacklist]
                type_whitelist = list(type_whitelist) + [a for a in st.type_whitelist if a not in type_whitelist]
        self._keep = []

        def arr(values):
            a = _lib.int_array(values if values else [0])
            self._keep.append(a)
            return a

        cfg = _lib.HCConfig()
        cfg.n_nodes, cfg.bn_type = n, _bn_code(model.type())
        arc_blacklist = list(arc_blacklist) + [a for a in _type_blacklist(model) if a not in set(map(tuple, arc_blacklist))]
        cfg.node_types = arr(self.node_type_codes(model))
        arcs = pairs(model.arcs())
        cfg.n_arcs, cfg.arcs = len(arcs) // 2, arr(arcs)
        bl, wl = pairs(arc_blacklist), pairs(arc_whitelist)
        cfg.n_arc_blacklist, cfg.arc_blacklist = len(bl) // 2, arr(bl)
        cfg.n_arc_whitelist, cfg.arc_whitelist = len(wl) // 2, arr(wl)
        tbl, twl = pairs(type_blacklist, _NODE_CODE.__getitem__), pairs(type_whitelist, _NODE_CODE.__getitem__)
        cfg.n_type_blacklist, cfg.type_blacklist = len(tbl) // 2, arr(tbl)
        cfg.n_type_whitelist, cfg.type_whitelist = len(twl) // 2, arr(twl)
        cfg.op_arcs = int(any(isinstance(st, ArcOperatorSet) for st in sets))
        cfg.op_node_type = int(any(isinstance(st, ChangeNodeTypeSet) for st in sets))
        cfg.arcs_first = int(isinstance(sets[0], ArcOperatorSet))
        cfg.max_indegree, cfg.max_iters = int(max_indegree), int(min(max_iters, 2 ** 31 - 1))
        cfg.epsilon, cfg.patience = float(epsilon), int(patience)
        cfg.n_interface = len(self.nodes) - self.n_nodes
        cfg.validated = int(isinstance(score, ValidatedScore) or bool(getattr(score, "validated", False)))
        # near ties of CKDE likelihood scores on fp64 tables (pbn_hc_config.near_tie_abs): 4 x the sum-only sweeps' error budget per
        # log-density (3.3e-7) x the test rows a local score sums over; PBN_NEAR_TIE=0 switches the check off (the reference has none)
        cfg.near_tie_abs = 0.0
        if device_score and os.environ.get("PBN_NEAR_TIE", "1") != "0" and getattr(score, "_kind", None) in (_lib.PBN_SCORE_CVLIK, _lib.PBN_SCORE_HOLDOUT):
            try:
                if score._table.dtype == _lib.PBN_F64:
                    _, _, n_cv, n_hold = score._layout()
                    cfg.near_tie_abs = 4.0 * 3.3e-7 * float(n_cv if score._kind == _lib.PBN_SCORE_CVLIK else n_hold)
            except Exception:   # a score without a device layout: no check
                cfg.near_tie_abs = 0.0
        self.cfg = cfg
        self.errors = []
        self.batch_hook = None

        def on_batch(_user, validated, n_cand, var, ntype, off, par, out):
            try:
                off_l = [off[i] for i in range(n_cand + 1)]
                var_l = [col_of_node[var[i]] for i in range(n_cand)]
                nt_l = [ntype[i] for i in range(n_cand)]
                par_l = [col_of_node[par[i]] for i in range(off_l[-1])]
                precise, validated = bool(validated & 2), validated & 1   # bit 1: "at full precision" (the search's near-tie check)
                if device_score:
                    kind = _lib.PBN_SCORE_HOLDOUT if validated else score._kind
                    if precise:
                        _lib.check(_lib.load().pbn_scoredata_set_precise(score._handle, 1))
                    try:
                        res = sharded_batch(score, model, var_l, nt_l, off_l, par_l, kind)
                    finally:
                        if precise:
                            _lib.check(_lib.load().pbn_scoredata_set_precise(score._handle, 0))
                else:  # Python-derived Score: one trampoline call per candidate, as the reference does
                    fn = score.vlocal_score_node_type if validated else score.local_score_node_type
                    plain = score.vlocal_score if validated else score.local_score
                    res = []
                    for i in range(n_cand):
                        v, ev = self.nodes[var_l[i]], [self.nodes[j] for j in par_l[off_l[i]: off_l[i + 1]]]
                        if model.type().homogeneous:
                            res.append(float(plain(model, v, ev)))
                        else:
                            res.append(float(fn(model, _NODE_FROM_CODE[nt_l[i]], v, ev)))
                if self.batch_hook is not None:
                    self.batch_hook(n_cand)
                for i in range(n_cand):
                    out[i] = res[i]
                return 0
            except Exception as ex:  # surfaced after the C call returns
                self.errors.append(ex)
                return 1

        self.callback = _lib.HC_SCORE_FN(on_batch)
        self.handle = None

    def node_type_codes(self, model):
        # set_unknown_node_types (hillclimbing.hpp:81-93): dictionary columns are DiscreteFactor nodes
        is_disc = getattr(self.score, "is_discrete", lambda v: False)
        return [_lib.PBN_NODE_DISCRETE if is_disc(v) else _NODE_CODE[model.node_type(v)] for v in self.nodes]

    def check(self, rc):
        if self.errors:
            err, self.errors = self.errors[0], []
            raise err
        _lib.check(rc)

    # ---- stateful interface -----------------------------------------------------------------------------
    def create(self):
        h = C.c_void_p()
        self.check(_lib.load().pbn_hc_create(C.byref(self.cfg), self.callback, None, C.byref(h)))
        self.handle = h

    def sync(self, model):
        if model.nodes() + (model.interface_nodes() if hasattr(model, "interface_nodes") else []) != self.nodes:
            raise ValueError("The model's nodes changed since cache_scores().")
        arcs = []
        for a, b in model.arcs():
            arcs += [self.idx[a], self.idx[b]]
        self.check(_lib.load().pbn_hc_set_model(self.handle, len(arcs) // 2, _lib.int_array(arcs or [0]),
                                               _lib.int_array(self.node_type_codes(model))))

    def cache_scores(self):
        self.check(_lib.load().pbn_hc_cache_scores(self.handle))

    def find_max(self, tabu=None):
        flat = []
        for op in (tabu._ops if tabu is not None else ()):
            if isinstance(op, ChangeNodeType):
                flat += [3, self.idx[op.node()], _NODE_CODE[op.node_type()], 0]
            else:
                flat += [op._kind, self.idx[op.source()], self.idx[op.target()], 0]
        out = (C.c_int * 3)()
        delta = C.c_double(0.0)
        self.check(_lib.load().pbn_hc_find_max(self.handle, len(flat) // 4, _lib.int_array(flat or [0]), out, C.byref(delta)))
        return self.make_op(out[0], out[1], out[2], delta.value)

    def make_op(self, kind, a, b, d):
        if kind < 0:
            return None
        if kind == 3:
            return ChangeNodeType(self.nodes[a], _NODE_FROM_CODE[b], d)
        return (AddArc, RemoveArc, FlipArc)[kind](self.nodes[a], self.nodes[b], d)

    def update_scores(self, variables):
        ids = [self.idx[v] for v in variables]
        self.check(_lib.load().pbn_hc_update_scores(self.handle, len(ids), _lib.int_array(ids or [0])))

    def local_scores(self):
        out = np.zeros(self.n_nodes)
        self.check(_lib.load().pbn_hc_get(self.handle, _lib.dptr(out), None, None))
        return out

    def deltas(self):
        n = self.n_nodes
        arcs, types = np.zeros((len(self.nodes), n), order="F"), np.zeros(n)
        self.check(_lib.load().pbn_hc_get(self.handle, None, _lib.dptr(arcs), _lib.dptr(types)))
        return arcs, types

    def close(self):
        if self.handle is not None:
            _lib.load().pbn_hc_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            if not _lib.alive():
                return
            self.close()
        except Exception:
            pass


class OperatorSet:
    """OperatorSet interface of learning/operators/operators.hpp:340-355 on the C++ engine.  The engine keeps
    its own copy of the model; find_max / update_scores re-synchronise it with the caller's model first."""

    def __init__(self):
        self._binding = None

    def _sets(self):
        return [self]

    def _invalidate(self):
        if self._binding is not None:
            self._binding.close()
        self._binding = None

    def _require(self):
        if self._binding is None or self._binding.handle is None:
            raise ValueError("Local cache not initialized. Call cache_scores() before find_max()")
        return self._binding

    def cache_scores(self, model, score):
        if not score.compatible_bn(model):
            raise ValueError("BayesianNetwork is not compatible with the score.")
        self._invalidate()
        self._binding = _EngineBinding(_flatten_ops(self), score, model)
        self._binding.create()
        self._binding.cache_scores()

    def find_max(self, model):
        b = self._require()
        b.sync(model)
        return b.find_max()

    def find_max_tabu(self, model, tabu_set):
        b = self._require()
        b.sync(model)
        return b.find_max(tabu_set)

    def update_scores(self, model, score, variables):
        b = self._require()
        b.sync(model)
        b.update_scores(list(variables))

    def local_score_cache(self):
        return LocalScoreCache(self._require())

    def delta(self):
        """(arc delta matrix n x n [source, target], node-type delta vector) of the cached scores."""
        return self._require().deltas()

    def finished(self):
        self._invalidate()

    # OperatorSet setters (operators.hpp:346-353): every set accepts all of them, each keeps what concerns it
    def set_arc_blacklist(self, blacklist):
        for st in self._sets():
            if st is not self:
                st.set_arc_blacklist(blacklist)

    def set_arc_whitelist(self, whitelist):
        for st in self._sets():
            if st is not self:
                st.set_arc_whitelist(whitelist)

    def set_max_indegree(self, max_indegree):
        for st in self._sets():
            if st is not self:
                st.set_max_indegree(max_indegree)

    def set_type_blacklist(self, type_blacklist):
        for st in self._sets():
            if st is not self:
                st.set_type_blacklist(type_blacklist)

    def set_type_whitelist(self, type_whitelist):
        for st in self._sets():
            if st is not self:
                st.set_type_whitelist(type_whitelist)


class ArcOperatorSet(OperatorSet):
    def __init__(self, blacklist=(), whitelist=(), max_indegree=0):
        super().__init__()
        self.blacklist, self.whitelist, self.max_indegree = list(blacklist), list(whitelist), int(max_indegree)

    def set_arc_blacklist(self, blacklist):
        self.blacklist = list(blacklist)
        self._invalidate()

    def set_arc_whitelist(self, whitelist):
        self.whitelist = list(whitelist)
        self._invalidate()

    def set_max_indegree(self, max_indegree):
        self.max_indegree = int(max_indegree)
        self._invalidate()


class ChangeNodeTypeSet(OperatorSet):
    def __init__(self, type_blacklist=(), type_whitelist=()):
        super().__init__()
        self.type_blacklist, self.type_whitelist = list(type_blacklist), list(type_whitelist)

    def set_type_blacklist(self, type_blacklist):
        self.type_blacklist = list(type_blacklist)
        self._invalidate()

    def set_type_whitelist(self, type_whitelist):
        self.type_whitelist = list(type_whitelist)
        self._invalidate()


class OperatorPool(OperatorSet):
    def __init__(self, opsets):
        super().__init__()
        if not opsets:
            raise ValueError("op_sets argument cannot be empty.")
        self.opsets = list(opsets)

    def _sets(self):
        return self.opsets

    def _invalidate(self):
        super()._invalidate()


def _flatten_ops(operators):
    sets = operators._sets() if isinstance(operators, OperatorSet) else [operators]
    kinds = [type(s) for s in sets]
    for k in kinds:
        if k not in (ArcOperatorSet, ChangeNodeTypeSet):
            raise ValueError("Only ArcOperatorSet and ChangeNodeTypeSet are implemented on device.")
    return sets


class HCResult:
    """Extra information about the last GreedyHillClimbing.estimate call."""

    def __init__(self):
        self.iterations = 0
        self.cells_scored = 0
        self.local_score_evals = 0
        self.trace = []


class GreedyHillClimbing:
    def __init__(self):
        self.last = HCResult()

    def estimate(self, operators, score, start, arc_blacklist=(), arc_whitelist=(), type_blacklist=(), type_whitelist=(),
                 callback=None, max_indegree=0, max_iters=2 ** 31 - 1, epsilon=0.0, patience=0, verbose=0, batch_hook=None):
        if not score.compatible_bn(start):
            raise ValueError("BayesianNetwork is not compatible with the score.")  # hillclimbing.hpp:292-294
        sets = _flatten_ops(operators)
        binding = _EngineBinding(sets, score, start, arc_blacklist, arc_whitelist, type_blacklist, type_whitelist,
                                 max_indegree, max_iters, epsilon, patience)
        binding.batch_hook = batch_hook
        nodes, n = binding.nodes, binding.n_nodes
        interface = nodes[n:]
        out_arcs = (C.c_int * (2 * len(nodes) * n))()
        out_n = C.c_int(0)
        out_types = (C.c_int * n)()
        stats = _lib.HCStats()
        cap = 4 * n * n + 1024
        trace = (C.c_int * (4 * cap))()
        tdelta = (C.c_double * cap)()
        stats.trace_capacity, stats.trace, stats.trace_delta = cap, trace, tdelta
        if callback is not None:
            def on_iter(_user, iteration, op, delta, n_arcs, arcs, ntypes):
                try:
                    cur_arcs = [(nodes[arcs[2 * i]], nodes[arcs[2 * i + 1]]) for i in range(n_arcs)]
                    cur_types = [(nodes[i], _NODE_FROM_CODE[ntypes[i]]) for i in range(n)] + [(v, start.node_type(v)) for v in interface]
                    cur = start.clone()._set_structure(cur_arcs, cur_types)
                    callback.call(cur, binding.make_op(op[0], op[1], op[2], delta), score, iteration)
                    return 0
                except Exception as ex:
                    binding.errors.append(ex)
                    return 1

            binding.iter_callback = _lib.HC_ITER_FN(on_iter)
            binding.cfg.on_iter = binding.iter_callback
        rc = _lib.load().pbn_hc_estimate(C.byref(binding.cfg), binding.callback, None, out_arcs, C.byref(out_n), out_types,
                                         C.byref(stats))
        binding.check(rc)
        res_types = [(nodes[i], _NODE_FROM_CODE[out_types[i]]) for i in range(n)] + [(v, start.node_type(v)) for v in interface]
        res_arcs = [(nodes[out_arcs[2 * i]], nodes[out_arcs[2 * i + 1]]) for i in range(out_n.value)]
        result = start.clone()._set_structure(res_arcs, res_types)   # hillclimbing.hpp:296: the result is a clone of start
        self.last = HCResult()
        self.last.iterations = stats.iterations
        self.last.cells_scored = stats.cells_scored
        self.last.local_score_evals = stats.local_score_evals
        self.last.near_tie_redos = int(stats.near_tie_redos)
        self.last.trace = [binding.make_op(trace[4 * i], trace[4 * i + 1], trace[4 * i + 2], tdelta[i]) for i in range(stats.trace_len)]
        return result


class MMHC:
    """Max-Min Hill-Climbing (learning/algorithms/mmhc.cpp:77-160): MMPC over all variables with the given independence
    test, symmetric correction, every arc outside the candidate parents-and-children blacklisted, then the greedy
    hill-climb of this package.  `MMHC().estimate(hypot_test, operators, score, nodes=[], bn_type=GaussianNetworkType(),
    arc_blacklist=[], arc_whitelist=[], edge_blacklist=[], edge_whitelist=[], type_blacklist=[], type_whitelist=[],
    callback=None, max_indegree=0, max_iters=2**31-1, epsilon=0, patience=0, alpha=0.05, verbose=0)`."""

    def __init__(self):
        self.last_cpcs, self.last_tests, self.hc = None, 0, GreedyHillClimbing()

    def estimate(self, hypot_test, operators, score, nodes=(), bn_type=None, arc_blacklist=(), arc_whitelist=(), edge_blacklist=(),
                 edge_whitelist=(), type_blacklist=(), type_whitelist=(), callback=None, max_indegree=0, max_iters=2 ** 31 - 1,
                 epsilon=0.0, patience=0, alpha=0.05, verbose=0):
        from .independences import mmpc_cpcs, validate_restrictions
        from .models import GaussianNetworkType

        bn_type = bn_type if bn_type is not None else GaussianNetworkType()
        nodes = list(nodes) if nodes else list(hypot_test.variable_names())
        if not hypot_test.has_variables(nodes):
            raise ValueError("IndependenceTest do not contain all the variables in nodes list.")
        bn = bn_type.new_bn(nodes)
        if not score.compatible_bn(bn):
            raise ValueError("BayesianNetwork is not compatible with the score.")
        if not score.has_variables(nodes):
            raise ValueError("Score do not contain all the variables in nodes list.")
        _, a_wl, e_bl, e_wl = validate_restrictions(nodes, arc_blacklist, arc_whitelist, edge_blacklist, edge_whitelist)
        cpcs, self.last_tests = mmpc_cpcs(hypot_test, nodes, alpha, a_wl, e_bl, e_wl, symmetric=True)
        self.last_cpcs = cpcs
        allowed = [set(c) for c in cpcs]
        hc_blacklist = []
        for i in range(len(nodes) - 1):               # create_hc_blacklist (mmhc.cpp:24-42)
            for j in range(i + 1, len(nodes)):
                if nodes[j] not in allowed[i]:
                    hc_blacklist += [(nodes[i], nodes[j]), (nodes[j], nodes[i])]
        hc_blacklist += [tuple(a) for a in arc_blacklist]
        hc_whitelist = [(nodes[s], nodes[t]) for s, t in a_wl]
        return self.hc.estimate(operators, score, bn, hc_blacklist, hc_whitelist, type_blacklist, type_whitelist, callback,
                                max_indegree, max_iters, epsilon, patience, verbose)


    def estimate_conditional(self, hypot_test, operators, score, nodes, interface_nodes=(), bn_type=None, arc_blacklist=(),
                             arc_whitelist=(), edge_blacklist=(), edge_whitelist=(), type_blacklist=(), type_whitelist=(),
                             callback=None, max_indegree=0, max_iters=2 ** 31 - 1, epsilon=0.0, patience=0, alpha=0.05, verbose=0):
        """MMHC::estimate_conditional (mmhc.cpp:162-244): MMPC over the conditional graph, then the hill-climb of a
        conditional network; arcs between nodes outside each other's CPC and interface -> node arcs outside the node's
        CPC are blacklisted (create_conditional_hc_blacklist, :44-75)."""
        from .independences import mmpc_cpcs, validate_restrictions
        from .models import GaussianNetworkType

        nodes, interface_nodes = list(nodes), list(interface_nodes)
        if not nodes:
            raise ValueError("Node list cannot be empty to train a Conditional Bayesian network.")
        bn_type = bn_type if bn_type is not None else GaussianNetworkType()
        if not interface_nodes:
            return self.estimate(hypot_test, operators, score, nodes, bn_type, arc_blacklist, arc_whitelist, edge_blacklist,
                                 edge_whitelist, type_blacklist, type_whitelist, callback, max_indegree, max_iters, epsilon,
                                 patience, alpha, verbose).conditional_bn()
        if not hypot_test.has_variables(nodes) or not hypot_test.has_variables(interface_nodes):
            raise ValueError("IndependenceTest do not contain all the variables in nodes/interface_nodes lists.")
        if not score.has_variables(nodes) or not score.has_variables(interface_nodes):
            raise ValueError("Score do not contain all the variables in nodes list.")
        bn = bn_type.new_cbn(nodes, interface_nodes)
        if not score.compatible_bn(bn):
            raise ValueError("BayesianNetwork is not compatible with the score.")
        joint = nodes + interface_nodes
        _, a_wl, e_bl, e_wl = validate_restrictions(joint, arc_blacklist, arc_whitelist, edge_blacklist, edge_whitelist)
        cpcs, self.last_tests = mmpc_cpcs(hypot_test, nodes, alpha, a_wl, e_bl, e_wl, symmetric=True, interface_nodes=interface_nodes)
        self.last_cpcs = cpcs
        allowed = [set(c) for c in cpcs]
        hc_blacklist = []
        for i in range(len(nodes) - 1):
            for j in range(i + 1, len(nodes)):
                if nodes[j] not in allowed[i]:
                    hc_blacklist += [(nodes[i], nodes[j]), (nodes[j], nodes[i])]
        for i, node in enumerate(nodes):
            hc_blacklist += [(inode, node) for inode in interface_nodes if inode not in allowed[i]]
        hc_blacklist += [tuple(a) for a in arc_blacklist]
        hc_whitelist = [(joint[s], joint[t]) for s, t in a_wl]
        return self.hc.estimate(operators, score, bn, hc_blacklist, hc_whitelist, type_blacklist, type_whitelist, callback,
                                max_indegree, max_iters, epsilon, patience, verbose)


def hc(df, bn_type=None, start=None, score=None, operators=None, arc_blacklist=(), arc_whitelist=(), type_blacklist=(),
       type_whitelist=(), callback=None, max_indegree=0, max_iters=2 ** 31 - 1, epsilon=0.0, patience=0, seed=None,
       num_folds=10, test_holdout_ratio=0.2, verbose=0):
    """hc() convenience wrapper (learning/algorithms/hillclimbing.cpp:26-90, util/validate_options.cpp:16-91)."""
    from .dataset import as_record_batch

    rb = as_record_batch(df)
    if start is None:
        if bn_type is None:
            raise ValueError("\"bn_type\" or \"start\" parameter must be specified.")
        start = bn_type.new_bn([f.name for f in rb.schema])
    bn_type = start.type()
    if isinstance(score, str) or score is None:
        from . import scores as S

        table = {None: None, "bic": lambda: S.BIC(rb), "bge": lambda: S.BGe(rb),
                 "cv-lik": lambda: S.CVLikelihood(rb, num_folds, seed),
                 "holdout-lik": lambda: S.HoldoutLikelihood(rb, test_holdout_ratio, seed),
                 "validated-lik": lambda: S.ValidatedLikelihood(rb, test_holdout_ratio, num_folds, seed)}
        if score is None:
            score = default_score(bn_type, rb, seed, num_folds, test_holdout_ratio)
        elif score in table:
            score = table[score]()
        else:
            raise ValueError("\"score\" should be one of: bic, bge, cv-lik, holdout-lik, validated-lik")
    if operators is None:
        operators = ["arcs"] if bn_type.homogeneous else ["arcs", "node_type"]
    if isinstance(operators, (list, tuple)) and all(isinstance(o, str) for o in operators):
        sets = []
        for o in operators:
            if o == "arcs":
                sets.append(ArcOperatorSet())
            elif o == "node_type":
                sets.append(ChangeNodeTypeSet())
            else:
                raise ValueError("\"operators\" should be a list containing: arcs, node_type")
        operators = sets[0] if len(sets) == 1 else OperatorPool(sets)
    return GreedyHillClimbing().estimate(operators, score, start, arc_blacklist, arc_whitelist, type_blacklist, type_whitelist,
                                         callback, max_indegree, max_iters, epsilon, patience, verbose)
