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

import numpy as np

from . import _lib
from .models import (BayesianNetwork, CKDEType, CLGNetworkType, DiscreteFactorType, GaussianNetwork, GaussianNetworkType,
                     KDENetworkType, LinearGaussianCPDType, SemiparametricBNType)
from .scores import Score, default_score

_BN_CODE = {GaussianNetworkType: _lib.PBN_BN_GAUSSIAN, SemiparametricBNType: _lib.PBN_BN_SEMIPARAMETRIC,
            KDENetworkType: _lib.PBN_BN_KDE, CLGNetworkType: _lib.PBN_BN_CLG}
_NODE_CODE = {LinearGaussianCPDType(): _lib.PBN_NODE_LG, CKDEType(): _lib.PBN_NODE_CKDE,
              DiscreteFactorType(): _lib.PBN_NODE_DISCRETE}
_NODE_FROM_CODE = {v: k for k, v in _NODE_CODE.items()}


class Operator:
    def __init__(self, delta):
        self._delta = delta

    def delta(self):
        return self._delta


class AddArc(Operator):
    def __init__(self, source, target, delta):
        super().__init__(delta)
        self._source, self._target = source, target

    def source(self):
        return self._source

    def target(self):
        return self._target

    def apply(self, model):
        model.add_arc(self._source, self._target)

    def __repr__(self):
        return f"AddArc({self._source} -> {self._target}; {self._delta})"


class RemoveArc(AddArc):
    def apply(self, model):
        model.remove_arc(self._source, self._target)

    def __repr__(self):
        return f"RemoveArc({self._source} -> {self._target}; {self._delta})"


class FlipArc(AddArc):
    def apply(self, model):
        model.flip_arc(self._source, self._target)

    def __repr__(self):
        return f"FlipArc({self._source} -> {self._target}; {self._delta})"


class ChangeNodeType(Operator):
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


class OperatorSet:
    pass


class ArcOperatorSet(OperatorSet):
    def __init__(self, blacklist=(), whitelist=(), max_indegree=0):
        self.blacklist, self.whitelist, self.max_indegree = list(blacklist), list(whitelist), int(max_indegree)


class ChangeNodeTypeSet(OperatorSet):
    def __init__(self, type_blacklist=(), type_whitelist=()):
        self.type_blacklist, self.type_whitelist = list(type_blacklist), list(type_whitelist)


class OperatorPool(OperatorSet):
    def __init__(self, opsets):
        if not opsets:
            raise ValueError("op_sets argument cannot be empty.")
        self.opsets = list(opsets)


def _flatten_ops(operators):
    sets = operators.opsets if isinstance(operators, OperatorPool) else [operators]
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
        nodes = start.nodes()
        idx = {n: i for i, n in enumerate(nodes)}
        n = len(nodes)
        bn_code = _BN_CODE[type(start.type())]
        col_of_node = [score._col[v] for v in nodes]

        def pairs(lst, second=idx.__getitem__):
            flat = []
            for a, b in lst:
                flat += [idx[a], second(b)]
            return flat

        arc_bl, arc_wl = list(arc_blacklist), list(arc_whitelist)
        type_bl, type_wl = list(type_blacklist), list(type_whitelist)
        for s in sets:
            if isinstance(s, ArcOperatorSet):
                max_indegree = max_indegree or s.max_indegree
        # set_unknown_node_types (hillclimbing.hpp:81-93): dictionary columns are DiscreteFactor nodes
        is_disc = getattr(score, "is_discrete", lambda v: False)
        node_types = [_lib.PBN_NODE_DISCRETE if is_disc(v) else _NODE_CODE[start.node_type(v)] for v in nodes]
        arcs = pairs(start.arcs())
        cfg = _lib.HCConfig()
        keep = []

        def arr(values):
            a = _lib.int_array(values if values else [0])
            keep.append(a)
            return a

        cfg.n_nodes, cfg.bn_type = n, bn_code
        cfg.node_types = arr(node_types)
        cfg.n_arcs, cfg.arcs = len(arcs) // 2, arr(arcs)
        bl, wl = pairs(arc_bl), pairs(arc_wl)
        cfg.n_arc_blacklist, cfg.arc_blacklist = len(bl) // 2, arr(bl)
        cfg.n_arc_whitelist, cfg.arc_whitelist = len(wl) // 2, arr(wl)
        tbl, twl = pairs(type_bl, _NODE_CODE.__getitem__), pairs(type_wl, _NODE_CODE.__getitem__)
        cfg.n_type_blacklist, cfg.type_blacklist = len(tbl) // 2, arr(tbl)
        cfg.n_type_whitelist, cfg.type_whitelist = len(twl) // 2, arr(twl)
        cfg.op_arcs = int(any(isinstance(s, ArcOperatorSet) for s in sets))
        cfg.op_node_type = int(any(isinstance(s, ChangeNodeTypeSet) for s in sets))
        cfg.arcs_first = int(isinstance(sets[0], ArcOperatorSet))
        cfg.max_indegree, cfg.max_iters = int(max_indegree), int(min(max_iters, 2 ** 31 - 1))
        cfg.epsilon, cfg.patience = float(epsilon), int(patience)
        cfg.validated = int(getattr(score, "validated", False))

        errors = []
        from .distributed import sharded_batch

        def on_batch(_user, validated, n_cand, var, ntype, off, par, out):
            try:
                off_l = [off[i] for i in range(n_cand + 1)]
                var_l = [col_of_node[var[i]] for i in range(n_cand)]
                nt_l = [ntype[i] for i in range(n_cand)]
                par_l = [col_of_node[par[i]] for i in range(off_l[-1])]
                kind = _lib.PBN_SCORE_HOLDOUT if validated else score._kind
                res = sharded_batch(score, start, var_l, nt_l, off_l, par_l, kind)
                if batch_hook is not None:
                    batch_hook(n_cand)
                for i in range(n_cand):
                    out[i] = res[i]
                return 0
            except Exception as ex:  # surfaced after pbn_hc_estimate returns
                errors.append(ex)
                return 1

        cb = _lib.HC_SCORE_FN(on_batch)
        out_arcs = (C.c_int * (2 * n * n))()
        out_n = C.c_int(0)
        out_types = (C.c_int * n)()
        stats = _lib.HCStats()
        cap = 4 * n * n + 1024
        trace = (C.c_int * (4 * cap))()
        tdelta = (C.c_double * cap)()
        stats.trace_capacity, stats.trace, stats.trace_delta = cap, trace, tdelta
        rc = _lib.load().pbn_hc_estimate(C.byref(cfg), cb, None, out_arcs, C.byref(out_n), out_types, C.byref(stats))
        if errors:
            raise errors[0]
        _lib.check(rc)
        res_types = [(nodes[i], _NODE_FROM_CODE[out_types[i]]) for i in range(n)]
        res_arcs = [(nodes[out_arcs[2 * i]], nodes[out_arcs[2 * i + 1]]) for i in range(out_n.value)]
        result = BayesianNetwork(start.type(), nodes, res_arcs, [] if start.type().homogeneous else res_types)
        self.last = HCResult()
        self.last.iterations = stats.iterations
        self.last.cells_scored = stats.cells_scored
        self.last.local_score_evals = stats.local_score_evals
        ops = []
        for i in range(stats.trace_len):
            kind, a, b = trace[4 * i], trace[4 * i + 1], trace[4 * i + 2]
            d = tdelta[i]
            if kind == 3:
                ops.append(ChangeNodeType(nodes[a], _NODE_FROM_CODE[b], d))
            else:
                ops.append((AddArc, RemoveArc, FlipArc)[kind](nodes[a], nodes[b], d))
        self.last.trace = ops
        return result


def hc(df, bn_type=None, start=None, score=None, operators=None, arc_blacklist=(), arc_whitelist=(), type_blacklist=(),
       type_whitelist=(), callback=None, max_indegree=0, max_iters=2 ** 31 - 1, epsilon=0.0, patience=0, seed=None,
       num_folds=10, test_holdout_ratio=0.2, verbose=0):
    """hc() convenience wrapper (learning/algorithms/hillclimbing.cpp:26-90, util/validate_options.cpp:16-91)."""
    from .dataset import as_record_batch

    rb = as_record_batch(df)
    if start is None:
        if bn_type is None:
            raise ValueError("\"bn_type\" or \"start\" parameter must be specified.")
        start = BayesianNetwork(bn_type, [f.name for f in rb.schema])
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
