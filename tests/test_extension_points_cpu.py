"""CPU tier: the reference's Python extension points on top of the C++ search engine (docs/source/extending.rst,
pybindings_scores.cpp:281-440, callbacks/callback.hpp, factors/arguments.hpp): a Score / ValidatedScore written in
Python drives pbn_hc_estimate candidate by candidate, Callback sees every iteration, LocalScoreCache and Arguments
behave as the reference's.  No device is involved."""
import numpy as np
import pytest

import pybnesian_amd as pbn
from test_hc_cpu import CKDE, LG, TableScore, run_product


class PyScore(pbn.Score):
    """A user-defined score: the deterministic table of test_hc_cpu behind the documented Score interface."""

    def __init__(self, ts, names):
        self.ts, self.idx = ts, {n: i for i, n in enumerate(names)}

    def has_variables(self, variables):
        variables = [variables] if isinstance(variables, str) else variables
        return all(v in self.idx for v in variables)

    def compatible_bn(self, model):
        return self.has_variables(model.nodes())

    def _code(self, t):
        return CKDE if t == pbn.CKDEType() else LG

    def local_score(self, model, variable, evidence=None):
        evidence = model.parents(variable) if evidence is None else evidence
        return self.local_score_node_type(model, model.node_type(variable), variable, evidence)

    def local_score_node_type(self, model, variable_type, variable, evidence):
        return self.ts.raw(self.idx[variable], self._code(variable_type), [self.idx[e] for e in evidence])


class PyValidatedScore(PyScore, pbn.ValidatedScore):
    def vlocal_score(self, model, variable, evidence=None):
        evidence = model.parents(variable) if evidence is None else evidence
        return self.vlocal_score_node_type(model, model.node_type(variable), variable, evidence)

    def vlocal_score_node_type(self, model, variable_type, variable, evidence):
        return self.ts.raw(self.idx[variable], self._code(variable_type), [self.idx[e] for e in evidence], True)


@pytest.mark.parametrize("seed", [1, 4])
def test_python_score_drives_engine(ensure_built, seed):
    n = 7
    names = [f"n{i}" for i in range(n)]
    ref = run_product(TableScore(n, seed), "gaussian", n)
    got = run_product(PyScore(TableScore(n, seed), names), "gaussian", n)
    assert got[2] == ref[2] and sorted(got[0]) == sorted(ref[0])
    # semiparametric + validated: the node-type overload and vlocal_score are used
    ref = run_product(TableScore(n, seed, validated=True), "spbn", n, [0] * n, op_types=True, patience=2)
    got = run_product(PyValidatedScore(TableScore(n, seed, validated=True), names), "spbn", n, [0] * n, op_types=True, patience=2)
    assert got[2] == ref[2] and sorted(got[0]) == sorted(ref[0]) and got[1] == ref[1]


def test_incompatible_and_abstract_scores(ensure_built):
    class Nothing(pbn.Score):
        def has_variables(self, variables):
            return False

        def compatible_bn(self, model):
            return False

    with pytest.raises(ValueError, match="BayesianNetwork is not compatible with the score."):
        pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), Nothing(), pbn.GaussianNetwork(["a", "b"]))
    with pytest.raises(NotImplementedError, match="pure virtual"):
        pbn.Score().local_score(pbn.GaussianNetwork(["a"]), "a", [])

    class Broken(Nothing):
        def compatible_bn(self, model):
            return True

        def local_score(self, model, variable, evidence=None):
            raise KeyError("boom")

    with pytest.raises(KeyError, match="boom"):   # exceptions of the Python score surface unchanged
        pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), Broken(), pbn.GaussianNetwork(["a", "b"]))


def test_callback_sequence(ensure_built):
    n = 6
    names = [f"n{i}" for i in range(n)]
    score = PyScore(TableScore(n, 2), names)
    seen = []

    class Recorder(pbn.Callback):
        def call(self, model, operator, sc, iteration):
            assert sc is score
            seen.append((iteration, operator, sorted(model.arcs())))

    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names), callback=Recorder())
    trace = hc.last.trace
    assert seen[0] == (0, None, [])                               # hillclimbing.hpp:127
    assert [s[0] for s in seen[1:-1]] == list(range(1, len(trace) + 1))
    assert [s[1] for s in seen[1:-1]] == trace                    # operator equality ignores delta (operators.hpp:108)
    assert all(abs(s[1].delta() - t.delta()) == 0 for s, t in zip(seen[1:-1], trace))
    model = pbn.GaussianNetwork(names)
    for it, op, arcs in seen[1:-1]:
        op.apply(model)
        assert sorted(model.arcs()) == arcs                       # the callback sees the model AFTER the operator
    assert seen[-1][1] is None and seen[-1][2] == sorted(res.arcs())   # hillclimbing.hpp:195

    class Failing(pbn.Callback):
        def call(self, model, operator, sc, iteration):
            if iteration == 2:
                raise RuntimeError("stop here")

    with pytest.raises(RuntimeError, match="stop here"):
        pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names), callback=Failing())


def test_local_score_cache(ensure_built):
    n = 5
    names = [f"n{i}" for i in range(n)]
    score = PyValidatedScore(TableScore(n, 3, validated=True), names)
    model = pbn.GaussianNetwork(names, [("n0", "n1"), ("n2", "n1")])
    lc = pbn.LocalScoreCache(model)
    lc.cache_local_scores(model, score)
    want = [score.local_score(model, v) for v in names]
    assert [lc.local_score(model, v) for v in names] == want and lc.sum() == pytest.approx(sum(want))
    model.add_arc("n3", "n1")
    lc.update_local_score(model, score, "n1")
    assert lc.local_score(model, "n1") == score.local_score(model, "n1") != want[1]
    lv = pbn.LocalScoreCache()
    lv.cache_vlocal_scores(model, score)
    assert lv.sum() == pytest.approx(score.vscore(model))
    lv.update_vlocal_score(model, score, "n0")
    # the view handed out by an operator set tracks the engine's cache
    ops = pbn.ArcOperatorSet()
    ops.cache_scores(model, score)
    view = ops.local_score_cache()
    assert [view.local_score(model, v) for v in names] == [score.local_score(model, v) for v in names]
    with pytest.raises(ValueError):
        view.cache_local_scores(model, score)
    ops.finished()


def test_operator_set_setters(ensure_built):
    arcs, types = pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()
    pool = pbn.OperatorPool([arcs, types])
    pool.set_arc_blacklist([("a", "b")])
    pool.set_arc_whitelist([("b", "c")])
    pool.set_max_indegree(2)
    pool.set_type_blacklist([("a", pbn.CKDEType())])
    pool.set_type_whitelist([("b", pbn.LinearGaussianCPDType())])
    assert arcs.blacklist == [("a", "b")] and arcs.whitelist == [("b", "c")] and arcs.max_indegree == 2
    assert types.type_blacklist == [("a", pbn.CKDEType())] and types.type_whitelist == [("b", pbn.LinearGaussianCPDType())]
    arcs.set_type_blacklist([("a", pbn.CKDEType())])   # accepted and ignored, as in the reference
    types.set_max_indegree(3)
    with pytest.raises(ValueError, match="cannot be empty"):
        pbn.OperatorPool([])


def test_arguments_lookup():
    lg, ck = pbn.LinearGaussianCPDType(), pbn.CKDEType()
    a = pbn.Arguments({"x": ((1, 2), {"param": 3}), ck: pbn.Kwargs(bandwidth_selector="s"), ("x", ck): (pbn.Args(7), pbn.Kwargs(k=1)),
                       "y": {"p": 1}, lg: pbn.Args(5)})
    assert a.args("x", ck) == ((7,), {"k": 1})                     # (name, type) first
    assert a.args("x", lg) == (((1, 2), {"param": 3}), {})          # then name; a plain tuple is *args
    assert a.args("z", ck) == ((), {"bandwidth_selector": "s"})     # then type
    assert a.args("y", lg) == ((), {"p": 1}) and a.args("z", lg) == ((5,), {})
    assert pbn.Arguments().args("q", lg) == ((), {}) and repr(a) == "Arguments"
    with pytest.raises(ValueError, match="Key value"):
        pbn.Arguments({3: (1,)})
