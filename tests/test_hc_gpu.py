"""GPU tier: end-to-end GreedyHillClimbing with device scores against the serial oracle restatement with
CPU oracle scores, on the reference's 4-variable test table, plus the behavioural assertions of
/root/reference/tests/learning/algorithms/hillclimbing_test.py:8-58."""
import numpy as np
import pandas as pd
import pytest

from helpers import COLS, frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def oracle_scores(data, kind, **kw):
    from oracle import oracle

    def score(v, t, ps, validated=False):
        cols = data[:, [v] + list(ps)]
        if kind == "bic":
            return oracle.bic_lg(cols)
        if kind == "bge":
            return oracle.bge(cols, data.shape[1])
        nt = "lg" if t == 0 else "ckde"
        if kind == "cv":
            return oracle.cv_likelihood(cols, nt, kw["k"], kw["seed"])
        if kind == "validated":
            if validated:
                return oracle.holdout_likelihood(cols, nt, kw["ratio"], kw["seed"])
            return oracle.validated_cv_likelihood(cols, nt, kw["ratio"], kw["k"], kw["seed"])
        raise ValueError(kind)

    return score


def trace_of(pbn, hc, names):
    idx = {n: i for i, n in enumerate(names)}
    out = []
    for op in hc.last.trace:
        if isinstance(op, pbn.ChangeNodeType):
            out.append((3, idx[op.node()], 0 if op.node_type() == pbn.LinearGaussianCPDType() else 1))
        else:
            out.append(({pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}[type(op)], idx[op.source()], idx[op.target()]))
    return out


def test_first_arc_and_epsilon_behaviour(pbn, golden):
    """hillclimbing_test.py:8-58: the delta of the first operator equals score(res) - score(start);
    blacklisting it yields the reversed arc with the same delta (score equivalence); a large epsilon
    returns the start graph."""
    df = frame(golden["train10k"][:1000])
    bic = pbn.BIC(df)
    start = pbn.GaussianNetwork(COLS)
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), bic, start, max_iters=1)
    assert res.num_arcs() == 1
    op = hc.last.trace[0]
    assert isinstance(op, pbn.AddArc)
    assert np.isclose(op.delta(), bic.score(res) - bic.score(start))
    res2 = hc.estimate(pbn.ArcOperatorSet(), bic, start, max_iters=1, arc_blacklist=[(op.source(), op.target())])
    op2 = hc.last.trace[0]
    assert (op2.source(), op2.target()) == (op.target(), op.source())
    assert np.isclose(op2.delta(), op.delta())
    res3 = hc.estimate(pbn.ArcOperatorSet(), bic, start, epsilon=op.delta() + 0.01)
    assert res3.num_arcs() == 0
    full = hc.estimate(pbn.ArcOperatorSet(), bic, start)
    assert full.num_arcs() >= 3 and bic.score(full) > bic.score(start)


@pytest.mark.parametrize("kind", ["bic", "bge"])
def test_hc_gaussian_end_to_end_vs_oracle(pbn, golden, kind):
    """Score-equivalent orientations tie mathematically for BIC/BGe; a blacklist of one orientation per pair
    removes the ties so that the operator sequence is decided by real score differences and must agree."""
    from oracle import hc_oracle

    data = golden["train10k"][:2000]
    df = frame(data)
    score = pbn.BIC(df) if kind == "bic" else pbn.BGe(df)
    bl = [(j, i) for i in range(4) for j in range(4) if j > i]  # only "forward" arcs allowed
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(COLS), arc_blacklist=[(COLS[a], COLS[b]) for a, b in bl])
    o_arcs, _, o_trace, info = hc_oracle.estimate(4, 0, oracle_scores(data, kind), arc_blacklist=bl)
    assert trace_of(pbn, hc, COLS) == [t[:3] for t in o_trace]
    assert sorted((COLS.index(s), COLS.index(t)) for s, t in res.arcs()) == sorted(o_arcs)
    assert hc.last.cells_scored == info["cells_scored"]
    deltas = [op.delta() for op in hc.last.trace]
    assert np.allclose(deltas, [t[3] for t in o_trace], rtol=1e-6)


def test_hc_spbn_cv_end_to_end_vs_oracle(pbn, golden):
    """Semiparametric network, CVLikelihood, arcs + node-type operators (config C3 in miniature)."""
    from oracle import hc_oracle

    data = golden["train10k"][:600]
    df = frame(data)
    score = pbn.CVLikelihood(df, 4, 1)
    hc = pbn.GreedyHillClimbing()
    ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
    res = hc.estimate(ops, score, pbn.SemiparametricBN(COLS), max_indegree=2)
    sc = oracle_scores(data, "cv", k=4, seed=1)
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(4, 1, sc, op_types=True, max_indegree=2)
    assert trace_of(pbn, hc, COLS) == [t[:3] for t in o_trace]
    assert sorted((COLS.index(s), COLS.index(t)) for s, t in res.arcs()) == sorted(o_arcs)
    assert [0 if res.node_type(c) == pbn.LinearGaussianCPDType() else 1 for c in COLS] == o_types
    assert np.allclose([op.delta() for op in hc.last.trace], [t[3] for t in o_trace], rtol=1e-5, atol=1e-7)


def test_hc_validated_patience_end_to_end_vs_oracle(pbn, golden):
    from oracle import hc_oracle

    data = golden["train10k"][:500]
    df = frame(data)
    score = pbn.ValidatedLikelihood(df, 0.2, 3, 2)
    hc = pbn.GreedyHillClimbing()
    ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
    res = hc.estimate(ops, score, pbn.SemiparametricBN(COLS), patience=2, max_indegree=2)
    sc = oracle_scores(data, "validated", ratio=0.2, k=3, seed=2)
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(4, 1, lambda v, t, ps: sc(v, t, ps, False),
                                                        lambda v, t, ps: sc(v, t, ps, True), op_types=True, patience=2,
                                                        max_indegree=2)
    assert trace_of(pbn, hc, COLS) == [t[:3] for t in o_trace]
    assert sorted((COLS.index(s), COLS.index(t)) for s, t in res.arcs()) == sorted(o_arcs)
    assert [0 if res.node_type(c) == pbn.LinearGaussianCPDType() else 1 for c in COLS] == o_types


def test_hc_convenience_wrapper(pbn, golden):
    df = frame(golden["train10k"][:800])
    res = pbn.hc(df, bn_type=pbn.GaussianNetworkType(), score="bic", max_iters=2)
    assert res.num_arcs() == 2
    with pytest.raises(ValueError):
        pbn.hc(df, bn_type=pbn.GaussianNetworkType(), score="nope")


def test_conditional_gaussian_network_hc_gpu(pbn):
    """A conditional network on device scores: interface columns are parents only; learned arcs never enter them, and
    fit / slogl of the result use factors of the nodes only (ConditionalBayesianNetwork, BayesianNetwork.hpp:140-222)."""
    rng = np.random.default_rng(12)
    n = 6000
    x = rng.normal(size=n)
    y = rng.normal(size=n)
    a = 1.5 * x + rng.normal(scale=0.5, size=n)
    b = -a + 0.8 * y + rng.normal(scale=0.5, size=n)
    c = rng.normal(size=n)
    df = pd.DataFrame({"a": a, "b": b, "c": c, "x": x, "y": y})
    score = pbn.BIC(df)
    start = pbn.ConditionalGaussianNetwork(["a", "b", "c"], ["x", "y"])
    res = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), score, start)
    arcs = set(res.arcs())
    assert ("x", "a") in arcs and ("y", "b") in arcs and (("a", "b") in arcs or ("b", "a") in arcs)
    assert all(t in ("a", "b", "c") for _, t in arcs) and res.interface_nodes() == ["x", "y"]
    assert score.score(res) > score.score(start)
    res.fit(df)
    want = sum(res.cpd(v).slogl(df) for v in ("a", "b", "c"))
    assert abs(res.slogl(df) - want) <= 1e-9 * abs(want)
    with pytest.raises(ValueError, match="not compatible"):
        pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), pbn.BIC(df[["a", "b", "c"]]), start)


def test_conditional_spbn_cv_hc_vs_oracle(pbn):
    """SURVEY.md §8 f2 on the device: a conditional semiparametric network (two interface nodes) learnt with CVLikelihood
    device scores over arcs and node types against hc_oracle's conditional restatement (operators.cpp:134-256,365-437;
    operators.hpp:526-578) driven by oracle scores: same operator trace, arcs, node types and number of scored cells."""
    from oracle import hc_oracle, oracle

    rng = np.random.default_rng(7)
    n = 700
    x = rng.normal(size=n)
    y = rng.normal(size=n)
    a = np.tanh(1.5 * x) + rng.normal(scale=0.4, size=n)
    b = -0.8 * a + 0.8 * y + rng.normal(scale=0.5, size=n)
    c = 0.5 * b + rng.normal(scale=0.7, size=n)
    names = ["a", "b", "c", "x", "y"]                 # nodes first, interface nodes behind them (the engine's id order)
    df = pd.DataFrame({"a": a, "b": b, "c": c, "x": x, "y": y})
    data = df.to_numpy()
    score = pbn.CVLikelihood(df, 3, 2)
    start = pbn.ConditionalSemiparametricBN(["a", "b", "c"], ["x", "y"])
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), score, start, max_indegree=2)

    def sc(v, t, ps):
        return oracle.cv_likelihood(data[:, [v] + list(ps)], "lg" if t == 0 else "ckde", 3, 2)

    o_arcs, o_types, o_trace, info = hc_oracle.estimate(3, 1, sc, op_types=True, max_indegree=2, n_interface=2)
    assert trace_of(pbn, hc, names) == [t[:3] for t in o_trace]
    assert sorted((names.index(s), names.index(t)) for s, t in res.arcs()) == sorted(o_arcs)
    assert [0 if res.node_type(v) == pbn.LinearGaussianCPDType() else 1 for v in ("a", "b", "c")] == list(o_types)[:3]
    assert hc.last.cells_scored == info["cells_scored"] and len(o_trace) >= 3
    assert np.allclose([op.delta() for op in hc.last.trace], [t[3] for t in o_trace], rtol=1e-5, atol=1e-7)
