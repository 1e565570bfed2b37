"""GPU tier: dynamic networks (learning/algorithms/dmmhc.cpp, models/DynamicBayesianNetwork.cpp) - DMMHC on a synthetic
vector autoregression, conditional MMHC, and the log-likelihood bookkeeping of DynamicBayesianNetwork."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def var_process(n, seed):
    """a_t = 0.8 a_{t-1} + e; b_t = 0.5 b_{t-1} + 0.9 a_t + e; c_t = -0.7 b_{t-1} + e   (order 1)."""
    rng = np.random.default_rng(seed)
    a, b, c = np.zeros(n), np.zeros(n), np.zeros(n)
    for t in range(1, n):
        a[t] = 0.8 * a[t - 1] + rng.normal(scale=0.5)
        b[t] = 0.5 * b[t - 1] + 0.9 * a[t] + rng.normal(scale=0.5)
        c[t] = -0.7 * b[t - 1] + rng.normal(scale=0.5)
    return pd.DataFrame({"a": a, "b": b, "c": c})


def test_dmmhc_recovers_var_structure(pbn):
    df = var_process(6000, 1)
    ddf = pbn.DynamicDataFrame(df, 1)
    test = pbn.DynamicLinearCorrelation(ddf)
    score = pbn.DynamicBIC(ddf)
    assert test.has_variables(["a", "c"]) and score.has_variables("b") and not score.has_variables("z")
    dm = pbn.DMMHC()
    dbn = dm.estimate(test, pbn.ArcOperatorSet(), score, markovian_order=1, alpha=0.01)
    assert dbn.variables() == ["a", "b", "c"] and dbn.markovian_order() == 1
    tr = dbn.transition_bn()
    arcs = set(tr.arcs())
    assert {("a_t_1", "a_t_0"), ("b_t_1", "b_t_0"), ("b_t_1", "c_t_0")} <= arcs
    assert ("a_t_0", "b_t_0") in arcs or ("b_t_0", "a_t_0") in arcs
    assert all(t.endswith("_t_0") for _, t in arcs)                      # nothing enters the lagged (interface) nodes
    assert tr.interface_nodes() == ["a_t_1", "b_t_1", "c_t_1"]
    # the same transition structure from a direct conditional MMHC
    mm = pbn.MMHC()
    direct = mm.estimate_conditional(test.transition_tests(), pbn.ArcOperatorSet(), score.transition_score(),
                                     ["a_t_0", "b_t_0", "c_t_0"], ["a_t_1", "b_t_1", "c_t_1"], alpha=0.01)
    assert sorted(direct.arcs()) == sorted(tr.arcs())
    # likelihood bookkeeping
    dbn.fit(df)
    assert dbn.fitted()
    test_df = var_process(500, 2)
    ll = dbn.logl(test_df)
    assert ll.shape == (500,) and abs(ll.sum() - dbn.slogl(test_df)) <= 1e-9 * abs(ll.sum())
    tt = pbn.DynamicDataFrame(test_df, 1).transition_df()
    assert np.allclose(ll[1:], sum(tr.cpd(v).logl(tt) for v in tr.nodes()))
    with pytest.raises(ValueError, match="Not enough information"):
        pbn.DynamicBayesianNetwork(["a", "b", "c"], 2, bn_type=pbn.GaussianNetworkType()).fitted() or dbn.logl(test_df.iloc[:0])


def test_dmmhc_order2_semiparametric(pbn):
    df = var_process(3000, 5)
    ddf = pbn.DynamicDataFrame(df, 2)
    dbn = pbn.DMMHC().estimate(pbn.DynamicLinearCorrelation(ddf), pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]),
                               pbn.DynamicValidatedLikelihood(ddf, 0.2, 3, 0), bn_type=pbn.SemiparametricBNType(), markovian_order=2,
                               max_iters=4, alpha=0.01)
    st, tr = dbn.static_bn(), dbn.transition_bn()
    assert st.nodes() == ["a_t_1", "a_t_2", "b_t_1", "b_t_2", "c_t_1", "c_t_2"] and tr.num_interface_nodes() == 6
    for s, t in st.arcs():                                                  # static blacklist: no recent -> older arcs
        assert not (s.endswith("_t_1") and t.endswith("_t_2"))
    assert tr.num_arcs() <= 4 and all(t.endswith("_t_0") for _, t in tr.arcs())


def test_dbn_fit_logl_reference_recipe(pbn, golden):
    """/root/reference/tests/models/DynamicBayesianNetwork_test.py:78-230 re-typed: fit bookkeeping and logl / slogl
    against the row-by-row normal-density recipe (static part: row i scored by the factors of slice order - i)."""
    import re

    from scipy.stats import norm

    from helpers import frame

    df = frame(golden["train10k"]).iloc[:1000].reset_index(drop=True)
    test_df = frame(golden["train500"]).iloc[:100].reset_index(drop=True)
    variables = ["a", "b", "c", "d"]
    gbn = pbn.DynamicGaussianNetwork(variables, 2)
    assert not gbn.fitted() and not gbn.static_bn().fitted() and not gbn.transition_bn().fitted()
    ddf = pbn.DynamicDataFrame(df, 2)
    gbn2 = pbn.DynamicGaussianNetwork(variables, 2)
    gbn2.static_bn().fit(ddf.static_df())
    assert not gbn2.fitted() and gbn2.static_bn().fitted() and not gbn2.transition_bn().fitted()
    gbn2.transition_bn().fit(ddf.transition_df())
    assert gbn2.fitted()
    st, tr = gbn.static_bn(), gbn.transition_bn()
    for t in (2, 1):
        st.add_arc(f"a_t_{t}", f"c_t_{t}")
        st.add_arc(f"b_t_{t}", f"c_t_{t}")
        st.add_arc(f"c_t_{t}", f"d_t_{t}")
        for v in variables:
            tr.add_arc(f"{v}_t_{t}", f"{v}_t_0")
    gbn.fit(df)
    assert gbn.fitted()

    def lg(cpd, value, evidence_values):
        m = cpd.beta[0] + np.dot(cpd.beta[1:], evidence_values)
        return norm(m, np.sqrt(cpd.variance)).logpdf(value)

    def lookup(e, base_row):
        m = re.search(r"(.*)_t_(\d+)", e)
        return m[1], int(m[2])

    want = np.zeros(test_df.shape[0])
    order = gbn.markovian_order()
    for i in range(order):
        for v in variables:
            cpd = st.cpd(f"{v}_t_{order - i}")
            ev = [test_df.loc[order - lookup(e, i)[1], lookup(e, i)[0]] for e in cpd.evidence()]
            want[i] += lg(cpd, test_df.loc[i, v], ev)
    for i in range(order, test_df.shape[0]):
        for v in variables:
            cpd = tr.cpd(f"{v}_t_0")
            ev = [test_df.loc[i - lookup(e, i)[1], lookup(e, i)[0]] for e in cpd.evidence()]
            want[i] += lg(cpd, test_df.loc[i, v], ev)
    ll = gbn.logl(test_df)
    assert np.all(np.isclose(want, ll))
    assert np.isclose(gbn.slogl(test_df), want.sum())


def test_dbn_sample(pbn):
    """DynamicBayesianNetwork.sample (DynamicBayesianNetwork.cpp:259-466): the first `order` rows are one sample of the
    static network, every later row is drawn factor by factor from the window of the previous rows with seed + i."""
    import pyarrow as pa

    df = var_process(4000, 5)
    dbn = pbn.DynamicGaussianNetwork(["a", "b", "c"], 2)
    for s, t in (("a_t_1", "a_t_0"), ("b_t_1", "b_t_0"), ("a_t_0", "b_t_0"), ("b_t_1", "c_t_0")):
        dbn.transition_bn().add_arc(s, t)
    dbn.static_bn().add_arc("a_t_2", "a_t_1")
    with pytest.raises(ValueError, match="not fitted"):
        dbn.sample(10, 0)
    dbn.fit(df)
    s = dbn.sample(300, seed=7)
    assert s.schema.names == ["a", "b", "c"] and s.num_rows == 300 and all(c.type == pa.float64() for c in s.columns)
    again = dbn.sample(300, seed=7)
    assert s.equals(again) and not s.equals(dbn.sample(300, seed=8))
    sp = s.to_pandas()
    st = dbn.static_bn().sample(1, 7).to_pandas()
    for v in "abc":
        assert sp[v][0] == st[f"{v}_t_2"][0] and sp[v][1] == st[f"{v}_t_1"][0]
    # row i of a variable = its factor's one-value sample given the window, seed + i
    tr = dbn.transition_bn()
    for i in (2, 17, 299):
        window = {f"{v}_t_{k}": [sp[v][i - k]] for v in "abc" for k in range(3)}
        for v in "abc":
            cpd = tr.cpd(f"{v}_t_0")
            ev = pd.DataFrame({e: window[e] for e in cpd.evidence()}) if cpd.evidence() else None
            assert sp[v][i] == cpd.sample(1, ev, 7 + i)[0].as_py()
    # the process keeps its dynamics: strong positive lag-1 autocorrelation of a, b follows a
    long = dbn.sample(3000, seed=1).to_pandas()
    assert np.corrcoef(long["a"][1:], long["a"][:-1])[0, 1] > 0.6
    assert np.corrcoef(long["a"], long["b"])[0, 1] > 0.5
    assert dbn.sample(1, 3).num_rows == 1 and dbn.sample(0, 3).num_rows == 0
    with pytest.raises(ValueError, match="non-negative"):
        dbn.sample(-1, 0)


@pytest.mark.parametrize("order,spbn", [(1, False), (2, False), (1, True)])
def test_dmmhc_matches_the_serial_restatement(pbn, order, spbn):
    """DMMHC with device scores against oracle/dmmhc_oracle.py (dmmhc.cpp:34-118 + DynamicScoreAdaptator, scores.hpp:74-101, restated
    over mmpc_oracle / hc_oracle / the oracle's local scores): the same CPC sets and numbers of independence tests in both phases,
    the same operator sequences of the two hill-climbs (static network; conditional transition network with the static nodes as
    interface), the same arcs and node types.  Likelihood scores (cross-validated LinearGaussian; validated CKDE / LinearGaussian with
    node-type changes): no score-equivalence ties, so the sequences are comparable step by step."""
    from oracle import dmmhc_oracle, oracle

    rng = np.random.default_rng(7 + order)
    n = 2500
    a, b, c = np.zeros(n), np.zeros(n), np.zeros(n)
    for t in range(2, n):
        a[t] = 0.7 * a[t - 1] + rng.normal(scale=0.5)
        b[t] = 0.4 * b[t - 1] + np.tanh(1.5 * a[t]) + (0.5 * a[t - 2] if order > 1 else 0.0) + rng.normal(scale=0.4)
        c[t] = -0.8 * b[t - 1] + 0.3 * c[t - 1] + rng.normal(scale=0.5)
    df = pd.DataFrame({"a": a, "b": b, "c": c})
    names = list(df.columns)
    ddf = pbn.DynamicDataFrame(df, order)
    k, seed, ratio, alpha = 3, 0, 0.2, 0.01
    if spbn:
        score = pbn.DynamicValidatedLikelihood(ddf, ratio, k, seed)
        ops, bn_type = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), pbn.SemiparametricBNType()

        def make_score(table):
            return (lambda v, t, ps: oracle.validated_cv_likelihood(table[:, [v] + list(ps)], "ckde" if t == 1 else "lg", ratio, k, seed),
                    lambda v, t, ps: oracle.holdout_likelihood(table[:, [v] + list(ps)], "ckde" if t == 1 else "lg", ratio, seed))
    else:
        score = pbn.DynamicCVLikelihood(ddf, k, seed)
        ops, bn_type = pbn.ArcOperatorSet(), pbn.GaussianNetworkType()

        def make_score(table):
            return (lambda v, t, ps: oracle.cv_likelihood(table[:, [v] + list(ps)], "lg", k, seed)), None
    dm = pbn.DMMHC()
    dbn = dm.estimate(pbn.DynamicLinearCorrelation(ddf), ops, score, bn_type=bn_type, markovian_order=order, alpha=alpha, max_indegree=3)
    want = dmmhc_oracle.dmmhc(df.to_numpy(), names, order, alpha, make_score, bn_type=1 if spbn else 0, op_types=spbn, max_indegree=3)

    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    tcode = {pbn.LinearGaussianCPDType(): 0, pbn.CKDEType(): 1}
    for phase, net, cpcs, tests, search in (("static", dbn.static_bn(), dm.static_cpcs, dm.static_tests, dm.static_search),
                                            ("transition", dbn.transition_bn(), dm.transition_cpcs, dm.transition_tests, dm.transition_search)):
        w = want[phase]
        joint = w["nodes"] + w.get("interface", [])
        col = {v: i for i, v in enumerate(joint)}
        assert net.nodes() == w["nodes"]
        assert [[col[v] for v in cpc] for cpc in cpcs] == w["cpcs"], phase
        assert tests == w["tests"], phase
        got = [(3, col[op.node()], tcode[op.node_type()]) if isinstance(op, pbn.ChangeNodeType) else (kinds[type(op)], col[op.source()], col[op.target()])
               for op in search.trace]
        assert got == [tuple(t) for t in w["trace"]], (phase, got, w["trace"])
        assert np.allclose([op.delta() for op in search.trace], w["deltas"], rtol=1e-6, atol=1e-6)
        assert sorted((col[s], col[t]) for s, t in net.arcs()) == w["arcs"], phase
        assert [tcode[net.node_type(v)] for v in w["nodes"]] == w["types"][: len(w["nodes"])]
        assert search.cells_scored == w["cells"]
    assert dbn.transition_bn().num_arcs() >= 3
