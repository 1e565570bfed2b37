"""GPU tier: hybrid scoring — CLinearGaussianCPD / HCKDE through
DiscreteAdaptator semantics, bic_clg, bic_discrete, DiscreteFactor likelihoods — against the oracle's
per-slice restatement, plus a small CLG / semiparametric hill-climb with discrete nodes."""
import numpy as np
import pandas as pd
import pytest

from helpers import RTOL_F64

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def make_hybrid(n, seed=0):
    """In the spirit of tests/helpers/util_test.py:103-141 (generate_hybrid_data): discrete A (2), B (3),
    continuous x | A, y | x, B, z | x, y with configuration-dependent means."""
    rng = np.random.default_rng(seed)
    A = rng.integers(0, 2, size=n)
    B = (rng.random(n) < np.where(A == 0, 0.3, 0.6)).astype(int) + (rng.random(n) < 0.2)
    x = rng.normal(loc=np.where(A == 0, -1.0, 2.0), scale=1.0)
    y = 0.7 * x + np.array([0.0, 3.0, -2.0])[B] + rng.normal(scale=0.5, size=n)
    z = np.tanh(x) - 0.4 * y + rng.normal(scale=0.3, size=n)
    df = pd.DataFrame({
        "A": pd.Categorical.from_codes(A, ["a0", "a1"]),
        "B": pd.Categorical.from_codes(B, ["b0", "b1", "b2"]),
        "x": x, "y": y, "z": z,
    })
    return df, {"A": A.astype(np.int32), "B": B.astype(np.int32)}, {"A": 2, "B": 3}


def close(a, b, rtol=RTOL_F64):
    if np.isinf(a) or np.isinf(b):
        return a == b
    return abs(a - b) <= rtol * max(abs(b), 1e-12)


def test_bic_clg_and_discrete(pbn, oracle):
    df, codes, cards = make_hybrid(3000)
    bic = pbn.BIC(df)
    net = pbn.CLGNetwork(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])
    for var, dpar, cpar in [("x", ["A"], []), ("y", ["B"], ["x"]), ("z", ["A", "B"], ["x", "y"]), ("y", ["B", "A"], [])]:
        got = bic.local_score(net, var, dpar + cpar)
        want = oracle.bic_clg(df[[var] + cpar].to_numpy(), [codes[d] for d in dpar], [cards[d] for d in dpar])
        assert close(got, want), (var, dpar, cpar, got, want)
        # parent order between discrete and continuous evidence does not matter
        assert close(bic.local_score(net, var, cpar + dpar), want)
    for var, par in [("A", []), ("B", ["A"]), ("A", ["B"])]:
        got = bic.local_score(net, var, par)
        want = oracle.bic_discrete(codes[var], cards[var], [codes[p] for p in par], [cards[p] for p in par])
        assert close(got, want), (var, par)
    with pytest.raises(ValueError, match="non-discrete"):
        bic.local_score(net, "A", ["x"])


@pytest.mark.parametrize("node_type", ["lg", "ckde"])
def test_hybrid_cv_and_holdout_likelihood(pbn, oracle, node_type):
    n = 1200
    df, codes, cards = make_hybrid(n, seed=3)
    net = pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])
    nt = pbn.LinearGaussianCPDType() if node_type == "lg" else pbn.CKDEType()
    cv = pbn.CVLikelihood(df, 4, 7)
    ho = pbn.HoldoutLikelihood(df, 0.25, 2)
    for var, dpar, cpar in [("x", ["A"], []), ("y", ["B"], ["x"]), ("z", ["B", "A"], ["x", "y"])]:
        cont = df[[var] + cpar].to_numpy()
        dc, dk = [codes[d] for d in dpar], [cards[d] for d in dpar]
        fn = lambda tr, te: oracle.adaptator_fit_slogl(cont, dc, dk, tr, te, node_type)
        got = cv.local_score_node_type(net, nt, var, dpar + cpar)
        want = oracle.hybrid_cv(fn, n, 4, 7)
        assert close(got, want), (var, got, want)
        got = ho.local_score_node_type(net, nt, var, cpar + dpar)
        want = oracle.hybrid_holdout(fn, n, 0.25, 2)
        assert close(got, want), (var, got, want)


def test_discrete_factor_likelihoods(pbn, oracle):
    n = 900
    df, codes, cards = make_hybrid(n, seed=5)
    net = pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])
    cv = pbn.CVLikelihood(df, 5, 1)
    vl = pbn.ValidatedLikelihood(df, 0.2, 3, 4)
    for var, par in [("A", []), ("B", ["A"]), ("A", ["B"])]:
        pc, pk = [codes[p] for p in par], [cards[p] for p in par]
        fn = lambda tr, te: oracle.discrete_fit_slogl(codes[var], cards[var], pc, pk, tr, te)
        assert close(cv.local_score(net, var, par), oracle.hybrid_cv(fn, n, 5, 1))
        assert close(vl.vlocal_score(net, var, par), oracle.hybrid_holdout(fn, n, 0.2, 4))
        tr, _ = oracle.holdout_split(n, 0.2, 4)
        want = sum(fn(tr[a], tr[b]) for a, b in oracle.cv_folds(tr.size, 3, 4))
        assert close(vl.local_score(net, var, par), want)


def test_sparse_configurations_drop_factors(pbn, oracle):
    """Slices too small for a CKDE (SingularCovarianceData swallowed) or with constant data (variance < tol)
    contribute nothing (DiscreteAdaptator.hpp:339-344)."""
    n = 400
    rng = np.random.default_rng(0)
    A = np.zeros(n, dtype=np.int32)
    A[:3] = 1                      # configuration a1 has 3 rows only
    A[3:40] = 2                    # configuration a2: constant target
    x = rng.normal(size=n)
    y = 0.5 * x + rng.normal(scale=0.2, size=n)
    y[3:40] = 1.25
    df = pd.DataFrame({"A": pd.Categorical.from_codes(A, ["a0", "a1", "a2"]), "x": x, "y": y})
    net = pbn.SemiparametricBN(["A", "x", "y"], [], [("A", pbn.DiscreteFactorType())])
    ho = pbn.HoldoutLikelihood(df, 0.3, 0)
    cont = df[["y", "x"]].to_numpy()
    for node_type, nt in (("lg", pbn.LinearGaussianCPDType()), ("ckde", pbn.CKDEType())):
        fn = lambda tr, te: oracle.adaptator_fit_slogl(cont, [A], [3], tr, te, node_type)
        got = ho.local_score_node_type(net, nt, "y", ["x", "A"])
        assert close(got, oracle.hybrid_holdout(fn, n, 0.3, 0)), node_type


def test_hc_with_discrete_nodes_vs_oracle(pbn, oracle):
    from oracle import hc_oracle

    n = 1500
    df, codes, cards = make_hybrid(n, seed=9)
    names = list(df.columns)
    col = {c: i for i, c in enumerate(names)}
    disc = {"A", "B"}

    def score(v, t, ps):
        var = names[v]
        par = [names[p] for p in ps]
        if var in disc:
            return oracle.bic_discrete(codes[var], cards[var], [codes[p] for p in par], [cards[p] for p in par])
        dpar = [p for p in par if p in disc]
        cpar = [p for p in par if p not in disc]
        if not dpar:
            return oracle.bic_lg(df[[var] + cpar].to_numpy())
        return oracle.bic_clg(df[[var] + cpar].to_numpy(), [codes[d] for d in dpar], [cards[d] for d in dpar])

    types = [2 if c in disc else 0 for c in names]
    bl = [(j, i) for i in range(5) for j in range(5) if j > i]
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), pbn.BIC(df), pbn.CLGNetwork(names), arc_blacklist=[(names[a], names[b]) for a, b in bl])
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(5, 3, score, node_types=types, arc_blacklist=bl)
    got_trace = [({pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}[type(op)], col[op.source()], col[op.target()]) for op in hc.last.trace]
    assert got_trace == [t[:3] for t in o_trace]
    assert sorted((col[s], col[t]) for s, t in res.arcs()) == sorted(o_arcs)
    assert all(res.node_type(c) == pbn.DiscreteFactorType() for c in disc)
    assert not any(s not in disc and t in disc for s, t in res.arcs())
    assert hc.last.cells_scored == info["cells_scored"]


def test_hybrid_factor_classes_and_model_fit(pbn, oracle):
    """HCKDE / CLinearGaussianCPD / DiscreteFactor as stand-alone factors and BayesianNetwork.fit / logl / slogl
    (models/BayesianNetwork.hpp:960-994) on a hybrid table."""
    n = 1000
    df, codes, cards = make_hybrid(n, seed=11)
    train, test = df.iloc[:800], df.iloc[800:]
    tr, te = np.arange(800), np.arange(800, n)
    cont = df[["y", "x"]].to_numpy()
    for cls, node_type in ((pbn.CLinearGaussianCPD, "lg"), (pbn.HCKDE, "ckde")):
        f = cls("y", ["x", "B"])
        f.fit(train)
        want = oracle.adaptator_fit_slogl(cont, [codes["B"]], [3], tr, te, node_type)
        assert close(f.slogl(test), want)
        assert close(float(np.nansum(f.logl(test))), want)
    d = pbn.DiscreteFactor("B", ["A"])
    d.fit(train)
    want = oracle.discrete_fit_slogl(codes["B"], 3, [codes["A"]], [2], tr, te)
    assert close(d.slogl(test), want)
    net = pbn.SemiparametricBN(list(df.columns), [("A", "B"), ("A", "x"), ("x", "y"), ("B", "y"), ("y", "z")],
                               [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType()), ("z", pbn.CKDEType())])
    net.fit(train)
    total = net.slogl(test)
    parts = (oracle.discrete_fit_slogl(codes["A"], 2, [], [], tr, te)
             + oracle.discrete_fit_slogl(codes["B"], 3, [codes["A"]], [2], tr, te)
             + oracle.adaptator_fit_slogl(df[["x"]].to_numpy(), [codes["A"]], [2], tr, te, "lg")
             + oracle.adaptator_fit_slogl(df[["y", "x"]].to_numpy(), [codes["B"]], [3], tr, te, "lg")
             + oracle.adaptator_fit_slogl(df[["z", "y"]].to_numpy(), [], [], tr, te, "ckde"))
    assert close(total, parts)
    assert close(float(net.logl(test).sum()), total, rtol=1e-9)
    assert isinstance(net.cpd("y"), pbn.CLinearGaussianCPD) and isinstance(net.cpd("z"), pbn.CKDE)


def test_conditional_factor_by_assignment(pbn):
    """DiscreteAdaptator::conditional_factor (DiscreteAdaptator.hpp:350-356) keyed by an Assignment (factors/assignment.hpp)."""
    df = make_hybrid(4000, 3)[0]
    clg = pbn.CLinearGaussianCPD("y", ["x", "A", "B"])
    clg.fit(df)
    hck = pbn.HCKDE("z", ["x", "B"])
    hck.fit(df)
    for a, b in (("a0", "b0"), ("a1", "b2")):
        rows = df[(df["A"] == a) & (df["B"] == b)]
        f = clg.conditional_factor(pbn.Assignment({"A": a, "B": b}))
        want = pbn.LinearGaussianCPD("y", ["x"])
        want.fit(rows)
        assert np.allclose(f.beta, want.beta, rtol=1e-10) and np.isclose(f.variance, want.variance, rtol=1e-10)
    rows = df[df["B"] == "b1"]
    k = hck.conditional_factor(pbn.Assignment({"B": "b1", "unused": 1.5}))
    ref = pbn.CKDE("z", ["x"])
    ref.fit(rows)
    assert k.num_instances() == len(rows) and np.allclose(k.logl(rows.iloc[:50]), ref.logl(rows.iloc[:50]), rtol=1e-10)
    with pytest.raises(ValueError, match="not found"):
        clg.conditional_factor(pbn.Assignment({"A": "a0"}))
    with pytest.raises(ValueError, match="Category"):
        clg.conditional_factor(pbn.Assignment({"A": "zz", "B": "b0"}))
    asg = pbn.Assignment({"A": "a0", "B": "b1"})
    assert asg.size() == 2 and not asg.empty() and asg.value("B") == "b1" and asg == pbn.Assignment({"B": "b1", "A": "a0"})
    assert dict(iter(asg)) == {"A": "a0", "B": "b1"} and hash(asg) == hash(pbn.Assignment({"B": "b1", "A": "a0"}))


@pytest.mark.parametrize("dtype,kind", [("float64", "cv"), ("float32", "cv"), ("float64", "holdout")])
def test_hybrid_ckde_parts_add_up_to_the_local_score(pbn, dtype, kind):
    """pbn_score_batch_parts: the slices (configuration, fold) of a CKDE candidate with discrete parents fall into 64 fixed parts; the
    per-part sums of any split into n ranks, added over the ranks and then over the parts in order, are pbn_score_batch's value bit
    for bit (the ranks of a job share a candidate's slices)."""
    from pybnesian_amd import _lib

    rng = np.random.default_rng(31)
    n = 12000
    d1 = rng.integers(0, 3, size=n)
    d2 = (rng.random(n) < 0.3).astype(np.int64)
    x = rng.normal(size=n) + 0.7 * d1
    y = np.sin(x) * (1 + 0.4 * d2) + rng.normal(scale=0.5, size=n)
    z = 0.6 * y - 0.3 * x + rng.normal(scale=0.6, size=n)
    df = pd.DataFrame({"x": x, "y": y, "z": z}).astype(dtype)
    df["d1"] = pd.Categorical.from_codes(d1, ["a", "b", "c"])
    df["d2"] = pd.Categorical.from_codes(d2, ["p", "q"])
    make = (lambda: pbn.CVLikelihood(df, k=5, seed=3)) if kind == "cv" else (lambda: pbn.HoldoutLikelihood(df, test_ratio=0.25, seed=3))
    code = _lib.PBN_SCORE_CVLIK if kind == "cv" else _lib.PBN_SCORE_HOLDOUT
    net = pbn.SemiparametricBN(list(df.columns), [], [(v, pbn.CKDEType()) for v in "xyz"])
    cands = [("y", ["x", "d1"]), ("z", ["d2", "y", "d1"]), ("x", ["d1"]), ("z", ["y", "x", "d2"])]
    ref = make()
    want = [ref.local_score(net, v, p) for v, p in cands]
    var, ntype, off, par = ref._encode([(v, pbn.CKDEType(), p) for v, p in cands])
    for world in (1, 3, 8, 64):
        score = make()   # fresh caches: every rank's share is really evaluated
        total = np.zeros((len(cands), 64))
        for r in range(world):
            share = score._batch_parts(net, var, ntype, off, par, code, r, world)
            assert np.all((share == 0) | (total == 0))   # a part is non-zero on one rank only
            total += share
        got = []
        for row in total:
            acc = 0.0
            for v in row.tolist():
                acc += v
            got.append(acc)
        assert got == want, (world, got, want)
    with pytest.raises(ValueError, match="discrete parents"):
        ref._batch_parts(net, [0], [_lib.PBN_NODE_CKDE], [0, 1], [1], code, 0, 2)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_a_failing_candidate_does_not_poison_the_batch_engine(pbn, dtype):
    """The hybrid candidates of a pbn_score_batch call are in flight together (hybrid.hip: HybridBatch).  A call whose LAST candidate is
    invalid fails as a whole - after the earlier candidates' sweeps were enqueued - and must leave nothing behind: the same handle then
    scores the valid candidates to the bits a fresh handle gives."""
    from pybnesian_amd import _lib

    rng = np.random.default_rng(5)
    n = 9000
    d1 = rng.integers(0, 3, size=n)
    x = rng.normal(size=n) + 0.5 * d1
    y = np.cos(x) + 0.3 * d1 + rng.normal(scale=0.5, size=n)
    z = 0.5 * y + rng.normal(scale=0.7, size=n)
    df = pd.DataFrame({"x": x, "y": y, "z": z}).astype(dtype)
    df["d1"] = pd.Categorical.from_codes(d1, ["a", "b", "c"])
    net = pbn.SemiparametricBN(list(df.columns), [], [(v, pbn.CKDEType()) for v in "xyz"])
    good = [("y", pbn.CKDEType(), ["x", "d1"]), ("z", pbn.CKDEType(), ["y", "d1"]), ("x", pbn.CKDEType(), ["d1"])]
    fresh = pbn.CVLikelihood(df, k=4, seed=2)
    want = fresh._batch(net, good, fresh._kind).tolist()
    score = pbn.CVLikelihood(df, k=4, seed=2)
    var, ntype, off, par = score._encode(good)
    col = score._col
    # a discrete column asked for as a CKDE node: rejected by the engine after the three good candidates were enqueued
    bad = (var + [col["d1"]], ntype + [_lib.PBN_NODE_CKDE], off + [off[-1] + 1], par + [col["x"]])
    with pytest.raises(ValueError, match="discrete column"):
        score._batch_raw(net, *bad, score._kind)
    assert score._batch(net, good, score._kind).tolist() == want
    assert [score.local_score_node_type(net, t, v, p) for v, t, p in good] == want


@pytest.mark.parametrize("node_type", ["lg", "ckde"])
def test_hybrid_candidate_with_20_continuous_parents(pbn, oracle, node_type):
    """A child of one discrete and 20 continuous parents (DiscreteAdaptator slices over 21 continuous columns): the per-cell Gram
    of more than 8 columns and, for CKDE, the 17-32-dimension sweeps - the reference has no limit here
    (factors/discrete/DiscreteAdaptator.hpp:201-348); the library's was 16 continuous parents until round 4."""
    rng = np.random.default_rng(2)
    n, p = 2400, 20
    A = rng.integers(0, 2, size=n)
    ev = rng.normal(size=(n, p)) @ (np.eye(p) + np.tril(rng.uniform(-0.3, 0.3, (p, p)), -1)).T + 0.8 * A[:, None]
    y = 0.2 * ev.sum(axis=1) + np.where(A == 0, -1.0, 1.5) + rng.normal(scale=0.6, size=n)
    names = ["y"] + [f"e{i}" for i in range(p)]
    df = pd.DataFrame(np.column_stack([y, ev]), columns=names)
    df["A"] = pd.Categorical.from_codes(A, ["a0", "a1"])
    net = pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType())])
    nt = pbn.LinearGaussianCPDType() if node_type == "lg" else pbn.CKDEType()
    ho = pbn.HoldoutLikelihood(df, 0.25, 3)
    tr, te = oracle.holdout_split(n, 0.25, 3)
    want = oracle.adaptator_fit_slogl(df[names].to_numpy(), [A.astype(np.int32)], [2], tr, te, node_type)
    got = ho.local_score_node_type(net, nt, "y", names[1:] + ["A"])
    assert close(got, want), (got, want)
