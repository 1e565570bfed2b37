"""CPU tier: the C++ hill-climbing host logic (pbn_hc_estimate, batched score requests) against the
pure-Python serial restatement of the reference (oracle/hc_oracle.py), driven by the SAME deterministic
score function — so every decision (operator sequence, deltas, final arcs and node types, number of cells
scored) must be bit-identical, including exact ties (scores are rounded to create them) resolved by
libstdc++'s unstable std::sort on the persistent index vector."""
import itertools

import numpy as np
import pytest

from oracle import hc_oracle

LG, CKDE = 0, 1


class TableScore:
    """Deterministic synthetic decomposable score over node indices; ties on purpose (rounding)."""

    validated = False
    _kind = 0

    def __init__(self, n, seed, validated=False, decimals=2):
        rng = np.random.default_rng(seed)
        self.n = n
        self.base = rng.normal(size=(n, 3)) * 3
        self.w = rng.normal(size=(n, n, 3)) * 2
        self.pair = rng.normal(size=(n, n, n)) * 0.7
        self.vnoise = rng.normal(size=(n, n, 3)) * 0.8
        self.decimals = decimals
        self.validated = validated
        self._col = {f"n{i}": i for i in range(n)}
        self.calls = 0

    def compatible_bn(self, model):
        return True

    def raw(self, var, ntype, parents, validated=False):
        self.calls += 1
        ps = sorted(parents)
        s = self.base[var, ntype] + sum(self.w[var, p, ntype] for p in ps)
        s += sum(self.pair[var, a, b] for a, b in itertools.combinations(ps, 2))
        s -= 1.1 * len(ps) ** 2
        if validated:
            s += sum(self.vnoise[var, p, ntype] for p in ps) - 0.3 * len(ps)
        return float(np.round(s, self.decimals))

    def _batch_raw(self, model, var, ntype, off, par, kind):
        validated = kind == 3
        return [self.raw(var[i], ntype[i], par[off[i]: off[i + 1]], validated) for i in range(len(var))]


def run_product(ts, bn_type_name, n, node_types=None, **kw):
    import pybnesian_amd as pbn

    names = [f"n{i}" for i in range(n)]
    tcode = {LG: pbn.LinearGaussianCPDType(), CKDE: pbn.CKDEType()}
    if bn_type_name == "gaussian":
        start = pbn.GaussianNetwork(names)
    elif bn_type_name == "kde":
        start = pbn.KDENetwork(names)
    else:
        start = pbn.SemiparametricBN(names, [], [(names[i], tcode[t]) for i, t in enumerate(node_types or [LG] * n)])
    ops = []
    if kw.pop("op_arcs", True):
        ops.append(pbn.ArcOperatorSet())
    if kw.pop("op_types", False):
        ops.append(pbn.ChangeNodeTypeSet())
    if not kw.pop("arcs_first", True):
        ops.reverse()
    opset = ops[0] if len(ops) == 1 else pbn.OperatorPool(ops)
    to_names = lambda prs: [(names[a], names[b]) for a, b in prs]
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(opset, ts, start,
                      arc_blacklist=to_names(kw.pop("arc_blacklist", [])), arc_whitelist=to_names(kw.pop("arc_whitelist", [])),
                      type_blacklist=[(names[a], tcode[t]) for a, t in kw.pop("type_blacklist", [])],
                      type_whitelist=[(names[a], tcode[t]) for a, t in kw.pop("type_whitelist", [])], **kw)
    idx = {nm: i for i, nm in enumerate(names)}
    arcs = [(idx[s], idx[t]) for s, t in res.arcs()]
    types = [LG if res.node_type(nm) == pbn.LinearGaussianCPDType() else CKDE for nm in names]
    trace = []
    for op in hc.last.trace:
        if isinstance(op, pbn.ChangeNodeType):
            trace.append((3, idx[op.node()], LG if op.node_type() == pbn.LinearGaussianCPDType() else CKDE, op.delta()))
        else:
            kind = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}[type(op)]
            trace.append((kind, idx[op.source()], idx[op.target()], op.delta()))
    return arcs, types, trace, hc.last


BN_CODE = {"gaussian": 0, "spbn": 1, "kde": 2}


def check(n, seed, bn="gaussian", validated=False, node_types=None, **kw):
    ts = TableScore(n, seed, validated)
    p_arcs, p_types, p_trace, last = run_product(ts, bn, n, node_types, **dict(kw))
    okw = dict(kw)
    score = lambda v, t, ps: ts.raw(v, t, ps, False)
    vscore = (lambda v, t, ps: ts.raw(v, t, ps, True)) if validated else None
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(n, BN_CODE[bn], score, vscore, node_types=node_types, **okw)
    assert p_trace == o_trace  # operator sequence AND deltas, bit for bit
    assert sorted(p_arcs) == sorted(o_arcs)
    assert p_types == o_types
    assert last.iterations == info["iterations"]
    assert last.cells_scored == info["cells_scored"]
    return p_trace


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("n", [4, 9, 17])
def test_hc_gaussian_matches_reference_restatement(ensure_built, n, seed):
    trace = check(n, seed)
    assert n == 4 or len(trace) > 0


@pytest.mark.parametrize("seed", range(4))
def test_hc_with_restrictions_and_indegree(ensure_built, seed):
    n = 10
    rng = np.random.default_rng(seed)
    pairs = [(a, b) for a in range(n) for b in range(n) if a != b]
    rng.shuffle(pairs)
    bl = [tuple(map(int, p)) for p in pairs[:12]]
    wl = [(0, 1), (2, 3)]
    bl = [p for p in bl if p not in wl and (p[1], p[0]) not in wl]
    check(n, seed, arc_blacklist=bl, arc_whitelist=wl, max_indegree=2)
    check(n, seed + 100, max_indegree=1, epsilon=0.5)
    check(n, seed + 200, max_iters=3)


@pytest.mark.parametrize("seed", range(5))
def test_hc_semiparametric_pool(ensure_built, seed):
    n = 8
    types = [int(x) for x in np.random.default_rng(seed).integers(0, 2, size=n)]
    check(n, seed, bn="spbn", node_types=types, op_types=True)
    check(n, seed, bn="spbn", node_types=types, op_types=True, arcs_first=False)
    check(n, seed, bn="spbn", node_types=types, op_types=True, type_blacklist=[(1, 1 - types[1])], type_whitelist=[(2, CKDE)])
    check(n, seed, bn="kde")


@pytest.mark.parametrize("seed", range(6))
@pytest.mark.parametrize("patience", [0, 1, 3])
def test_hc_validated_patience_tabu(ensure_built, seed, patience):
    check(9, seed, bn="spbn", validated=True, op_types=True, patience=patience)
    check(7, seed + 50, bn="gaussian", validated=True, patience=patience)


def test_hc_errors(ensure_built):
    import pybnesian_amd as pbn

    ts = TableScore(4, 0)
    start = pbn.GaussianNetwork(["n0", "n1", "n2", "n3"])
    with pytest.raises(ValueError, match="non-homogeneous"):
        pbn.GreedyHillClimbing().estimate(pbn.ChangeNodeTypeSet(), ts, start)
    with pytest.raises(ValueError):
        pbn.OperatorPool([])
    # epsilon above every delta returns the start model (hillclimbing_test.py:30-36)
    res = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), ts, start, epsilon=1e9)
    assert res.num_arcs() == 0


@pytest.mark.parametrize("seed", range(4))
def test_hc_with_discrete_nodes_type_rule(ensure_built, seed):
    """No continuous -> discrete arcs (SemiparametricBN.hpp:93-98); discrete nodes have no alternative type."""
    n = 8
    types = [2, 2, 0, 1, 0, 2, 1, 0]

    class Disc(TableScore):
        def is_discrete(self, v):
            return types[int(v[1:])] == 2

    ts = Disc(n, seed)
    import pybnesian_amd as pbn

    names = [f"n{i}" for i in range(n)]
    tcode = {0: pbn.LinearGaussianCPDType(), 1: pbn.CKDEType(), 2: pbn.DiscreteFactorType()}
    start = pbn.SemiparametricBN(names, [], [(names[i], tcode[t]) for i, t in enumerate(types)])
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), ts, start)
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(n, 1, lambda v, t, ps: ts.raw(v, t, ps), node_types=types, op_types=True)
    idx = {nm: i for i, nm in enumerate(names)}
    assert sorted((idx[s], idx[t]) for s, t in res.arcs()) == sorted(o_arcs)
    assert [{pbn.LinearGaussianCPDType(): 0, pbn.CKDEType(): 1, pbn.DiscreteFactorType(): 2}[res.node_type(nm)] for nm in names] == o_types
    assert hc.last.cells_scored == info["cells_scored"] and hc.last.iterations == info["iterations"]
    for s, t in res.arcs():
        assert not (types[idx[t]] == 2 and types[idx[s]] != 2)
    assert [o_types[i] for i in range(n) if types[i] == 2] == [2] * types.count(2)


def test_operator_set_piecewise_api_reproduces_estimate(ensure_built):
    """OperatorSet.cache_scores / find_max / find_max_tabu / update_scores + LocalScoreCache + OperatorTabuSet
    (operators.hpp:258-355) driven from Python reproduce GreedyHillClimbing.estimate step for step."""
    import pybnesian_amd as pbn

    n, seed = 9, 2
    names = [f"n{i}" for i in range(n)]
    ts = TableScore(n, seed)
    hc = pbn.GreedyHillClimbing()
    ref = hc.estimate(pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), ts, pbn.SemiparametricBN(names))
    ref_trace = list(hc.last.trace)

    pool = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
    model = pbn.SemiparametricBN(names)
    with pytest.raises(ValueError, match="Local cache not initialized"):
        pool.find_max(model)
    pool.cache_scores(model, ts)
    cache = pool.local_score_cache()
    for v in names:
        assert cache.local_score(model, v) == ts.raw(int(v[1:]), 0, [])
    darcs, dtypes = pool.delta()
    i, j = 2, 5  # delta(source, target) of an addition = S(t | s) - S(t)
    assert darcs[i, j] == ts.raw(j, 0, [i]) - ts.raw(j, 0, [])
    assert dtypes[3] == ts.raw(3, 1, []) - ts.raw(3, 0, [])
    trace = []
    while True:
        op = pool.find_max(model)
        if op is None or op.delta() < 1.4901161193847656e-08:
            break
        op.apply(model)
        trace.append(op)
        pool.update_scores(model, ts, op.nodes_changed(model))
    assert trace == ref_trace and [o.delta() for o in trace] == [o.delta() for o in ref_trace]
    assert sorted(model.arcs()) == sorted(ref.arcs())
    assert cache.sum() == pytest.approx(sum(ts.raw(int(v[1:]), 1 if model.node_type(v) == pbn.CKDEType() else 0,   # unknown -> LinearGaussian
                                                   [int(p[1:]) for p in model.parents(v)]) for v in names))
    # tabu: the best operator is skipped when it is in the tabu set
    pool.cache_scores(pbn.SemiparametricBN(names), ts)
    fresh = pbn.SemiparametricBN(names)
    best = pool.find_max(fresh)
    tabu = pbn.OperatorTabuSet()
    assert tabu.empty() and not tabu.contains(best)
    tabu.insert(best)
    second = pool.find_max_tabu(fresh, tabu)
    assert tabu.contains(best) and second != best and second.delta() <= best.delta()
    assert best.opposite(fresh) == (pbn.RemoveArc(best.source(), best.target(), 0) if isinstance(best, pbn.AddArc) else best.opposite(fresh))
    pool.finished()
    with pytest.raises(ValueError):
        pool.find_max(fresh)


# ---- conditional networks: interface nodes (operators.cpp:134-256,365-437; operators.hpp:526-578) ---------------------------
@pytest.mark.parametrize("seed", range(5))
@pytest.mark.parametrize("n,ni", [(5, 2), (8, 4), (6, 1)])
def test_conditional_hc_matches_reference_restatement(ensure_built, n, ni, seed):
    import pybnesian_amd as pbn

    J = n + ni
    names = [f"n{i}" for i in range(J)]
    tcode = {LG: pbn.LinearGaussianCPDType(), CKDE: pbn.CKDEType()}
    rng = np.random.default_rng(100 + seed)
    for bn, validated, op_types, patience in (("gaussian", False, False, 0), ("spbn", True, True, 2)):
        ts = TableScore(J, seed, validated=validated)
        types = [int(t) for t in rng.integers(0, 2, size=J)] if bn == "spbn" else None
        bl = [(int(a), int(b)) for a, b in zip(rng.integers(0, J, size=4), rng.integers(0, n, size=4)) if a != b]
        wl = [(n, 0)] if seed % 2 == 0 else []          # whitelisted interface -> node arc
        bl = [p for p in bl if p not in wl]
        if bn == "gaussian":
            start = pbn.ConditionalGaussianNetwork(names[:n], names[n:])
        else:
            start = pbn.ConditionalSemiparametricBN(names[:n], names[n:], [], [(names[i], tcode[t]) for i, t in enumerate(types)])
        ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]) if op_types else pbn.ArcOperatorSet()
        hc = pbn.GreedyHillClimbing()
        res = hc.estimate(ops, ts, start, arc_blacklist=[(names[a], names[b]) for a, b in bl],
                          arc_whitelist=[(names[a], names[b]) for a, b in wl], max_indegree=3, patience=patience)
        score = lambda v, t, ps: ts.raw(v, t, ps, False)
        vscore = (lambda v, t, ps: ts.raw(v, t, ps, True)) if validated else None
        o_arcs, o_types, o_trace, info = hc_oracle.estimate(n, BN_CODE[bn], score, vscore, node_types=types, arc_blacklist=bl,
                                                            arc_whitelist=wl, op_types=op_types, max_indegree=3, patience=patience,
                                                            n_interface=ni)
        idx = {nm: i for i, nm in enumerate(names)}
        trace = []
        for op in hc.last.trace:
            if isinstance(op, pbn.ChangeNodeType):
                trace.append((3, idx[op.node()], LG if op.node_type() == pbn.LinearGaussianCPDType() else CKDE, op.delta()))
            else:
                trace.append(({pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}[type(op)], idx[op.source()], idx[op.target()], op.delta()))
        assert trace == o_trace
        assert sorted((idx[s], idx[t]) for s, t in res.arcs()) == sorted(o_arcs)
        assert hc.last.cells_scored == info["cells_scored"] and hc.last.iterations == info["iterations"]
        assert res.interface_nodes() == names[n:] and res.nodes() == names[:n]
        assert all(idx[t] < n for _, t in res.arcs())                       # nothing points into an interface node
        if bn == "spbn":
            assert [LG if res.node_type(nm) == pbn.LinearGaussianCPDType() else CKDE for nm in names[:n]] == o_types[:n]


def test_conditional_network_surface(ensure_built):
    import pybnesian_amd as pbn

    m = pbn.ConditionalGaussianNetwork(["a", "b"], ["x", "y"], [("x", "a"), ("a", "b")])
    assert m.nodes() == ["a", "b"] and m.interface_nodes() == ["x", "y"] and m.joint_nodes() == ["a", "b", "x", "y"]
    assert m.num_nodes() == 2 and m.num_interface_nodes() == 2 and m.num_joint_nodes() == 4
    assert m.is_interface("x") and not m.is_interface("a") and m.contains_joint_node("y") and not m.contains_node("y")
    assert m.can_add_arc("y", "b") and not m.can_add_arc("a", "x") and not m.can_flip_arc("x", "a")
    with pytest.raises(ValueError, match="cannot have parents"):
        m.add_arc("a", "x")
    u = m.unconditional_bn()
    assert u.nodes() == ["a", "b", "x", "y"] and sorted(u.arcs()) == [("a", "b"), ("x", "a")] and not u.interface_nodes()
    c = u.conditional_bn(["a", "b", "y"], ["x"])
    assert c.interface_nodes() == ["x"] and sorted(c.arcs()) == [("a", "b"), ("x", "a")]
    assert m.topological_sort() == ["a", "b"]
    c2 = m.clone()
    assert c2.interface_nodes() == ["x", "y"] and c2.arcs() == m.arcs()
    m.add_cpds([pbn.LinearGaussianCPD("a", ["x"], [1.0, 2.0], 0.01), pbn.LinearGaussianCPD("b", ["a"], [0.0, -1.0], 0.01)])
    import pandas as pd

    ev = pd.DataFrame({"x": np.linspace(-1, 1, 50), "y": np.zeros(50)})
    s = m.sample(ev, 3, concat_evidence=True, ordered=True)
    assert s.schema.names == ["a", "b", "x", "y"]
    a, b = s.column(0).to_numpy(), s.column(1).to_numpy()
    assert np.allclose(a, 1.0 + 2.0 * ev["x"], atol=0.5) and np.allclose(b, -a, atol=0.5)


def test_cross_validation_and_holdout_objects(ensure_built):
    """dataset::CrossValidation / HoldOut (tests/dataset/crossvalidation_test.py, holdout_test.py of the reference):
    fold sizes, disjointness, determinism in the seed, null handling, and the libstdc++ shuffle known answer."""
    import pandas as pd

    import pybnesian_amd as pbn
    from oracle import oracle

    n = 103
    df = pd.DataFrame({"a": np.arange(n, dtype=float), "b": np.arange(n, dtype=float) * 2})
    cv = pbn.CrossValidation(df, 10, 0)
    folds = list(cv.indices())
    assert [len(te) for _, te in folds] == [11, 11, 11] + [10] * 7               # first n % k folds one longer
    assert sorted(np.concatenate([te for _, te in folds]).tolist()) == list(range(n))
    perm = oracle.shuffled_indices(n, 0)
    assert np.array_equal(np.concatenate([te for _, te in folds]), perm)         # folds are slices of the shuffled order
    for tr, te in folds:
        assert len(tr) + len(te) == n and not set(tr) & set(te)
    tr0, te0 = cv.fold(0)
    assert te0.num_rows == 11 and np.array_equal(te0.column(0).to_numpy(), perm[:11].astype(float))
    assert [te.num_rows for _, te in pbn.CrossValidation(df, 10, 0)] == [len(te) for _, te in folds]
    assert not np.array_equal(list(pbn.CrossValidation(df, 10, 1).indices())[0][1], folds[0][1])
    assert cv.loc(["b"]).fold(1)[1].schema.names == ["b"]
    with pytest.raises(ValueError, match="Cannot split"):
        pbn.CrossValidation(df, 200, 0)
    dn = df.copy()
    dn.loc[[3, 50], "a"] = np.nan
    cvn = pbn.CrossValidation(dn, 5, 0)
    assert sorted(np.concatenate([te for _, te in cvn.indices()]).tolist()) == [i for i in range(n) if i not in (3, 50)]
    assert sum(te.num_rows for _, te in pbn.CrossValidation(dn, 5, 0, include_null=True)) == n
    ho = pbn.HoldOut(df, 0.2, 0)
    assert ho.test_data().num_rows == round(n * 0.2) and ho.training_data().num_rows == n - round(n * 0.2)
    both = np.concatenate([ho.training_data().column(0).to_numpy(), ho.test_data().column(0).to_numpy()])
    assert np.array_equal(both, perm.astype(float))                              # train = first n - test of the shuffle
    with pytest.raises(ValueError, match="test_ratio"):
        pbn.HoldOut(df, 1.5, 0)


def test_oracle_follow_mode_accounts_for_ties():
    """hc_oracle.estimate(follow=...) - the tie-flip accounting used by tests/test_tieflip_gpu.py: replaying the restatement's
    own trace records no flips; a trace whose first operator is the reversed (exactly tied) arc is accepted and recorded with
    gap 0; a trace that starts with a clearly worse operator is rejected."""
    from oracle import hc_oracle

    n = 5
    rng = np.random.default_rng(3)
    base = rng.normal(size=(n, n))
    sym = base + base.T                      # score-equivalent toy score: s(v | {p}) - s(v | {}) symmetric in (v, p)

    def score(v, t, ps):
        return float(sum(sym[v, p] for p in ps)) - 0.3 * len(ps) ** 2

    arcs, _, trace, info = hc_oracle.estimate(n, 0, score)
    assert len(trace) >= 2 and info["flips"] == []
    _, _, _, same = hc_oracle.estimate(n, 0, score, follow=[t[:3] for t in trace])
    assert same["flips"] == []
    k, a, b, d = trace[0]
    flipped = [(k, b, a)] + [t[:3] for t in trace[1:]]
    try:
        r_arcs, _, _, rinfo = hc_oracle.estimate(n, 0, score, follow=flipped)
        assert rinfo["flips"] and rinfo["flips"][0]["iteration"] == 1 and rinfo["flips"][0]["gap"] <= 1e-12
    except AssertionError as ex:             # later operators of the original trace may not be greedy any more on the flipped graph
        assert "iteration 1:" not in str(ex)
    worst = min(((score(t, 0, [s]) - score(t, 0, []), s, t) for s in range(n) for t in range(n) if s != t))
    with pytest.raises(AssertionError, match="not a tie"):
        hc_oracle.estimate(n, 0, score, follow=[(0, worst[1], worst[2])])
