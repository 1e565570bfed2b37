"""GPU tier: BASELINE config 5 in miniature - hybrid table in **fp32**, ValidatedLikelihood local scores of HCKDE and
CLinearGaussianCPD candidates (DiscreteAdaptator slices, a8 of SURVEY.md §8a) against the oracle's per-slice restatement at
the north star's fp32 tolerance (1e-3 relative), including slices large enough (>= 32 768 training rows) for the pruned
f16x2 sweeps, and MMHC end to end (hybrid MutualInformation -> CPCs -> restricted hill-climb) against
`mmpc_oracle` + `hc_oracle` driven by oracle scores.

Reference: factors/discrete/DiscreteAdaptator.hpp:201-348, learning/scores/validated_likelihood.hpp:12-75,
learning/scores/cv_likelihood.cpp:5-25, learning/scores/holdout_likelihood.cpp:8-23, learning/algorithms/mmhc.cpp:24-88.
No reference test covers hybrid scoring (SURVEY.md §8c "parity unpinned"): the oracle restatement is the only check."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu

RTOL_F32 = 1e-3     # BASELINE.json: slogl within 1e-3 relative in fp32


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def oracle():
    from oracle import oracle as o

    return o


def make_c5(n, seed=0, dtype=np.float32):
    """C5's shape in small: discrete A (2), B (3); continuous x | A, y | x, B, z | x, y (non-linear), w | z."""
    rng = np.random.default_rng(seed)
    A = rng.integers(0, 2, size=n)
    B = (rng.random(n) < np.where(A == 0, 0.3, 0.6)).astype(int) + (rng.random(n) < 0.2)
    x = rng.normal(loc=np.where(A == 0, -1.0, 2.0), scale=1.0)
    y = 0.7 * x + np.array([0.0, 3.0, -2.0])[B] + rng.normal(scale=0.5, size=n)
    z = np.tanh(x) * 2.0 - 0.4 * y + rng.normal(scale=0.3, size=n)
    w = 0.5 * z + rng.normal(scale=0.7, size=n)
    df = pd.DataFrame({"x": x, "y": y, "z": z, "w": w}).astype(dtype)
    df["A"] = pd.Categorical.from_codes(A, ["a0", "a1"])
    df["B"] = pd.Categorical.from_codes(B, ["b0", "b1", "b2"])
    return df, {"A": A.astype(np.int32), "B": B.astype(np.int32)}, {"A": 2, "B": 3}


def close(a, b, rtol):
    return abs(a - b) <= rtol * max(abs(b), 1e-12)


def spbn(pbn, df):
    return pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])


def validated_oracle(oracle, cont, dcodes, dcards, node_type, n, ratio, k, seed):
    """(local_score, vlocal_score) of ValidatedLikelihood(ratio, k, seed): CV over the hold-out training part with the
    same seed (validated_likelihood.hpp:19-20), hold-out likelihood."""
    fn = lambda tr, te: oracle.adaptator_fit_slogl(cont, dcodes, dcards, tr, te, node_type)
    tr, te = oracle.holdout_split(n, ratio, seed)
    local = sum(fn(tr[a], tr[b]) for a, b in oracle.cv_folds(tr.size, k, seed))
    return local, fn(tr, te)


CASES = [("x", ["A"], []), ("y", ["B"], ["x"]), ("z", ["B", "A"], ["x", "y"]), ("w", [], ["z"]), ("z", ["A"], ["x", "y", "w"])]


@pytest.mark.parametrize("node_type", ["ckde", "lg"])
@pytest.mark.parametrize("prune_rows", [None, 1024])
def test_fp32_hybrid_validated_scores(pbn, oracle, node_type, prune_rows, monkeypatch):
    """HCKDE / CLG slices in fp32 through ValidatedLikelihood.  prune_rows = 1024 lowers the pruning threshold so that the
    Morton-ordered, tile-pruned f16x2 sweeps run on every slice of this small table (default: 32 768 rows)."""
    if prune_rows is not None:
        monkeypatch.setenv("PBN_PRUNE_MIN_ROWS", str(prune_rows))
    n = 6000
    df, codes, cards = make_c5(n, seed=3)
    net = spbn(pbn, df)
    nt = pbn.CKDEType() if node_type == "ckde" else pbn.LinearGaussianCPDType()
    vl = pbn.ValidatedLikelihood(df, 0.2, 3, 4)
    for var, dpar, cpar in CASES:
        cont = df[[var] + cpar].to_numpy()
        assert cont.dtype == np.float32
        dc, dk = [codes[d] for d in dpar], [cards[d] for d in dpar]
        want_l, want_v = validated_oracle(oracle, cont, dc, dk, node_type, n, 0.2, 3, 4)
        got_l = vl.local_score_node_type(net, nt, var, dpar + cpar)
        got_v = vl.vlocal_score_node_type(net, nt, var, cpar + dpar)
        assert close(got_l, want_l, RTOL_F32), (var, dpar, cpar, got_l, want_l)
        assert close(got_v, want_v, RTOL_F32), (var, dpar, cpar, got_v, want_v)


def test_fp32_hybrid_scores_vs_the_float_arithmetic_of_the_reference(pbn, oracle):
    """The same fp32 scores against the oracle in the reference's OWN float arithmetic (cov, bandwidth rule input, distances, logsumexp
    and logl in float: NormalReferenceRule.hpp:124-133, KDE.hpp:466-470, KDE.cl.src with float) at the reference tests' own fp32
    tolerance - atol 5e-4 per logl (tests/factors/continuous/KDE_test.py:181-182), i.e. 5e-4 x the rows a score sums over - and, beside
    it, against fp64 arithmetic on the same float values (the truth for those values): the device sits between the two."""
    n = 6000
    df, codes, cards = make_c5(n, seed=3)
    net = spbn(pbn, df)
    vl = pbn.ValidatedLikelihood(df, 0.2, 3, 4)
    tr, te = oracle.holdout_split(n, 0.2, 4)
    worst32 = worst64 = 0.0
    for var, dpar, cpar in CASES:
        cont = df[[var] + cpar].to_numpy()
        assert cont.dtype == np.float32
        dc, dk = [codes[d] for d in dpar], [cards[d] for d in dpar]
        got_v = vl.vlocal_score_node_type(net, pbn.CKDEType(), var, cpar + dpar)
        got_l = vl.local_score_node_type(net, pbn.CKDEType(), var, dpar + cpar)
        for arith in ("float32", "float64"):
            fn = lambda a, b: oracle.adaptator_fit_slogl(cont, dc, dk, a, b, "ckde", arithmetic=arith)
            want_v = fn(tr, te)
            want_l = sum(fn(tr[a], tr[b]) for a, b in oracle.cv_folds(tr.size, 3, 4))
            per_v, per_l = abs(got_v - want_v) / te.size, abs(got_l - want_l) / tr.size
            assert per_v <= 5e-4 and per_l <= 5e-4, (var, dpar, cpar, arith, per_v, per_l)
            if arith == "float32":
                worst32 = max(worst32, per_v, per_l)
            else:
                worst64 = max(worst64, per_v, per_l)
    # fp64 arithmetic on the float values is what the device's bf16x3 / fp32-accumulated sweeps approximate: closer to it than the
    # reference's float arithmetic is
    assert worst64 <= 5e-5, (worst32, worst64)


def test_fp32_hybrid_large_slices_pruned(pbn, oracle):
    """Slices of >= 32 768 training rows: the default pruned f16x2 path of C5's per-configuration sweeps (hold-out
    likelihood: 96 000 training rows over 2 configurations, 24 000 test rows), and a no-discrete-parent CKDE whose 2-fold CV
    trains on 48 000 rows."""
    n = 120_000
    df, codes, cards = make_c5(n, seed=5)
    net = spbn(pbn, df)
    vl = pbn.ValidatedLikelihood(df, 0.2, 2, 1)
    tr, te = oracle.holdout_split(n, 0.2, 1)
    for var, dpar, cpar in [("y", ["A"], ["x"]), ("z", ["A"], ["x", "y"])]:
        cont = df[[var] + cpar].to_numpy()
        dc, dk = [codes[d] for d in dpar], [cards[d] for d in dpar]
        sizes = np.bincount(codes["A"][tr])
        assert sizes.min() >= 32768
        want = oracle.adaptator_fit_slogl(cont, dc, dk, tr, te, "ckde")
        got = vl.vlocal_score_node_type(net, pbn.CKDEType(), var, dpar + cpar)
        assert close(got, want, RTOL_F32), (var, got, want)
    cont = df[["w", "z"]].to_numpy()
    want = sum(oracle.adaptator_fit_slogl(cont, [], [], tr[a], tr[b], "ckde") for a, b in oracle.cv_folds(tr.size, 2, 1))
    got = vl.local_score_node_type(net, pbn.CKDEType(), "w", ["z"])
    assert close(got, want, RTOL_F32), (got, want)


def test_fp32_pruned_equals_unpruned(pbn, monkeypatch):
    """The tile-pruned fp32 sweeps drop at most 1.5e-5 of a sum (margin 36 at 10^6 rows - the size of the fp32 Gram form's own
    error): same scores as the unpruned sweeps far inside the fp32 bar of 1e-3."""
    n = 40_000
    df, _, _ = make_c5(n, seed=8)
    net = spbn(pbn, df)
    got = {}
    for prune in ("0", "1"):
        monkeypatch.setenv("PBN_SWEEP_PRUNE", prune)
        monkeypatch.setenv("PBN_PRUNE_MIN_ROWS", "2048")
        vl = pbn.ValidatedLikelihood(df, 0.2, 2, 0)
        got[prune] = [vl.local_score_node_type(net, pbn.CKDEType(), v, p) for v, p in (("y", ["x", "B"]), ("z", ["x", "y", "A"]), ("w", ["z"]))]
    assert np.allclose(got["0"], got["1"], rtol=3e-5), got     # fp32 margin 36 (round 4): the bound on the dropped mass is 1.5e-5 of a sum


def test_mmhc_hybrid_end_to_end_vs_oracles(pbn, oracle):
    """C5's pipeline on a small fp64 table so that every decision is comparable: hybrid MutualInformation p-values -> MMPC
    CPCs (vs mmpc_oracle over mi_oracle p-values) -> arc blacklist -> ValidatedLikelihood hill-climb over arcs and node types
    (vs hc_oracle driven by oracle scores): same CPCs, same operator trace, same structure and node types."""
    from oracle import hc_oracle, mmpc_oracle
    from oracle.mi_oracle import MIOracle
    from pybnesian_amd.independences import mmpc_cpcs

    n = 2500
    df, codes, cards = make_c5(n, seed=13, dtype=np.float64)
    names = list(df.columns)              # x y z w A B
    col = {c: i for i, c in enumerate(names)}
    disc = {"A", "B"}
    cols = {c: ((codes[c].astype(np.int64), cards[c]) if c in disc else df[c].to_numpy()) for c in names}
    mi = MIOracle(cols)
    test = pbn.MutualInformation(df)
    got_cpcs, ntests = mmpc_cpcs(test, names, 0.05)
    want_cpcs, calls = mmpc_oracle.mmpc_all_variables(lambda a, b, c: mi.pvalue(names[a], names[b], [names[i] for i in c]), len(names), 0.05)
    assert [[col[v] for v in c] for c in got_cpcs] == want_cpcs and ntests == calls

    ratio, k, seed = 0.2, 3, 0
    tr, te = oracle.holdout_split(n, ratio, seed)
    folds = oracle.cv_folds(tr.size, k, seed)
    LG, CKDE, DISC = 0, 1, 2

    def unit(v, t, ps, train, test_rows):
        var = names[v]
        par = [names[p] for p in ps]
        if var in disc:
            return oracle.discrete_fit_slogl(codes[var], cards[var], [codes[p] for p in par], [cards[p] for p in par], train, test_rows)
        dpar = [p for p in par if p in disc]
        cpar = [p for p in par if p not in disc]
        return oracle.adaptator_fit_slogl(df[[var] + cpar].to_numpy(), [codes[d] for d in dpar], [cards[d] for d in dpar], train,
                                          test_rows, "ckde" if t == CKDE else "lg")

    score = lambda v, t, ps: sum(unit(v, t, ps, tr[a], tr[b]) for a, b in folds)
    vscore = lambda v, t, ps: unit(v, t, ps, tr, te)
    types = [DISC if c in disc else LG for c in names]
    bl = [(i, j) for i in range(len(names)) for j in range(len(names)) if i != j and j not in want_cpcs[i]]
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(len(names), 1, score, vscore=vscore, node_types=types, arc_blacklist=bl,
                                                        op_types=True, max_indegree=3, patience=0)
    vl = pbn.ValidatedLikelihood(df, ratio, k, seed)
    mm = pbn.MMHC()
    res = mm.estimate(test, pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), vl, bn_type=pbn.SemiparametricBNType(),
                      alpha=0.05, max_indegree=3)
    assert [[col[v] for v in c] for c in mm.last_cpcs] == want_cpcs
    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    tcode = {pbn.LinearGaussianCPDType(): LG, pbn.CKDEType(): CKDE, pbn.DiscreteFactorType(): DISC}
    got_trace = [(3, col[op.node()], tcode[op.node_type()]) if isinstance(op, pbn.ChangeNodeType) else (kinds[type(op)], col[op.source()], col[op.target()])
                 for op in mm.hc.last.trace]
    assert got_trace == [t[:3] for t in o_trace], (got_trace, o_trace)
    assert sorted((col[s], col[t]) for s, t in res.arcs()) == sorted(o_arcs), (res.arcs(), o_arcs)
    assert [tcode[res.node_type(c)] for c in names] == list(o_types)
    assert mm.hc.last.cells_scored == info["cells_scored"]
    assert res.num_arcs() >= 3


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 2e-5)])
def test_grouped_hybrid_slices_match_per_slice_chains(pbn, monkeypatch, dtype, tol):
    """Hybrid CKDE candidates through the grouped evaluation (kde_group.hip: one pool per configuration and term, all folds of all
    configurations in one launch chain; fp32: f16x2 fragments, two plain terms) against the per-(fold, configuration) chains of
    round 2 (fp32: the fused joint + marginal sweep): CV and validation scores of five candidates with one and two discrete parents.
    fp64: the same pairs in another order, equal to rounding; fp32: also another split of the terms."""
    rng = np.random.default_rng(9)
    n = 300_000
    A, B = rng.integers(0, 3, size=n), rng.integers(0, 2, size=n)
    x = rng.normal(size=n) + 1.5 * A
    y = 0.6 * x + np.where(B == 1, 1.0, -1.0) + rng.normal(scale=0.7, size=n)
    z = np.tanh(y) + 0.4 * x + rng.normal(scale=0.5, size=n)
    df = pd.DataFrame({"x": x.astype(dtype), "y": y.astype(dtype), "z": z.astype(dtype)})
    df["A"] = pd.Categorical.from_codes(A, ["a0", "a1", "a2"])
    df["B"] = pd.Categorical.from_codes(B, ["b0", "b1"])
    bn = pbn.SemiparametricBN(list(df.columns), [], [("A", pbn.DiscreteFactorType()), ("B", pbn.DiscreteFactorType())])
    cands = (("y", ["x", "B"]), ("z", ["x", "y", "A"]), ("x", ["A"]), ("z", ["A", "B"]), ("y", ["x", "z", "A", "B"]))
    res = {}
    # two implementations of the same sums: compared with the pruning margins pinned where what is dropped is below the tolerance
    # (the two forms bound the queries' sums differently, so at the shipped margins they drop different - equally negligible - tiles)
    monkeypatch.setenv("PBN_PRUNE_MARGIN", "52")
    monkeypatch.setenv("PBN_PRUNE_MARGIN_F32", "40")
    for grouped in ("1", "0"):
        monkeypatch.setenv("PBN_SCORE_GROUPED", grouped)
        score = pbn.ValidatedLikelihood(df, 0.2, 4, 1)
        res[grouped] = np.array([score.local_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in cands] +
                                [score.vlocal_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in cands])
        entries, sweeps = score.kde_cache_stats()
        assert sweeps > 0
    monkeypatch.delenv("PBN_SCORE_GROUPED")
    monkeypatch.delenv("PBN_PRUNE_MARGIN")
    monkeypatch.delenv("PBN_PRUNE_MARGIN_F32")
    assert np.all(np.isfinite(res["1"]))
    assert np.allclose(res["1"], res["0"], rtol=tol, atol=0), (res["1"] - res["0"]) / res["0"]
    # the shipped margins: inside their dropped-mass bound (3e-7 / 3e-5 of a sum)
    score = pbn.ValidatedLikelihood(df, 0.2, 4, 1)
    shipped = np.array([score.local_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in cands] +
                       [score.vlocal_score_node_type(bn, pbn.CKDEType(), v, p) for v, p in cands])
    assert np.allclose(shipped, res["1"], rtol=3e-7 if dtype == "float64" else 3e-5, atol=0)


def test_fp32_engine_redoes_far_out_sets_on_fp64_fragments(pbn, oracle, monkeypatch):
    """fp32 tables in the SCORE ENGINE: the pack kernels report |z|^2 of every evaluation's farthest whitened training row; an
    evaluation beyond what the fp32 Gram form holds (2^-24 |z|^2 > 5e-4: here a small cluster 40 sigma from the bulk) is redone on
    fp64 fragments before its value is used and its variable set stays on them (pbn_scoredata::widen_sets).  Plain CKDE terms
    (CVLikelihood) and hybrid slices (discrete parent), against the oracle in fp64 arithmetic on the same float data; with the
    check switched off the error of the fp32 fragments on the cluster's rows is what is left."""
    rng = np.random.default_rng(123)
    n, far = 60_000, 60
    a = rng.normal(size=n)
    b = np.tanh(a) + 0.5 * rng.normal(size=n)
    idx = rng.choice(n, size=far, replace=False)
    a[idx] = 400.0 + 0.3 * rng.normal(size=far)
    b[idx] = -250.0 + 0.3 * rng.normal(size=far)
    D = rng.integers(0, 2, size=n)
    df = pd.DataFrame({"a": a, "b": b}).astype(np.float32)
    df["D"] = pd.Categorical.from_codes(D, ["d0", "d1"])
    net = pbn.SemiparametricBN(list(df.columns), [], [("D", pbn.DiscreteFactorType())])
    data64 = df[["b", "a"]].to_numpy().astype(np.float64)
    want_plain = oracle.cv_likelihood(data64, "ckde", 3, 2)
    tr, te = oracle.holdout_split(n, 0.2, 2)
    want_hyb = oracle.adaptator_fit_slogl(data64, [D.astype(np.int32)], [2], tr, te, "ckde")

    def run():
        cv = pbn.CVLikelihood(df, 3, 2)
        ho = pbn.HoldoutLikelihood(df, 0.2, 2)
        plain = cv.local_score_node_type(net, pbn.CKDEType(), "b", ["a"])
        hyb = ho.local_score_node_type(net, pbn.CKDEType(), "b", ["a", "D"])
        again = cv.local_score_node_type(net, pbn.CKDEType(), "a", ["b"])      # the joint set {a, b} is known by now: straight to fp64
        return plain, hyb, again, cv.kde_cache_stats()[1]

    plain, hyb, again, sweeps = run()
    assert abs(plain - want_plain) <= 2e-6 * abs(want_plain), (plain, want_plain)
    assert abs(hyb - want_hyb) <= 2e-6 * abs(want_hyb), (hyb, want_hyb)
    want_again = oracle.cv_likelihood(data64[:, ::-1].copy(), "ckde", 3, 2)
    assert abs(again - want_again) <= 2e-6 * abs(want_again)
    monkeypatch.setenv("PBN_F32_WIDEN_AT", "inf")
    plain32, hyb32, _, sweeps32 = run()
    assert sweeps > sweeps32                                   # the flagged evaluations were made twice
    assert abs(plain32 - want_plain) > 5 * abs(plain - want_plain) and abs(hyb32 - want_hyb) > 5 * abs(hyb - want_hyb)
    assert abs(plain32 - want_plain) <= RTOL_F32 * abs(want_plain)   # (a handful of rows: still inside the fp32 bar - see DESIGN.md 4)
