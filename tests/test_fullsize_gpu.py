"""GPU tier: BASELINE.json full sizes.  Size-independent properties (the oracle cannot finish the whole configurations): C2
(ProductKDE.slogl fp64, 1e6 x 1e5, d = 8), C4 (64-node BGe hill-climb on 2M rows), C3, C5 - and, since round 4, ORACLE VALUES at
those sizes wherever the oracle finishes in seconds: ~1 100 test rows of C2 against all 1e6 training rows (incl. the queries next to
the farthest-out training rows, where the sweep's rare paths run), C3 candidates on one full-size fold (CKDE handles: 512 test rows
against 450 000 training rows; the score engine's grouped path: a hold-out score with 499 000 training and 1 000 test rows), C5's
fp32 hybrid slices the same way, C4's BGe local scores on all 2M rows.  The reference's tests compare values, not properties
(tests/factors/continuous/KDE_test.py:167-203, CKDE_test.py:316-349)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def env():
    import torch

    import pybnesian_amd as pbn
    from pybnesian_amd import _lib

    assert torch.cuda.is_available(), "the gpu tier needs an MI355X (and torch imported before the library touches it: conftest.py)"
    return torch, pbn, _lib, pbn.Context(0)


def _table(torch, pbn, _lib, ctx, t, names):
    torch.cuda.synchronize()
    return pbn.DeviceTable.from_device_pointer(ctx, t.data_ptr(), t.shape[1], names, t.shape[1], _lib.PBN_F64, keepalive=t)


def test_c2_fullsize_properties(env):
    torch, pbn, _lib, ctx = env
    import bench

    dev = torch.device("cuda", 0)
    train_t, test_t = bench.make_tables(torch, dev, 1_000_000, 100_000, 0, 1, torch.float64)
    names = [f"v{i}" for i in range(8)]
    train, test = _table(torch, pbn, _lib, ctx, train_t, names), _table(torch, pbn, _lib, ctx, test_t, names)
    kde = pbn.ProductKDE(names)
    kde.fit_table(train)
    s = kde.slogl_table(test)
    assert np.isfinite(s) and s < 0
    # additivity over a ragged split of the test rows, and run-to-run bit reproducibility
    s1 = kde.slogl_table(test, row0=0, n=33_333)
    s2 = kde.slogl_table(test, row0=33_333, n=66_667)
    assert abs((s1 + s2) - s) <= 1e-10 * abs(s)
    assert kde.slogl_table(test) == s
    # the same rows in a different training order (bandwidth held fixed): only summation order changes
    perm = torch.randperm(1_000_000, device=dev, generator=torch.Generator(device=dev).manual_seed(5))
    train_p = train_t[:, perm].contiguous()
    kde2 = pbn.ProductKDE(names)
    kde2.fit_table(_table(torch, pbn, _lib, ctx, train_p, names))
    kde2.bandwidth = kde.bandwidth
    assert abs(kde2.slogl_table(test, row0=0, n=20_000) - kde.slogl_table(test, row0=0, n=20_000)) <= 1e-9 * abs(s)
    # shifting every column by a constant leaves the likelihood unchanged (centring is exact in the distances)
    shift = torch.arange(8, device=dev, dtype=torch.float64)[:, None] * 100.0
    kde3 = pbn.ProductKDE(names)
    kde3.fit_table(_table(torch, pbn, _lib, ctx, (train_t + shift).contiguous(), names))
    got = kde3.slogl_table(_table(torch, pbn, _lib, ctx, (test_t[:, :20_000] + shift).contiguous(), names))
    assert abs(got - kde.slogl_table(test, row0=0, n=20_000)) <= 1e-7 * abs(got)


def test_c2_fullsize_against_the_oracle(env):
    """bench.py's parity block as a test: the per-row logl of the whole C2 test table (the timed sweep's own launch shape) against
    oracle/pbn_oracle.cpp on the first 1024 rows + the 8 test rows nearest each of the 8 farthest-out training rows (||z||^2 > 1780:
    weights that underflow, redone chunks), all 1e6 training rows each - 1.1e9 pairs on the host."""
    torch, pbn, _lib, ctx = env
    import bench

    dev = torch.device("cuda", 0)
    train_t, test_t = bench.make_tables(torch, dev, 1_000_000, 100_000, 0, 1, torch.float64)
    names = [f"v{i}" for i in range(8)]
    train, test = _table(torch, pbn, _lib, ctx, train_t, names), _table(torch, pbn, _lib, ctx, test_t, names)
    kde = pbn.ProductKDE(names)
    kde.fit_table(train)
    par = bench.parity_block(torch, kde, test, train_t, test_t, kde.slogl_table(test), 1e-6)
    assert par["ok"], par
    assert par["rows"] >= 1024 + 8 and par["max_whitened_norm2_of_training_rows"] > 1000.0
    assert par["max_rel_logl"] <= 1e-8 and par["rel_slogl"] <= 1e-8, par     # the bar is 1e-6; per-row logl: degree-6 2^x (2.3e-9 absolute), slogl: 2^f on the fp32 unit
    # the full-covariance KDE (the fused sweep without the diagonal shortcut) on a 4-column subset, same rows
    kf = pbn.KDE(names[:4])
    kf.fit_table(train)
    from oracle import oracle

    rows = np.arange(0, 100_000, 390)
    got = kf.logl_table(test)[rows]
    want = oracle.kde_logl(train_t[:4].T.cpu().numpy(), np.asarray(kf.bandwidth), test_t[:4].T.cpu().numpy()[rows])
    assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-8


def test_c4_fullsize_against_the_oracle(env):
    """BGe local scores and the BIC fit on ALL 2M rows against the oracle (two-pass covariance, the reference's BGe arithmetic)."""
    torch, pbn, _lib, ctx = env
    import bench
    from oracle import oracle

    dev = torch.device("cuda", 0)
    t = bench.make_dag_table(torch, dev, 2_000_000, 64, 2, torch.float64)
    names = [f"x{i}" for i in range(64)]
    table = _table(torch, pbn, _lib, ctx, t, names)
    score = pbn.BGe(None, table=table)
    bic = pbn.BIC(None, table=table)
    net = pbn.GaussianNetwork(names)
    for v, par in ((7, []), (12, [3]), (40, [5, 17, 33]), (63, [1, 2, 30, 44, 60])):
        cols = t[[v] + par].T.cpu().numpy()
        want = oracle.bge(cols, 64)
        got = score.local_score(net, names[v], [names[p_] for p_ in par])
        assert abs(got - want) <= 1e-9 * abs(want), (v, par, got, want)
        want_b = oracle.bic_lg(cols)
        got_b = bic.local_score(net, names[v], [names[p_] for p_ in par])
        assert abs(got_b - want_b) <= 1e-9 * abs(want_b), (v, par, got_b, want_b)


def test_c3_fullsize_against_the_oracle(env):
    """C3's table (500 000 rows, fp64): (1) CKDE handles fitted on the training part of fold 0 (450 000 rows) - logl of 512 test-fold
    rows vs oracle.ckde_logl, one candidate with 1 parent (pruned, split into two plain sweeps) and one with 2; (2) the SCORE ENGINE's
    grouped, pruned path at full training size: HoldoutLikelihood with 1 000 test rows (499 000 training rows) vs
    oracle.holdout_likelihood, for 1, 2 and 3 parents."""
    torch, pbn, _lib, ctx = env
    import bench
    import pandas as pd
    from oracle import oracle

    dev = torch.device("cuda", 0)
    t = bench.make_dag_table(torch, dev, 500_000, 32, 2, torch.float64, nonlinear=True)
    names = [f"x{i}" for i in range(32)]
    for var, par in (("x5", ["x1"]), ("x9", ["x2", "x4"])):
        cols = [var] + par
        host = pd.DataFrame(t[[int(c[1:]) for c in cols]].T.cpu().numpy(), columns=cols)
        tr_idx, te_idx = next(iter(pbn.CrossValidation(host, 10, 0).indices()))
        cpd = pbn.CKDE(var, par)
        cpd.fit(host.iloc[tr_idx])
        test = host.iloc[te_idx[:512]]
        got = cpd.logl(test)
        want = oracle.ckde_logl(host.to_numpy()[tr_idx], np.asarray(cpd.bandwidth), test.to_numpy())
        assert np.max(np.abs(got - want) / np.maximum(1.0, np.abs(want))) <= 1e-6, (var, par)
        assert abs(got.sum() - want.sum()) <= 1e-8 * abs(want.sum())
    table = _table(torch, pbn, _lib, ctx, t, names)
    score = pbn.HoldoutLikelihood(None, 0.002, 0, table=table)
    start = pbn.SemiparametricBN(names, [], [(n_, pbn.CKDEType()) for n_ in names])
    for var, par in (("x5", ["x1"]), ("x9", ["x2", "x4"]), ("x20", ["x3", "x7", "x11"])):
        data = t[[int(c[1:]) for c in [var] + par]].T.cpu().numpy()
        want = oracle.holdout_likelihood(data, "ckde", 0.002, 0)
        got = score.local_score_node_type(start, pbn.CKDEType(), var, par)
        assert abs(got - want) <= 1e-6 * abs(want), (var, par, got, want)
        assert abs(got - want) <= 1e-7 * abs(want), (var, par, got, want)    # measured far inside the bar (the sum-only sweeps' own bound: 2.5e-7 of a sum)


def test_c4_fullsize_properties(env):
    torch, pbn, _lib, ctx = env
    import bench

    dev = torch.device("cuda", 0)
    t = bench.make_dag_table(torch, dev, 2_000_000, 64, 2, torch.float64)
    names = [f"x{i}" for i in range(64)]
    table = _table(torch, pbn, _lib, ctx, t, names)
    score = pbn.BGe(None, table=table)
    start = pbn.GaussianNetwork(names)
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, start)
    trace = [repr(op) for op in hc.last.trace]
    assert res.num_arcs() > 60 and score.score(res) > score.score(start)
    # every applied delta equals the change of the total score (decomposability), checked on a prefix of the trace
    model = pbn.GaussianNetwork(names)
    total = score.score(model)
    for op in hc.last.trace[:10]:
        op.apply(model)
        new = score.score(model)
        assert abs((new - total) - op.delta()) <= 1e-7 * abs(new)
        total = new
    # idempotence: a second run takes exactly the same decisions; restarting from the result changes nothing
    res2 = hc.estimate(pbn.ArcOperatorSet(), score, start)
    assert [repr(op) for op in hc.last.trace] == trace and sorted(res2.arcs()) == sorted(res.arcs())
    res3 = hc.estimate(pbn.ArcOperatorSet(), score, res)
    assert sorted(res3.arcs()) == sorted(res.arcs()) and len(hc.last.trace) == 0
    # BIC on the same moments agrees with a direct device fit of one node
    bic = pbn.BIC(None, table=table)
    node = max(names, key=lambda n_: res.num_parents(n_))
    beta, var = bic.mle_lg(node, res.parents(node))
    cpd = pbn.LinearGaussianCPD(node, res.parents(node))
    cpd.fit_table(table)
    assert np.allclose(beta, cpd.beta, rtol=1e-9) and np.isclose(var, cpd.variance, rtol=1e-9)


def _trace(pbn, hc):
    return [(repr(op), op.delta()) for op in hc.last.trace]


def test_c3_fullsize_properties(env, monkeypatch):
    """BASELINE config 3 at full size (32-node SemiparametricBN, all-CKDE start, 10-fold CVLikelihood, 500 000 rows fp64): what the
    miniature oracle comparisons of test_hc_gpu.py cannot reach, through properties - decomposability of the applied deltas,
    pruned / grouped evaluation against the unpruned per-(set, fold) sweeps on three candidates, fold additivity against explicit
    CKDE handles, and a bit-identical second run from a fresh score object."""
    torch, pbn, _lib, ctx = env
    import bench

    dev = torch.device("cuda", 0)
    t = bench.make_dag_table(torch, dev, 500_000, 32, 2, torch.float64, nonlinear=True)
    names = [f"x{i}" for i in range(32)]
    table = _table(torch, pbn, _lib, ctx, t, names)
    ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])
    start = pbn.SemiparametricBN(names, [], [(n_, pbn.CKDEType()) for n_ in names])

    def run():
        score = pbn.CVLikelihood(None, 10, 0, table=table)
        hc = pbn.GreedyHillClimbing()
        res = hc.estimate(ops, score, start, max_indegree=3, max_iters=3)
        return score, hc, res

    score, hc, res = run()
    trace = _trace(pbn, hc)
    assert len(trace) == 3 and hc.last.cells_scored > 1000
    # decomposability: every applied delta is the change of the sum of local scores
    model = start.clone()
    total = score.score(model)
    for op in hc.last.trace:
        op.apply(model)
        new = score.score(model)
        assert abs((new - total) - op.delta()) <= 1e-9 * abs(new), repr(op)
        total = new
    # three candidates (1, 2, 3 parents): grouped + pruned (default) = per-unit pruned = unpruned, from fresh score objects
    cands = [("x5", ["x1"]), ("x9", ["x2", "x4"]), ("x20", ["x3", "x7", "x11"])]
    ref = [score.local_score_node_type(start, pbn.CKDEType(), v, p) for v, p in cands]
    # (the default margin of the sum-only sweeps bounds the dropped mass at 1.1e-7 of a sum, and which tiles are dropped depends on
    #  the boxes of the evaluation form: 3e-7; tests/test_switches_gpu.py compares the forms with the margin pinned at 52, at 1e-10)
    for env_kv, tol in (({"PBN_SCORE_GROUPED": "0"}, 3e-7), ({"PBN_SWEEP_PRUNE": "0"}, 3e-7)):
        for k_, v_ in env_kv.items():
            monkeypatch.setenv(k_, v_)
        other = pbn.CVLikelihood(None, 10, 0, table=table)
        got = [other.local_score_node_type(start, pbn.CKDEType(), v, p) for v, p in cands]
        for k_ in env_kv:
            monkeypatch.delenv(k_)
        assert np.allclose(got, ref, rtol=tol, atol=0), (env_kv, got, ref)
    # fold additivity against explicit factors: the CV score of x5 | x1 is the sum over the folds of CKDE.fit(train).slogl(test)
    import pandas as pd

    host = pd.DataFrame(t[[5, 1]].T.cpu().numpy(), columns=["x5", "x1"])
    acc = 0.0
    for tr_idx, te_idx in pbn.CrossValidation(host, 10, 0).indices():
        cpd = pbn.CKDE("x5", ["x1"])
        cpd.fit(host.iloc[tr_idx])
        acc += cpd.slogl(host.iloc[te_idx])
    assert abs(acc - ref[0]) <= 1e-9 * abs(acc)
    # run-to-run: a fresh score object and search take bit-identical decisions
    _, hc2, res2 = run()
    assert _trace(pbn, hc2) == trace and sorted(res2.arcs()) == sorted(res.arcs())


def test_c5_fullsize_properties(env, monkeypatch):
    """BASELINE config 5 at full size (48 columns, 16 discrete, 1 000 000 rows fp32, ValidatedLikelihood(0.2, 10)): decomposability
    of the applied deltas, pruned against unpruned slices on three hybrid candidates, validation-score additivity against explicit
    factors over the hold-out split, and a bit-identical second run."""
    torch, pbn, _lib, ctx = env
    import pandas as pd

    n_rows, n_disc, n_cont = 1_000_000, 16, 32
    rng = np.random.default_rng(3)
    cards = rng.integers(2, 5, size=n_disc)
    disc = {}
    for j in range(n_disc):
        base = rng.integers(0, cards[j], size=n_rows)
        if j > 0:
            base = np.where(rng.random(n_rows) < 0.3, disc[f"D{j - 1}"] % cards[j], base)
        disc[f"D{j}"] = base.astype(np.int32)
    import bench

    t = bench.make_dag_table(torch, torch.device("cuda", 0), n_rows, n_cont, 3, torch.float32, nonlinear=True).cpu().numpy()
    df = pd.DataFrame({f"x{j}": t[j] + 1.5 * disc[f"D{j % n_disc}"].astype(np.float32) for j in range(n_cont)})
    for j in range(n_disc):
        df[f"D{j}"] = pd.Categorical.from_codes(disc[f"D{j}"], [f"c{v}" for v in range(cards[j])])
    names = list(df.columns)
    pairs = [(a, b) for a in names for b in names if a != b]
    keep = rng.random(len(pairs)) < 0.15
    blacklist = [p for p, k_ in zip(pairs, keep) if not k_]
    start = pbn.SemiparametricBN(names, [], [(f"D{j}", pbn.DiscreteFactorType()) for j in range(n_disc)])
    ops = pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()])

    def run():
        score = pbn.ValidatedLikelihood(df, 0.2, 10, 0)
        hc = pbn.GreedyHillClimbing()
        res = hc.estimate(ops, score, start, max_indegree=3, arc_blacklist=blacklist, max_iters=4)
        return score, hc, res

    score, hc, res = run()
    trace = _trace(pbn, hc)
    assert len(trace) >= 2
    model = start.clone()
    total = score.score(model)
    for op in hc.last.trace:
        op.apply(model)
        new = score.score(model)
        assert abs((new - total) - op.delta()) <= 2e-6 * abs(new), repr(op)     # fp32 table: sums of ~1e6 float-accurate terms
        total = new
    cands = [("x3", pbn.CKDEType(), ["D3"]), ("x8", pbn.CKDEType(), ["x2", "D8"]), ("x17", pbn.LinearGaussianCPDType(), ["x4", "D1", "D2"])]
    ref = [score.local_score_node_type(start, ty, v, p) for v, ty, p in cands]
    vref = [score.vlocal_score_node_type(start, ty, v, p) for v, ty, p in cands]
    monkeypatch.setenv("PBN_SWEEP_PRUNE", "0")
    other = pbn.ValidatedLikelihood(df, 0.2, 10, 0)
    got = [other.local_score_node_type(start, ty, v, p) for v, ty, p in cands]
    monkeypatch.delenv("PBN_SWEEP_PRUNE")
    assert np.allclose(got, ref, rtol=3e-5, atol=0), (got, ref)       # fp32 margin 36: at most 1.5e-5 of a sum dropped
    # validation score of x3 | D3 (HCKDE) = the factor fitted on the hold-out training part, evaluated on its test part
    ho = pbn.HoldOut(df[["x3", "D3"]], 0.2, 0)
    f = pbn.HCKDE("x3", ["D3"])
    f.fit(ho.training_data())
    v_direct = f.slogl(ho.test_data())
    assert abs(v_direct - vref[0]) <= 1e-3 * abs(v_direct)                      # fp32 slices: the north star's fp32 bar
    _, hc2, res2 = run()
    assert _trace(pbn, hc2) == trace and sorted(res2.arcs()) == sorted(res.arcs())
    # ORACLE values for fp32 hybrid slices at full slice size: a hold-out score with 1 000 test rows (999 000 training rows, cut by the
    # discrete parents into slices of 250-500 k rows: the grouped, pruned f16x2 sweeps) against the per-slice restatement
    # (DiscreteAdaptator.hpp:201-348) in fp64 arithmetic on the same float data, at the north star's fp32 bar
    from oracle import oracle

    small = pbn.HoldoutLikelihood(df, 0.001, 0)
    tr, te = oracle.holdout_split(n_rows, 0.001, 0)
    for var, dpar, cpar in (("x3", ["D3"], []), ("x8", ["D8"], ["x2"]), ("x21", ["D5"], ["x4", "x9"])):
        cont = df[[var] + cpar].to_numpy().astype(np.float64)
        want = oracle.adaptator_fit_slogl(cont, [disc[d_] for d_ in dpar], [int(cards[int(d_[1:])]) for d_ in dpar], tr, te, "ckde")
        got = small.local_score_node_type(start, pbn.CKDEType(), var, cpar + dpar)
        assert abs(got - want) <= 1e-3 * abs(want), (var, dpar, cpar, got, want)
        # ... and against the reference's OWN float arithmetic on these slices (covariance, distances, logsumexp, logl in float:
        # NormalReferenceRule.hpp:124-133, KDE.hpp:466-470) at the reference tests' fp32 tolerance, atol 5e-4 per logl (KDE_test.py:181-182)
        want32 = oracle.adaptator_fit_slogl(cont, [disc[d_] for d_ in dpar], [int(cards[int(d_[1:])]) for d_ in dpar], tr, te, "ckde", arithmetic="float32")
        assert abs(got - want32) <= 5e-4 * te.size, (var, dpar, cpar, got, want32, want)
