"""GPU tier: LinearCorrelation (device covariance + host partial correlations / Student-t tail) against numpy eigh +
scipy (Boost) and MMHC end to end against the Python restatement of MMPC feeding the same hill-climb."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def dag_data(n_rows, n_cols, seed, dtype=np.float64):
    rng = np.random.default_rng(seed)
    cols = []
    for j in range(n_cols):
        k = int(rng.integers(0, min(3, j) + 1))
        parents = rng.choice(j, size=k, replace=False) if k else []
        col = rng.normal(scale=rng.uniform(0.5, 1.5), size=n_rows)
        for p in parents:
            col = col + rng.uniform(-1.5, 1.5) * cols[int(p)]
        cols.append(col)
    return pd.DataFrame(np.column_stack(cols).astype(dtype), columns=[f"x{i}" for i in range(n_cols)])


@pytest.mark.parametrize("rows", [40, 5000, 400000])
def test_linear_correlation_pvalues(pbn, rows):
    from oracle import mmpc_oracle

    df = dag_data(rows, 9, 4)
    df["x8"] = df["x0"] * 2.0 + 1.0                    # exactly collinear pair: pseudo-inverse path, cor = +-1
    test = pbn.LinearCorrelation(df)
    cov = np.cov(df.to_numpy(), rowvar=False)
    assert np.allclose(test.covariance(), cov, rtol=1e-10, atol=1e-12)
    names = list(df.columns)
    rng = np.random.default_rng(0)
    checked = 0
    for _ in range(150):
        k = int(rng.integers(0, 6))
        sel = rng.choice(9, size=k + 2, replace=False)
        got = test.pvalue(names[sel[0]], names[sel[1]], [names[i] for i in sel[2:]] if k != 1 else names[sel[2]])
        want = mmpc_oracle.lincor_pvalue(cov, rows, int(sel[0]), int(sel[1]), [int(i) for i in sel[2:]])
        if {0, 8} <= set(sel.tolist()):
            continue                                     # singular block: eigenvector basis of the null space is arbitrary
        assert got == pytest.approx(want, rel=2e-9, abs=1e-300), (sel, got, want)
        checked += 1
    assert checked > 100
    assert test.pvalue("x0", "x8") == 0.0
    with pytest.raises(ValueError, match="not present"):
        test.pvalue("x0", "nope")
    assert test.num_variables() == 9 and test.has_variables(["x1", "x2"]) and not test.has_variables("q")


def test_mmhc_gaussian_end_to_end(pbn):
    from oracle import mmpc_oracle
    from pybnesian_amd.independences import mmpc_cpcs

    df = dag_data(20000, 12, 11)
    names = list(df.columns)
    test = pbn.LinearCorrelation(df)
    cov = test.covariance()
    got, ntests = mmpc_cpcs(test, names, 0.05)
    want, calls = mmpc_oracle.mmpc_all_variables(lambda a, b, c: mmpc_oracle.lincor_pvalue(cov, len(df), a, b, c), len(names), 0.05)
    assert [[names.index(v) for v in c] for c in got] == want and ntests == calls
    score = pbn.BIC(df)
    mm = pbn.MMHC()
    model = mm.estimate(test, pbn.ArcOperatorSet(), score, bn_type=pbn.GaussianNetworkType(), alpha=0.05)
    # the same search restricted by the restatement's CPCs
    blacklist = [(a, b) for i, a in enumerate(names) for j, b in enumerate(names) if i != j and j not in want[i]]
    ref = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names), arc_blacklist=blacklist)
    assert sorted(model.arcs()) == sorted(ref.arcs()) and model.num_arcs() > 5
    for a, b in model.arcs():
        assert names.index(b) in want[names.index(a)]
    # MMHC with a semiparametric network and a validated score runs through the same path
    sp = pbn.MMHC().estimate(test, pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]),
                             pbn.ValidatedLikelihood(df.iloc[:3000], k=3, seed=0), bn_type=pbn.SemiparametricBNType(), max_iters=3)
    assert sp.num_arcs() <= 3
    with pytest.raises(ValueError, match="do not contain all the variables"):
        mm.estimate(test, pbn.ArcOperatorSet(), score, nodes=["x0", "zz"])


def test_linear_correlation_with_nulls(pbn):
    """continuous/linearcorrelation.cpp:20-122 (pvalue_impl): with nulls every test uses the covariance of ITS variables
    over the rows valid in all of them and valid_rows - 2 - |cond| degrees of freedom."""
    from oracle import mmpc_oracle

    df = dag_data(6000, 7, 2)
    rng = np.random.default_rng(4)
    for c, frac in (("x1", 0.04), ("x3", 0.07), ("x6", 0.02)):
        df.loc[df.index[rng.random(len(df)) < frac], c] = np.nan
    test = pbn.LinearCorrelation(df)
    names = list(df.columns)
    for _ in range(60):
        k = int(rng.integers(0, 4))
        sel = [names[i] for i in rng.choice(7, size=k + 2, replace=False)]
        sub = df[sel].dropna()
        cov = np.atleast_2d(np.cov(sub.to_numpy(), rowvar=False))
        want = mmpc_oracle.lincor_pvalue(cov, len(sub), 0, 1, list(range(2, k + 2)))
        got = test.pvalue(sel[0], sel[1], sel[2:] if k != 1 else sel[2])
        assert got == pytest.approx(want, rel=1e-8, abs=1e-300), (sel, got, want)
    with pytest.raises(ValueError, match="keeps no covariance"):
        test.covariance()
    # the same test drives MMPC through the native callback
    from pybnesian_amd.independences import mmpc_cpcs

    cpcs, ntests = mmpc_cpcs(test, names, 0.01)
    clean, _ = mmpc_cpcs(pbn.LinearCorrelation(dag_data(6000, 7, 2)), names, 0.01)
    assert ntests > 0 and [sorted(c) for c in cpcs] == [sorted(c) for c in clean]
