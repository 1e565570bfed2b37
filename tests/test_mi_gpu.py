"""GPU tier: hybrid MutualInformation (device per-configuration moments + host formulas) against the numpy restatement
of learning/independences/hybrid/mutual_information.cpp, every overload: discrete/discrete, mixed, continuous, with
discrete, continuous and mixed conditioning sets; degrees of freedom; chi-square tails; MMHC on a hybrid table."""
import itertools

import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def hybrid_table(n, seed, dtype="float64"):
    rng = np.random.default_rng(seed)
    d1 = rng.integers(0, 3, size=n)
    d2 = (d1 + rng.integers(0, 2, size=n) * (rng.random(n) < 0.6)) % 2
    d3 = rng.integers(0, 4, size=n)
    c1 = rng.normal(size=n) + 0.8 * d1
    c2 = 0.7 * c1 + rng.normal(scale=0.6, size=n) - 0.5 * d2
    c3 = rng.normal(size=n) * (1 + 0.3 * d3)
    c4 = 0.4 * c2 - 0.6 * c3 + rng.normal(scale=0.8, size=n)
    df = pd.DataFrame({"c1": c1, "c2": c2, "c3": c3, "c4": c4}).astype(dtype)
    for name, codes, k in (("d1", d1, 3), ("d2", d2, 2), ("d3", d3, 4)):
        df[name] = pd.Categorical.from_codes(codes, [f"{name}_{i}" for i in range(k)])
    return df


def make_oracle(df, asymptotic=True):
    from oracle.mi_oracle import MIOracle

    cols = {}
    for c in df.columns:
        if isinstance(df[c].dtype, pd.CategoricalDtype):
            cols[c] = (df[c].cat.codes.to_numpy().astype(np.int64), len(df[c].cat.categories))
        else:
            cols[c] = df[c].to_numpy().astype(np.float64)
    return MIOracle(cols, asymptotic)


@pytest.mark.parametrize("n,dtype", [(600, "float64"), (50000, "float64"), (20000, "float32")])
def test_mi_all_overloads(pbn, n, dtype):
    df = hybrid_table(n, 5, dtype)
    test = pbn.MutualInformation(df)
    orc = make_oracle(df)
    names = list(df.columns)
    rng = np.random.default_rng(1)
    rel = 1e-9 if dtype == "float64" else 2e-4
    cases = [(x, y, ()) for x, y in itertools.combinations(names, 2)]
    for _ in range(120):
        k = int(rng.integers(1, 5))
        sel = [names[i] for i in rng.choice(len(names), size=k + 2, replace=False)]
        cases.append((sel[0], sel[1], tuple(sel[2:])))
    for x, y, z in cases:
        zz = None if not z else (z[0] if len(z) == 1 else list(z))
        got_mi, want_mi = test.mi(x, y, zz), orc.mi(x, y, z)
        assert got_mi == pytest.approx(want_mi, rel=rel, abs=(1e-11 if dtype == "float64" else 1e-6)), (x, y, z)
        assert test.degrees_of_freedom(x, y, zz) == orc.df(x, y, z)
        if dtype == "float64":
            assert test.pvalue(x, y, zz) == pytest.approx(orc.pvalue(x, y, z), rel=1e-6, abs=1e-300), (x, y, z)
    dev, host = test.passes()
    assert dev > 0 and host == 0
    assert pbn.MutualInformation(df, asymptotic_df=False).degrees_of_freedom("d1", "c1", ["c2", "d3"]) == make_oracle(df, False).df("d1", "c1", ["c2", "d3"])
    with pytest.raises(ValueError, match="not present"):
        test.pvalue("c1", "zz")


def test_mi_many_configurations(pbn):
    """1024 configurations x 21 statistics.  With up to sixteen continuous variables the rows are grouped by configuration
    once and every block sums one configuration in registers (one device pass); with more, the LDS-cell kernel covers
    the configurations in several windows - both on the device, same numbers as the restatement."""
    rng = np.random.default_rng(3)
    n = 30000
    df = pd.DataFrame({f"c{i}": rng.normal(size=n) for i in range(20)})
    for j in range(6):
        df[f"d{j}"] = pd.Categorical.from_codes(rng.integers(0, 4, size=n), [f"k{i}" for i in range(4)])
    df["c1"] = df["c1"] + 0.5 * df["c0"] + 0.3 * df["d0"].cat.codes
    test, orc = pbn.MutualInformation(df), make_oracle(df)
    z = ["d1", "d2", "d3", "d4", "c2", "c3", "c4"]
    assert test.mi("d0", "c1", z) == pytest.approx(orc.mi("d0", "c1", z), rel=1e-8)
    assert test.mi("c1", "d0", z[::-1]) == pytest.approx(orc.mi("c1", "d0", z[::-1]), rel=1e-8)   # another variable order, same grouping
    dev, host = test.passes()
    assert host == 0 and dev == 2
    z = ["d1", "d2", "d3", "c2", "c3", "c4", "c5", "c6", "c7"]   # 7 continuous variables: still the register kernel
    assert test.mi("d0", "c1", z) == pytest.approx(orc.mi("d0", "c1", z), rel=1e-8)
    dev1, host = test.passes()
    assert host == 0 and dev1 == dev + 1
    z = ["d1", "d2", "d3"] + [f"c{i}" for i in range(2, 19)]     # 18 continuous variables: 256 x 190 statistics, windowed
    assert test.mi("d0", "c1", z) == pytest.approx(orc.mi("d0", "c1", z), rel=1e-7)
    dev2, host = test.passes()
    assert host == 0 and dev2 >= dev1 + 2


def test_mi_grouping_cache_and_count_only_tests(pbn):
    """Tests over the same set of discrete variables share one row grouping; a purely discrete test reads its counts off
    the grouping (ChiSquare and discrete-discrete MutualInformation make no data pass of their own)."""
    df = hybrid_table(20000, 5)
    test, orc = pbn.MutualInformation(df), make_oracle(df)
    chi = pbn.ChiSquare(df)
    for x, y, z in (("d1", "d2", ["d3"]), ("d2", "d1", ["d3"]), ("d3", "d1", ["d2"]), ("d1", "d2", [])):
        assert test.mi(x, y, z) == pytest.approx(orc.mi(x, y, z), rel=1e-9, abs=1e-12)
    from scipy.stats import chi2_contingency

    tab = pd.crosstab(df["d1"], df["d2"]).to_numpy()
    assert chi.pvalue("d1", "d2") == pytest.approx(chi2_contingency(tab, correction=False)[1], rel=1e-9, abs=1e-300)
    for x, y, z in (("c1", "d2", ["d1"]), ("c2", "c1", ["d1", "d2"]), ("d1", "c2", ["d2", "c1"]), ("c3", "c4", ["c1"])):
        assert test.mi(x, y, z) == pytest.approx(orc.mi(x, y, z), rel=1e-9, abs=1e-12)


def test_mmhc_hybrid(pbn):
    from oracle import mmpc_oracle
    from pybnesian_amd.independences import mmpc_cpcs

    df = hybrid_table(20000, 9)
    names = list(df.columns)
    test, orc = pbn.MutualInformation(df), make_oracle(df)
    got, ntests = mmpc_cpcs(test, names, 0.05)
    want, calls = mmpc_oracle.mmpc_all_variables(lambda a, b, c: orc.pvalue(names[a], names[b], [names[i] for i in c]), len(names), 0.05)
    assert [[names.index(v) for v in c] for c in got] == want and ntests == calls
    assert any(want)
    score = pbn.BIC(df)
    model = pbn.MMHC().estimate(test, pbn.ArcOperatorSet(), score, bn_type=pbn.CLGNetworkType(), alpha=0.05)
    allowed = [set(c) for c in want]
    for a, b in model.arcs():
        assert names.index(b) in allowed[names.index(a)]
        assert not (isinstance(df[b].dtype, pd.CategoricalDtype) and not isinstance(df[a].dtype, pd.CategoricalDtype))
    assert model.num_arcs() >= 3


def test_chi_square(pbn):
    """ChiSquare (chi_square.cpp:8-139) against scipy's contingency test, marginal and conditional."""
    from scipy.stats import chi2, chi2_contingency

    rng = np.random.default_rng(2)
    n = 20000
    a = rng.integers(0, 3, size=n)
    b = (a + rng.integers(0, 3, size=n) * (rng.random(n) < 0.5)) % 3
    c = rng.integers(0, 2, size=n)
    d = (c + (rng.random(n) < 0.2)) % 2
    df = pd.DataFrame({"x": rng.normal(size=n)})
    for name, codes, k in (("a", a, 3), ("b", b, 3), ("c", c, 2), ("d", d, 2)):
        df[name] = pd.Categorical.from_codes(codes, [f"{name}{i}" for i in range(k)])
    test = pbn.ChiSquare(df)
    tab = pd.crosstab(df["a"], df["b"]).to_numpy()
    stat, p, dof, _ = chi2_contingency(tab, correction=False)
    assert test.pvalue("a", "b") == pytest.approx(p, rel=1e-8, abs=1e-300)
    assert test.pvalue("a", "c") == pytest.approx(chi2_contingency(pd.crosstab(df["a"], df["c"]).to_numpy(), correction=False)[1], rel=1e-8)
    # conditional: sum of the per-configuration statistics, df multiplied by the configurations
    for z in (["c"], ["c", "d"]):
        stat, cfgs = 0.0, 0
        for _, sub in df.groupby(z, observed=False):
            cfgs += 1
            if len(sub) == 0:
                continue
            t = pd.crosstab(sub["a"], sub["b"], dropna=False).reindex(index=df["a"].cat.categories, columns=df["b"].cat.categories, fill_value=0).to_numpy().astype(float)
            e = np.outer(t.sum(1), t.sum(0)) / t.sum()
            stat += np.where(e != 0, (t - e) ** 2 / np.where(e != 0, e, 1), 0).sum()
        want = chi2.sf(stat, (3 - 1) * (3 - 1) * cfgs)
        assert test.pvalue("a", "b", z if len(z) > 1 else z[0]) == pytest.approx(want, rel=1e-8, abs=1e-300)
    with pytest.raises(ValueError, match="not present"):
        test.pvalue("a", "x")
    from pybnesian_amd.independences import mmpc_cpcs

    cpcs, _ = mmpc_cpcs(test, ["a", "b", "c", "d"], 0.01)
    assert "b" in cpcs[0] and "d" in cpcs[2] and "c" not in cpcs[0]


def test_mi_and_chisquare_with_nulls(pbn):
    """Null cells (hybrid/mutual_information.cpp:152-215, the contains_null overloads): every test runs over the rows that
    are valid in all of ITS variables - here: the restatement on the table filtered per test."""
    from scipy.stats import chi2_contingency

    df = hybrid_table(20000, 11)
    rng = np.random.default_rng(5)
    for c, frac in (("c1", 0.03), ("c3", 0.05), ("d1", 0.04), ("d3", 0.02)):
        df.loc[df.index[rng.random(len(df)) < frac], c] = np.nan
    test = pbn.MutualInformation(df)
    cases = [("c1", "c2", []), ("c1", "c2", ["c3"]), ("c2", "c4", ["d2"]), ("c1", "d2", ["d1"]), ("d1", "d2", ["d3"]),
             ("c3", "d3", ["c1", "d1"]), ("d1", "c4", ["c3", "d3", "c2"]), ("c2", "c4", ["d1", "d3", "c1"])]
    for x, y, z in cases:
        sub = df.dropna(subset=[x, y] + z).reset_index(drop=True)
        orc = make_oracle(sub[[x, y] + z])
        assert test.mi(x, y, z or None) == pytest.approx(orc.mi(x, y, tuple(z)), rel=1e-8, abs=1e-11), (x, y, z)
        assert test.pvalue(x, y, z or None) == pytest.approx(orc.pvalue(x, y, tuple(z)), rel=1e-6, abs=1e-300), (x, y, z)
    chi = pbn.ChiSquare(df)
    sub = df.dropna(subset=["d1", "d3"])
    tab = pd.crosstab(sub["d1"], sub["d3"]).to_numpy()
    assert chi.pvalue("d1", "d3") == pytest.approx(chi2_contingency(tab, correction=False)[1], rel=1e-9, abs=1e-300)
    # batched form (what MMPC uses) gives the same p-values
    from pybnesian_amd.independences import mmpc_cpcs

    clean = hybrid_table(20000, 11)
    cp_null, _ = mmpc_cpcs(test, list(df.columns), 0.05)
    cp_clean, _ = mmpc_cpcs(pbn.MutualInformation(clean), list(clean.columns), 0.05)
    assert [sorted(c) for c in cp_null] == [sorted(c) for c in cp_clean]   # 2-5 % missing cells do not change the skeleton here
