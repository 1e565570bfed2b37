"""GPU tier: KMutualInformation (learning/independences/continuous/mutual_information.{hpp,cpp}) - the estimate of every
overload and the permutation p-values against the CPU restatement (oracle/pbn_oracle.cpp: oracle_kmi), which follows the
reference's own routines (sorted-window scan for one conditioning variable, ball scan for several) where the product runs
brute-force neighbour kernels.  The reference's tests hold no fixture for this class (parity unpinned beyond the
restatement)."""
import numpy as np
import pandas as pd
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def table(n, seed, dtype="float64"):
    rng = np.random.default_rng(seed)
    a = rng.normal(size=n)
    b = 0.8 * a + rng.normal(scale=0.6, size=n)
    c = np.tanh(b) + rng.normal(scale=0.4, size=n)
    d = rng.normal(size=n)
    e = 0.5 * a - 0.5 * d + rng.normal(scale=0.7, size=n)
    return pd.DataFrame({"a": a, "b": b, "c": c, "d": d, "e": e}).astype(dtype)


@pytest.mark.parametrize("n,k,dtype", [(257, 3, "float64"), (1200, 10, "float64"), (700, 5, "float32")])
def test_kmi_estimates(pbn, n, k, dtype):
    from oracle import oracle

    df = table(n, 2, dtype)
    test = pbn.KMutualInformation(df, k, seed=0)
    assert test.num_variables() == 5 and test.has_variables(["a", "e"]) and not test.has_variables("z")
    cases = [("a", "b", []), ("a", "d", []), ("a", "c", ["b"]), ("a", "e", ["d"]), ("b", "c", ["a", "d"]), ("a", "e", ["b", "c", "d"])]
    for x, y, z in cases:
        want, _ = oracle.kmi(df[[x, y] + z].to_numpy(dtype=np.float64), k)
        got = test.mi(x, y, z if len(z) != 1 else z[0])
        assert got == pytest.approx(want, rel=1e-10, abs=1e-12), (x, y, z)
    assert test.mi("a", "b") > 0.2 and abs(test.mi("a", "d")) < 0.05           # dependent / independent pair
    assert test.mi("a", "c", "b") < 0.5 * test.mi("a", "c")                    # b screens a off from c
    with pytest.raises(ValueError, match="not present"):
        test.mi("a", "zz")


def test_kmi_permutation_pvalues(pbn):
    from oracle import oracle

    df = table(400, 7)
    for seed, samples, nbrs in ((3, 40, 5), (11, 25, 3)):
        test = pbn.KMutualInformation(df, 4, seed=seed, shuffle_neighbors=nbrs, samples=samples)
        for x, y, z in (("a", "d", []), ("a", "b", []), ("a", "c", ["b"]), ("a", "e", ["d"]), ("d", "b", ["a", "c"])):
            _, want = oracle.kmi(df[[x, y] + z].to_numpy(), 4, seed, nbrs, samples)
            got = test.pvalue(x, y, z or None)
            assert got == want, (seed, x, y, z, got, want)             # same permutations, same counts: exactly equal
    test = pbn.KMutualInformation(df, 4, seed=1, samples=60)
    assert test.pvalue("a", "b") == 0.0 and test.pvalue("a", "d") > 0.05
    assert test.pvalue("a", "c", "b") > 0.05 and test.pvalue("a", "e", "d") < 0.05


def test_kmi_drives_mmpc_and_dynamic_adaptator(pbn):
    from pybnesian_amd.independences import mmpc_cpcs

    df = table(500, 9)[["a", "b", "d"]]
    test = pbn.KMutualInformation(df, 5, seed=2, samples=30)
    cpcs, ntests = mmpc_cpcs(test, list(df.columns), 0.05)
    assert ntests >= 3 and sorted(cpcs[0]) == ["b"] and sorted(cpcs[1]) == ["a"] and cpcs[2] == []
    ddf = pbn.DynamicDataFrame(table(300, 1)[["a", "b"]], 1)
    dyn = pbn.DynamicKMutualInformation(ddf, 3, seed=0, samples=10)
    assert dyn.static_tests().num_variables() == 2 and dyn.transition_tests().num_variables() == 4
    with pytest.raises(ValueError, match="Wrong data type"):
        pbn.KMutualInformation(pd.DataFrame({"a": [1.0, 2.0, 3.0], "b": pd.Categorical(["x", "y", "x"])}), 1)


def test_window_form_gives_the_all_pairs_counts(pbn, monkeypatch):
    """Round 6: tables of at least 32 768 rows take the sorted-window walk (kmi.hip: kmi_window_kernel) instead of the all-pairs kernels.  Both
    produce the reference's integer counts (k-th neighbour distance, strictly-inside counts of the subspaces), the host adds the same digammas
    in the same row order: the estimates are EQUAL, not close - checked on small tables with the window form forced (against the all-pairs form
    and the oracle, every overload and the permutation p-values) and on a table that takes it by default (against the all-pairs form forced)."""
    from oracle import oracle

    cases = [("a", "b", []), ("a", "d", []), ("a", "c", ["b"]), ("a", "e", ["d"]), ("b", "c", ["a", "d"]), ("a", "e", ["b", "c", "d"])]

    def values(df, k, **kw):
        t = pbn.KMutualInformation(df, k, seed=0, **kw)
        return [t.mi(x, y, z if len(z) != 1 else z[0]) for x, y, z in cases]

    for n, k in ((257, 3), (1200, 10), (5000, 64)):
        df = table(n, 4)
        plain = values(df, k)
        monkeypatch.setenv("PBN_KMI_WINDOW_MIN_ROWS", "0")
        forced = values(df, k)
        monkeypatch.delenv("PBN_KMI_WINDOW_MIN_ROWS")
        assert forced == plain, (n, k)
        if n <= 1200:
            for (x, y, z), got in zip(cases, forced):
                want, _ = oracle.kmi(df[[x, y] + z].to_numpy(dtype=np.float64), k)
                assert got == pytest.approx(want, rel=1e-10, abs=1e-12)
    # permutation p-values: the permuted x column goes through the same walk
    df = table(400, 7)
    monkeypatch.setenv("PBN_KMI_WINDOW_MIN_ROWS", "0")
    test = pbn.KMutualInformation(df, 4, seed=3, shuffle_neighbors=5, samples=40)
    for x, y, z in (("a", "d", []), ("a", "c", ["b"]), ("d", "b", ["a", "c"])):
        _, want = oracle.kmi(df[[x, y] + z].to_numpy(), 4, 3, 5, 40)
        assert test.pvalue(x, y, z or None) == want
    monkeypatch.delenv("PBN_KMI_WINDOW_MIN_ROWS")
    # a table that takes the window form by default
    df = table(40_000, 9)
    window = values(df, 10)
    monkeypatch.setenv("PBN_KMI_WINDOW_MIN_ROWS", "1000000000")
    assert values(df, 10) == window
