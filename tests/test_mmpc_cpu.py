"""CPU tier: the C++ MMPC (pbn_mmpc_cpcs) against the Python restatement of learning/algorithms/mmpc.cpp
(oracle/mmpc_oracle.py), both driven by the SAME deterministic p-value function through the Python-derived
IndependenceTest path.  P-values are quantised so that exact ties (and exact zeros) are common: CPC contents AND their
libstdc++ iteration order AND the number of tests must match."""
import numpy as np
import pytest

import pybnesian_amd as pbn
from oracle import mmpc_oracle
from pybnesian_amd.independences import mmpc_cpcs, validate_restrictions


class TableTest(pbn.IndependenceTest):
    """p-value of x _||_ y | z from partial correlations of a fixed random covariance, rounded to create ties."""

    def __init__(self, n, seed, rows=300, decimals=3):
        rng = np.random.default_rng(seed)
        B = np.triu(rng.uniform(-1, 1, (n, n)) * (rng.random((n, n)) < 0.35), 1)
        A = np.linalg.inv(np.eye(n) - B)
        self.cov = A.T @ np.diag(rng.uniform(0.5, 1.5, n)) @ A
        self.rows, self.decimals = rows, decimals
        self.names = [f"v{i}" for i in range(n)]
        self.idx = {v: i for i, v in enumerate(self.names)}
        self.calls = 0

    def by_index(self, a, b, cond):
        self.calls += 1
        p = mmpc_oracle.lincor_pvalue(self.cov, self.rows, a, b, cond)
        return float(np.round(p, self.decimals))

    def pvalue(self, x, y, z=None):
        cond = [] if z is None else ([z] if isinstance(z, str) else list(z))
        return self.by_index(self.idx[x], self.idx[y], [self.idx[c] for c in cond])

    def variable_names(self):
        return list(self.names)


@pytest.mark.parametrize("n,seed,rows", [(6, 0, 300), (9, 1, 300), (12, 2, 2000), (14, 5, 100000), (10, 7, 60)])
def test_mmpc_matches_restatement(ensure_built, n, seed, rows):
    t1, t2 = TableTest(n, seed, rows), TableTest(n, seed, rows)
    for symmetric in (False, True):
        got, ntests = mmpc_cpcs(t1, t1.names, 0.05, symmetric=symmetric)
        want, calls = mmpc_oracle.mmpc_all_variables(t2.by_index, n, 0.05, symmetric=symmetric)
        assert [[t1.idx[v] for v in c] for c in got] == want     # same members in the same set iteration order
        assert ntests == calls
    assert any(len(c) > 1 for c in want)


def test_mmpc_restrictions(ensure_built):
    n = 8
    t1, t2 = TableTest(n, 3), TableTest(n, 3)
    names = t1.names
    a_bl, a_wl, e_bl, e_wl = validate_restrictions(names, arc_blacklist=[("v0", "v1"), ("v1", "v0"), ("v2", "v3")],
                                                   edge_blacklist=[("v4", "v5")])
    assert (0, 1) in e_bl and (4, 5) in e_bl and a_bl == [(2, 3)] and not a_wl and not e_wl
    got, ntests = mmpc_cpcs(t1, names, 0.05, a_wl, e_bl, e_wl)
    want, calls = mmpc_oracle.mmpc_all_variables(t2.by_index, n, 0.05, a_wl, e_bl, e_wl)
    assert [[t1.idx[v] for v in c] for c in got] == want and ntests == calls
    assert "v1" not in got[0] and "v5" not in got[4]
    # whitelisted edges stay in the CPC (the reference does not terminate on this path, mmpc.cpp:357-382)
    _, a_wl, e_bl, e_wl = validate_restrictions(names, edge_whitelist=[("v0", "v7")], arc_whitelist=[("v6", "v2")])
    got, _ = mmpc_cpcs(t1, names, 0.05, a_wl, e_bl, e_wl)
    want, _ = mmpc_oracle.mmpc_all_variables(t2.by_index, n, 0.05, a_wl, e_bl, e_wl)
    assert [[t1.idx[v] for v in c] for c in got] == want
    assert "v7" in got[0] and "v0" in got[7] and "v2" in got[6] and "v6" in got[2]
    with pytest.raises(ValueError, match="in blacklist and whitelist"):
        validate_restrictions(names, edge_blacklist=[("v0", "v1")], edge_whitelist=[("v1", "v0")])
    with pytest.raises(ValueError, match="not present"):
        validate_restrictions(names, arc_blacklist=[("v0", "zz")])


def test_mmpc_errors(ensure_built):
    class Broken(pbn.IndependenceTest):
        def variable_names(self):
            return ["a", "b", "c"]

        def pvalue(self, x, y, z=None):
            raise KeyError("no data")

    with pytest.raises(KeyError, match="no data"):
        mmpc_cpcs(Broken(), ["a", "b", "c"], 0.05)
    t = TableTest(4, 0)
    with pytest.raises(ValueError, match="alpha"):
        mmpc_cpcs(t, t.names, 1.5)


@pytest.mark.parametrize("rows", [12, 300, 20000, 3000000])
def test_linear_correlation_from_covariance(ensure_built, rows):
    """Host-only LinearCorrelation over a given covariance: partial correlations (Jacobi eigen-solver + pseudo-inverse)
    and the two-sided Student-t tail (continued fraction, large-df Gamma ratio) against numpy eigh + scipy, from
    p ~ 1 down to exact 0; then the native (no Python frame) MMPC path against the restatement."""
    rng = np.random.default_rng(rows)
    n = 10
    B = np.triu(rng.uniform(-1.2, 1.2, (n, n)) * (rng.random((n, n)) < 0.4), 1)
    A = np.linalg.inv(np.eye(n) - B)
    cov = A.T @ np.diag(rng.uniform(0.3, 2.0, n)) @ A
    names = [f"v{i}" for i in range(n)]
    test = pbn.LinearCorrelation.from_covariance(names, cov, rows)
    for _ in range(300):
        k = int(rng.integers(0, 7))
        if rows - 4 - k < 1:
            continue
        sel = [int(i) for i in rng.choice(n, size=k + 2, replace=False)]
        got = test.pvalue(names[sel[0]], names[sel[1]], [names[i] for i in sel[2:]])
        want = mmpc_oracle.lincor_pvalue(cov, rows, sel[0], sel[1], sel[2:])
        assert got == pytest.approx(want, rel=5e-9, abs=1e-305), (sel, got, want)
    got, ntests = mmpc_cpcs(test, names, 0.01)
    want, calls = mmpc_oracle.mmpc_all_variables(lambda a, b, c: mmpc_oracle.lincor_pvalue(cov, rows, a, b, c), n, 0.01)
    assert [[names.index(v) for v in c] for c in got] == want and ntests == calls


@pytest.mark.parametrize("n,ni,seed", [(6, 2, 0), (8, 4, 3), (9, 1, 5)])
def test_conditional_mmpc_matches_restatement(ensure_built, n, ni, seed):
    """mmpc_all_variables over a conditional graph (mmpc.cpp:875-908, 740-784): interface nodes are candidates of the
    nodes only."""
    t1, t2 = TableTest(n + ni, seed, 400), TableTest(n + ni, seed, 400)
    names = t1.names
    got, ntests = mmpc_cpcs(t1, names[:n], 0.05, interface_nodes=names[n:])
    want, calls = mmpc_oracle.mmpc_all_variables(t2.by_index, n + ni, 0.05, n_interface=ni)
    assert [[t1.idx[v] for v in c] for c in got] == want and ntests == calls
    for i in range(n, n + ni):
        assert all(v < n for v in want[i])           # no interface - interface candidates
    assert any(any(v >= n for v in want[i]) for i in range(n))
