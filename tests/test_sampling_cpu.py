"""CPU tier: the host-only sampling entry points (pbn_lg_sample, pbn_discrete_sample) and BayesianNetwork.sample over
factors with given parameters, bit-exact against the oracle's restatement of LinearGaussianCPD::sample
(LinearGaussianCPD.cpp:317-380) and DiscreteFactor::sample_indices (DiscreteFactor.hpp:144-205) - same libstdc++
mt19937 streams.  Known answers: the first normal deviates of std::mt19937{0} are pinned below."""
import numpy as np
import pandas as pd
import pyarrow as pa
import pytest

import pybnesian_amd as pbn
from oracle import oracle


def test_lg_sample_matches_oracle(ensure_built):
    rng = np.random.default_rng(0)
    ev = pd.DataFrame({"b": rng.normal(size=100), "c": rng.normal(size=100)})
    for seed in (0, 7, 2 ** 32 - 1):
        cpd = pbn.LinearGaussianCPD("a", ["b", "c"], [0.5, -1.25, 2.0], 0.3)
        got = cpd.sample(100, ev, seed).to_numpy()
        want = oracle.lg_sample(100, [0.5, -1.25, 2.0], 0.3, seed, ev.to_numpy())
        assert np.array_equal(got, want)
    root = pbn.LinearGaussianCPD("a", [], [1.0], 4.0)
    got = root.sample(5, None, 0).to_numpy()
    assert np.array_equal(got, oracle.lg_sample(5, [1.0], 4.0, 0))
    # regression anchor: libstdc++ std::normal_distribution<double> (polar method) over std::mt19937{0}, as produced
    # by this container's libstdc++ (the stream a reference build against the same libstdc++ draws)
    z = (got - 1.0) / 2.0
    assert np.allclose(z[:4], [1.12279494, 0.30280522, 0.07085924, 0.0730414], atol=1e-8)
    assert root.sample(0, None, 0).to_numpy().size == 0
    with pytest.raises(ValueError, match="non-negative"):
        root.sample(-1, None, 0)
    with pytest.raises(ValueError, match="Evidence values not present"):
        pbn.LinearGaussianCPD("a", ["b"], [0.0, 1.0], 1.0).sample(3, None, 0)
    # float32 evidence is accepted (LinearGaussianCPD.cpp:355-363)
    ev32 = ev.astype("float32")
    got32 = pbn.LinearGaussianCPD("a", ["b", "c"], [0.5, -1.25, 2.0], 0.3).sample(100, ev32, 3).to_numpy()
    want32 = oracle.lg_sample(100, [0.5, -1.25, 2.0], 0.3, 3, ev32.to_numpy().astype(np.float64))
    assert np.array_equal(got32, want32)


def test_discrete_sample_matches_oracle(ensure_built):
    rng = np.random.default_rng(1)
    n = 500
    df = pd.DataFrame({"x": pd.Categorical(rng.choice(["u", "v", "w"], size=n, p=[0.2, 0.5, 0.3])),
                       "y": pd.Categorical(rng.choice(["p", "q"], size=n)),
                       "z": pd.Categorical(rng.choice(["k", "l", "m", "n"], size=n))})
    f = pbn.DiscreteFactor("x", ["y", "z"])
    f.fit(df)
    ev = df[["y", "z"]].iloc[:200]
    got = f.sample(200, ev, 11)
    assert pa.types.is_dictionary(got.type) and got.dictionary.to_pylist() == ["u", "v", "w"]
    off = (ev["y"].cat.codes.to_numpy().astype(np.int64) * 3 + ev["z"].cat.codes.to_numpy().astype(np.int64) * 6).astype(np.int32)
    want = oracle.discrete_sample(200, f._logprob, 3, 11, off)
    assert np.array_equal(got.indices.to_numpy(), want)
    marg = pbn.DiscreteFactor("x", [])
    marg.fit(df)
    got = marg.sample(4000, None, 5).indices.to_numpy()
    assert np.array_equal(got, oracle.discrete_sample(4000, marg._logprob, 3, 5))
    freq = np.bincount(got, minlength=3) / 4000.0
    assert np.allclose(freq, np.exp(marg._logprob), atol=0.03)
    with pytest.raises(ValueError, match="do not have"):
        f.sample(10, ev, 0)


def test_network_sample_gaussian(ensure_built):
    model = pbn.GaussianNetwork(["a", "b", "c"], [("a", "b"), ("a", "c"), ("b", "c")])
    cpds = [pbn.LinearGaussianCPD("a", [], [2.0], 1.0), pbn.LinearGaussianCPD("b", ["a"], [0.0, 1.5], 0.25),
            pbn.LinearGaussianCPD("c", ["a", "b"], [-1.0, 0.5, -2.0], 0.5)]
    model.add_cpds(cpds)
    assert model.fitted() and model.topological_sort() == ["a", "b", "c"]
    s = model.sample(20000, 4)
    assert s.schema.names == ["a", "b", "c"] and s.num_rows == 20000
    a, b, c = (s.column(i).to_numpy() for i in range(3))
    # node i of the topological order is sampled with seed + i (BayesianNetwork.hpp:1041-1046)
    assert np.array_equal(a, oracle.lg_sample(20000, [2.0], 1.0, 4))
    assert np.array_equal(b, oracle.lg_sample(20000, [0.0, 1.5], 0.25, 5, a[:, None]))
    assert np.array_equal(c, oracle.lg_sample(20000, [-1.0, 0.5, -2.0], 0.5, 6, np.column_stack([a, b])))
    assert abs(a.mean() - 2.0) < 0.03 and abs(np.mean(b - 1.5 * a)) < 0.02 and abs(np.var(c + 1.0 - 0.5 * a + 2.0 * b) - 0.5) < 0.03
    rev = pbn.GaussianNetwork(["c", "b", "a"], [("a", "b"), ("a", "c"), ("b", "c")])
    rev.add_cpds(cpds)
    so = rev.sample(10, 4, ordered=True)
    assert so.schema.names == ["c", "b", "a"] and np.array_equal(so.column(2).to_numpy(), a[:10])
    with pytest.raises(ValueError, match="parent set"):
        model.add_cpds([pbn.LinearGaussianCPD("b", [], [0.0], 1.0)])
    with pytest.raises(ValueError, match="not fitted"):
        pbn.GaussianNetwork(["a"]).sample(3, 0)


def test_network_pickle_roundtrip(ensure_built, tmp_path):
    model = pbn.SemiparametricBN(["a", "b"], [("a", "b")], [("b", pbn.CKDEType())])
    model.add_cpds([pbn.LinearGaussianCPD("a", [], [0.0], 1.0)])
    model.save(str(tmp_path / "m"))
    back = pbn.load(str(tmp_path / "m.pickle"))
    assert back.nodes() == ["a", "b"] and back.arcs() == [("a", "b")] and back.node_type("b") == pbn.CKDEType()
    assert not back.fitted()
    g = pbn.GaussianNetwork(["a", "b"], [("a", "b")])
    g.add_cpds([pbn.LinearGaussianCPD("a", [], [0.0], 1.0), pbn.LinearGaussianCPD("b", ["a"], [1.0, 2.0], 0.5)])
    g.save(str(tmp_path / "g"), include_cpd=True)
    back = pbn.load(str(tmp_path / "g.pickle"))
    assert back.fitted() and np.array_equal(back.cpd("b").beta, [1.0, 2.0]) and back.cpd("b").variance == 0.5
    assert np.array_equal(back.sample(50, 1).column(1).to_numpy(), g.sample(50, 1).column(1).to_numpy())


def test_dynamic_dataframe_layout(ensure_built):
    """DynamicDataFrame (dataset/dynamic_dataset.cpp:16-85): slice i of order m holds rows [m - i, m - i + N - m) under the
    name v_t_i; the static table of order m has N - m + 1 rows and the slices 1..m."""
    n = 12
    df = pd.DataFrame({"a": np.arange(n, dtype=float), "b": np.arange(n, dtype=float) * 10})
    d1 = pbn.DynamicDataFrame(df, 1)
    assert d1.static_df().schema.names == ["a_t_1", "b_t_1"] and d1.static_df().num_rows == n
    t1 = d1.transition_df()
    assert t1.schema.names == ["a_t_0", "b_t_0", "a_t_1", "b_t_1"] and t1.num_rows == n - 1
    assert np.array_equal(t1.column(0).to_numpy(), np.arange(1, n)) and np.array_equal(t1.column(2).to_numpy(), np.arange(0, n - 1))
    d3 = pbn.DynamicDataFrame(df, 3)
    t3 = d3.transition_df()
    assert t3.schema.names == [f"{v}_t_{i}" for i in range(4) for v in ("a", "b")] and t3.num_rows == n - 3
    for i in range(4):
        assert np.array_equal(t3.column(2 * i).to_numpy(), np.arange(3 - i, n - i))
    s3 = d3.static_df()
    assert s3.schema.names == [f"{v}_t_{i}" for i in (1, 2, 3) for v in ("a", "b")] and s3.num_rows == n - 2
    assert np.array_equal(s3.column(0).to_numpy(), np.arange(2, n)) and np.array_equal(s3.column(4).to_numpy(), np.arange(0, n - 2))
    assert d3.temporal_slice(2).schema.names == ["a_t_2", "b_t_2"] and d3.num_rows() == n - 3 and d3.num_variables() == 2
    with pytest.raises(ValueError, match="at least 1"):
        pbn.DynamicDataFrame(df, 0)
    from pybnesian_amd.dynamic import static_blacklist, temporal_names

    assert temporal_names(["a", "b"], 1, 2) == ["a_t_1", "a_t_2", "b_t_1", "b_t_2"]
    assert static_blacklist(["a", "b"], 1) == []
    assert static_blacklist(["a", "b"], 2) == [("a_t_1", "a_t_2"), ("a_t_1", "b_t_2"), ("b_t_1", "a_t_2"), ("b_t_1", "b_t_2")]
    dbn = pbn.DynamicBayesianNetwork(["a", "b"], 2)
    assert dbn.static_bn().nodes() == ["a_t_1", "a_t_2", "b_t_1", "b_t_2"]
    assert dbn.transition_bn().nodes() == ["a_t_0", "b_t_0"] and dbn.transition_bn().interface_nodes() == dbn.static_bn().nodes()


def test_dynamic_network_create_and_variables(ensure_built):
    """/root/reference/tests/models/DynamicBayesianNetwork_test.py:12-76 re-typed (creation, type checks, add / remove
    variable)."""
    variables = ["a", "b", "c", "d"]
    gbn = pbn.DynamicGaussianNetwork(variables, 2)
    assert gbn.markovian_order() == 2 and gbn.variables() == variables and gbn.num_variables() == 4
    assert gbn.type() == pbn.GaussianNetworkType()
    transition_nodes = [v + "_t_0" for v in variables]
    static_nodes = [v + "_t_" + str(m) for v in variables for m in range(1, 3)]
    assert set(gbn.static_bn().nodes()) == set(static_nodes)
    assert set(gbn.transition_bn().interface_nodes()) == set(static_nodes)
    assert set(gbn.transition_bn().nodes()) == set(transition_nodes)
    static_bn = pbn.GaussianNetwork(static_nodes)
    transition_bn = pbn.ConditionalGaussianNetwork(transition_nodes, static_nodes)
    pbn.DynamicGaussianNetwork(variables, 2, static_bn, transition_bn)
    wrong_transition = pbn.ConditionalKDENetwork(transition_nodes, static_nodes)
    with pytest.raises(ValueError, match="Static and transition Bayesian networks do not have the same type"):
        pbn.DynamicGaussianNetwork(variables, 2, static_bn, wrong_transition)
    with pytest.raises(ValueError, match="Bayesian networks are not Gaussian."):
        pbn.DynamicGaussianNetwork(variables, 2, pbn.KDENetwork(static_nodes), wrong_transition)
    assert all(gbn.contains_variable(v) for v in variables)
    gbn.add_variable("e")
    assert set(gbn.variables()) == set(variables + ["e"]) and gbn.num_variables() == 5
    assert set(gbn.static_bn().nodes()) == set(v + "_t_" + str(m) for v in variables + ["e"] for m in range(1, 3))
    assert set(gbn.transition_bn().nodes()) == set(v + "_t_0" for v in variables + ["e"])
    gbn.remove_variable("b")
    left = ["a", "c", "d", "e"]
    assert set(gbn.variables()) == set(left) and gbn.num_variables() == 4
    assert set(gbn.static_bn().nodes()) == set(v + "_t_" + str(m) for v in left for m in range(1, 3))
    assert set(gbn.transition_bn().nodes()) == set(v + "_t_0" for v in left)
    assert set(gbn.transition_bn().interface_nodes()) == set(gbn.static_bn().nodes())
