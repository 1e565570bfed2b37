"""GPU tier: the halves of the reference's extension-point / serialization tests that fit factors or run the hill-climb
(/root/reference/tests/models/BayesianNetwork_type_test.py:118-171, serialization/serialize_factor_test.py:150-241,
serialize_models_test.py:196-307, 685-835).  The user-defined types are the ones of test_reference_types_cpu.py."""
import pickle

import numpy as np
import pandas as pd
import pytest

from test_reference_types_cpu import (ConditionalNewBN, ConditionalOtherBN, DynamicOtherBN, MyRestrictedGaussianNetworkType, NewBN,
                                      NonHomogeneousType, OtherBN, roundtrip)

pytestmark = [pytest.mark.gpu, pytest.mark.extra]
ABCD = ["a", "b", "c", "d"]


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def normal_indep(n, seed=0):
    rng = np.random.RandomState(seed)
    return pd.DataFrame({"a": rng.normal(3, 0.5, n), "b": rng.normal(2.5, 2, n), "c": rng.normal(-4.2, 0.75, n), "d": rng.normal(1.5, 1.5, n)})


def discrete_dependent(n, seed=0):
    rng = np.random.RandomState(seed)
    a = rng.choice(["a1", "a2"], n, p=[0.75, 0.25])
    b = np.where(a == "a1", rng.choice(["b1", "b2", "b3"], n, p=[0.33, 0.33, 0.34]), rng.choice(["b1", "b2", "b3"], n, p=[0.0, 0.4, 0.6]))
    c = np.where(b == "b1", rng.choice(["c1", "c2"], n, p=[0.5, 0.5]), rng.choice(["c1", "c2"], n, p=[0.2, 0.8]))
    d = rng.choice(["d1", "d2", "d3", "d4"], n)
    return pd.DataFrame({"a": a, "b": b, "c": c, "d": d}, dtype="category")


def test_new_specific_bn_type_hill_climb(pbn):   # BayesianNetwork_type_test.py:137-171
    df = normal_indep(1000)
    df["b"] += 0.8 * df["a"]
    df["d"] += 0.5 * df["c"] - 0.7 * df["a"]
    bic = pbn.BIC(df)
    start = NewBN(ABCD)
    hc = pbn.GreedyHillClimbing()
    estimated = hc.estimate(pbn.ArcOperatorSet(), bic, start)
    assert estimated.type() == start.type() and type(estimated) is NewBN
    assert estimated.num_arcs() >= 2 and all("a" in s for s, t in estimated.arcs())
    # without the user type's restriction the same search also links c and d
    free = hc.estimate(pbn.ArcOperatorSet(), bic, pbn.GaussianNetwork(ABCD))
    assert any({s, t} == {"c", "d"} for s, t in free.arcs())

    cstart = ConditionalNewBN(["a", "c"], ["b", "d"])
    cestimated = hc.estimate(pbn.ArcOperatorSet(), bic, cstart)
    assert cestimated.type() == cstart.type() and type(cestimated) is ConditionalNewBN
    assert cestimated.interface_nodes() == ["b", "d"]
    assert all("a" in s for s, t in cestimated.arcs())


def test_serialization_fitted_factor(pbn):   # serialize_factor_test.py:150-241
    rng = np.random.RandomState(1)
    data = pd.DataFrame({"a": rng.rand(10), "b": rng.rand(10), "c": rng.rand(10)}).astype(float)
    ckde = pbn.CKDE("c", ["a", "b"])
    ckde.fit(data)
    loaded = roundtrip(ckde)
    assert loaded.variable() == "c" and set(loaded.evidence()) == {"a", "b"} and loaded.fitted()
    assert loaded.type() == pbn.CKDEType() and loaded.num_instances() == 10
    tr = loaded.kde_joint().dataset().to_pandas()
    for col in "abc":
        assert np.all(tr[col] == data[col])
    np.testing.assert_allclose(loaded.logl(data), ckde.logl(data), rtol=1e-12)

    discrete = pbn.DiscreteFactor("c", ["a", "b"])
    train = pd.DataFrame({"a": ["a1", "a2", "a1", "a2", "a2", "a2", "a2", "a2"], "b": ["b1", "b1", "b1", "b1", "b1", "b2", "b1", "b2"],
                          "c": ["c1", "c1", "c1", "c1", "c2", "c2", "c2", "c2"]}, dtype="category")
    discrete.fit(train)
    loaded = roundtrip(discrete)
    assert loaded.variable() == "c" and set(loaded.evidence()) == {"a", "b"} and loaded.fitted()
    assert loaded.type() == pbn.DiscreteFactorType()
    test = pd.DataFrame({"a": ["a1", "a2", "a1", "a2", "a1", "a2", "a1", "a2"], "b": ["b1", "b1", "b2", "b2", "b1", "b1", "b2", "b2"],
                         "c": ["c1", "c1", "c1", "c1", "c2", "c2", "c2", "c2"]}, dtype="category")
    assert list(np.exp(loaded.logl(test))) == [1, 0.5, 0.5, 0, 0, 0.5, 0.5, 1]


def test_serialization_fitted_bn_mixed_factors(pbn):   # serialize_models_test.py:196-307
    other = OtherBN(ABCD, [("a", "b")], [("b", pbn.LinearGaussianCPDType()), ("c", pbn.CKDEType()), ("d", pbn.DiscreteFactorType())])
    cpd_c = pbn.CKDE("c", [])
    cpd_c.fit(normal_indep(100))
    cpd_d = pbn.DiscreteFactor("d", [])
    cpd_d.fit(discrete_dependent(100))
    other.add_cpds([pbn.LinearGaussianCPD("a", [], [0], 0.5), pbn.LinearGaussianCPD("b", ["a"], [1, 2], 2), cpd_c, cpd_d])
    other.include_cpd = True
    loaded = roundtrip(other)
    assert loaded.fitted() and type(loaded) is OtherBN and loaded.extra_info == "extra"
    assert loaded.cpd("a").beta == [0] and loaded.cpd("a").variance == 0.5 and loaded.cpd("a").type() == pbn.LinearGaussianCPDType()
    assert list(loaded.cpd("b").beta) == [1, 2] and loaded.cpd("b").evidence() == ["a"]
    c = loaded.cpd("c")
    assert c.evidence() == [] and c.fitted() and c.num_instances() == 100 and c.type() == pbn.CKDEType()
    d = loaded.cpd("d")
    assert d.evidence() == [] and d.fitted() and d.type() == pbn.DiscreteFactorType()
    assert loaded.node_type("a") == pbn.LinearGaussianCPDType()   # add_cpds resolved the unknown type


def test_serialization_fitted_dbn(pbn):   # serialize_models_test.py:685-835
    gaussian = pbn.DynamicGaussianNetwork(ABCD, 2)
    gaussian.static_bn().add_arc("a_t_2", "d_t_1")
    gaussian.transition_bn().add_arc("c_t_2", "b_t_0")
    df = normal_indep(1000)
    gaussian.fit(df)
    assert not roundtrip(gaussian).fitted()
    gaussian.include_cpd = True
    loaded = roundtrip(gaussian)
    assert loaded.fitted() and loaded.static_bn().fitted() and loaded.transition_bn().fitted()
    np.testing.assert_allclose(loaded.logl(df.iloc[:50]), gaussian.logl(df.iloc[:50]), rtol=1e-12)

    static_nodes = [v + "_t_" + str(m) for v in ABCD for m in range(1, 3)]
    transition_nodes = [v + "_t_0" for v in ABCD]
    other_static = OtherBN(static_nodes, [("a_t_2", "d_t_1")], [("b_t_2", pbn.DiscreteFactorType()), ("b_t_1", pbn.DiscreteFactorType()),
                                                                ("c_t_1", pbn.CKDEType()), ("d_t_1", pbn.LinearGaussianCPDType())])
    other_static.add_cpds([pbn.LinearGaussianCPD("d_t_1", ["a_t_2"], [1, 2], 2)])
    other_transition = ConditionalOtherBN(transition_nodes, static_nodes, [("a_t_2", "d_t_0")],
                                          [("b_t_0", pbn.DiscreteFactorType()), ("c_t_0", pbn.CKDEType()), ("d_t_0", pbn.LinearGaussianCPDType())])
    other_transition.add_cpds([pbn.LinearGaussianCPD("d_t_0", ["a_t_2"], [3, 4], 1.5)])
    dyn_other = DynamicOtherBN(ABCD, 2, other_static, other_transition)
    mixed = normal_indep(1000)
    mixed["b"] = discrete_dependent(1000)["b"]
    dyn_other.fit(mixed)
    dyn_other.include_cpd = True
    loaded = roundtrip(dyn_other)
    assert loaded.fitted() and loaded.static_bn().fitted() and loaded.transition_bn().fitted() and loaded.extra_info == "extra"
    assert loaded.type() == NonHomogeneousType()
    assert loaded.static_bn().node_type("b_t_1") == pbn.DiscreteFactorType() and loaded.static_bn().node_type("c_t_1") == pbn.CKDEType()
    assert loaded.transition_bn().node_type("c_t_0") == pbn.CKDEType()
    # the factors given through add_cpds were already fitted and are kept by fit() (BayesianNetwork.hpp:960-994)
    cpd = loaded.static_bn().cpd("d_t_1")
    assert cpd.evidence() == ["a_t_2"] and list(cpd.beta) == [1, 2] and cpd.variance == 2
    cpd = loaded.transition_bn().cpd("d_t_0")
    assert list(cpd.beta) == [3, 4] and cpd.variance == 1.5
    assert loaded.static_bn().cpd("a_t_1").type() == pbn.LinearGaussianCPDType()   # unknown -> the user type's default


def test_homogeneous_and_heterogeneous_networks_fit(pbn):
    """HomogeneousBN / HeterogeneousBN (models/HomogeneousBN.hpp, HeterogeneousBN.hpp) through fit and the hill-climb."""
    import pyarrow as pa

    df = normal_indep(600)
    df["b"] += 0.8 * df["a"]
    hom = pbn.HomogeneousBN(pbn.CKDEType(), ABCD, [("a", "b")])
    hom.fit(df)
    kde = pbn.KDENetwork(ABCD, [("a", "b")])
    kde.fit(df)
    np.testing.assert_allclose(hom.slogl(df), kde.slogl(df), rtol=1e-12)
    assert all(hom.cpd(v).type() == pbn.CKDEType() for v in ABCD)

    het = pbn.HeterogeneousBN({pa.float64(): [pbn.CKDEType(), pbn.LinearGaussianCPDType()]}, ABCD, [("a", "b")])
    assert het.has_unknown_node_types()
    het.fit(df)
    assert all(het.node_type(v) == pbn.CKDEType() for v in ABCD)
    np.testing.assert_allclose(het.slogl(df), kde.slogl(df), rtol=1e-12)
    het2 = pbn.HeterogeneousBN([pbn.LinearGaussianCPDType(), pbn.CKDEType()], ABCD, [("a", "b")])
    het2.set_unknown_node_types(df, [("c", pbn.LinearGaussianCPDType())])
    assert het2.node_type("a") == pbn.LinearGaussianCPDType() and het2.node_type("c") == pbn.CKDEType()

    learned = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), pbn.BIC(df), pbn.HomogeneousBN(pbn.LinearGaussianCPDType(), ABCD))
    ref = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), pbn.BIC(df), pbn.GaussianNetwork(ABCD))
    assert type(learned) is pbn.HomogeneousBN and sorted(learned.arcs()) == sorted(ref.arcs())


class ExtraNewBN(NewBN):
    """hillclimbing_test.py:180-195: a Python-derived network with extra pickled state."""

    def __init__(self, variables, arcs=None):
        NewBN.__init__(self, variables, arcs)
        self.extra_data = "extra"

    def __getstate_extra__(self):
        return self.extra_data

    def __setstate_extra__(self, extra):
        self.extra_data = extra


def test_hc_conditional_and_removed_nodes(pbn, golden):   # hillclimbing_test.py:60-101
    from helpers import frame

    df = frame(golden["train10k"])
    bic = pbn.BIC(df)
    names = list(df.columns)
    start = pbn.ConditionalGaussianNetwork(names[2:], names[:2])
    nodes, interface = names[2:], names[:2]
    nodes.insert(1, "e")
    interface.insert(1, "f")
    start_removed = pbn.ConditionalGaussianNetwork(nodes, interface)
    start_removed.remove_node("e")
    start_removed.remove_interface_node("f")
    arc_set, hc = pbn.ArcOperatorSet(), pbn.GreedyHillClimbing()
    res = hc.estimate(arc_set, bic, start, max_iters=1)
    assert res.num_arcs() == 1 and type(res) is pbn.ConditionalGaussianNetwork
    added = res.arcs()[0]
    op_delta = bic.score(res) - bic.score(start)
    res_removed = hc.estimate(arc_set, bic, start_removed, max_iters=1)
    assert res_removed.num_arcs() == 1
    added_removed = res_removed.arcs()[0]
    assert added == added_removed or added == added_removed[::-1]
    assert np.isclose(op_delta, bic.score(res_removed) - bic.score(start_removed))
    assert np.isclose(op_delta, bic.local_score(res, added[1], [added[0]]) - bic.local_score(res, added[1], []))
    assert hc.estimate(arc_set, bic, start, epsilon=op_delta + 0.01).num_arcs() == start.num_arcs()
    assert hc.estimate(arc_set, bic, start_removed, epsilon=op_delta + 0.01).num_arcs() == 0
    full = hc.estimate(arc_set, bic, start)
    assert full.num_arcs() > 1 and all(not full.is_interface(t) for _, t in full.arcs())
    full_removed = hc.estimate(arc_set, bic, start_removed)
    assert sorted(full_removed.arcs()) == sorted(full.arcs())


def test_hc_validated_and_shortcut(pbn, golden):   # hillclimbing_test.py:103-205
    from helpers import frame

    df = frame(golden["train10k"])
    names = list(df.columns)
    start = pbn.GaussianNetwork(names)
    names.insert(1, "e")
    names.insert(4, "f")
    start_removed = pbn.GaussianNetwork(names)
    start_removed.remove_node("e")
    start_removed.remove_node("f")
    vl = pbn.ValidatedLikelihood(df, seed=0)
    arc_set, hc = pbn.ArcOperatorSet(), pbn.GreedyHillClimbing()
    res = hc.estimate(arc_set, vl, start, max_iters=1)
    assert res.num_arcs() == 1
    added = res.arcs()[0]
    op_delta = vl.cv_lik.score(res) - vl.cv_lik.score(start)
    res_removed = hc.estimate(arc_set, vl, start_removed, max_iters=1)
    added_removed = res_removed.arcs()[0]
    assert added == added_removed or added == added_removed[::-1]
    assert np.isclose(op_delta, vl.cv_lik.score(res_removed) - vl.cv_lik.score(start_removed))
    assert np.isclose(op_delta, vl.cv_lik.local_score(res, added[1], [added[0]]) - vl.cv_lik.local_score(res, added[1], []))
    # CV is score equivalent for Gaussian networks: blacklisting the added arc adds its reverse
    res = hc.estimate(arc_set, vl, start, max_iters=1, arc_blacklist=[added])
    assert res.num_arcs() == 1 and res.arcs()[0][::-1] == added
    assert hc.estimate(arc_set, vl, start, epsilon=op_delta + 0.01).num_arcs() == 0
    hc.estimate(arc_set, vl, start)
    hc.estimate(arc_set, vl, start_removed)

    model = pbn.hc(df, bn_type=pbn.GaussianNetworkType())
    assert type(model) == pbn.GaussianNetwork
    model = pbn.hc(df, bn_type=MyRestrictedGaussianNetworkType(), score="bic", operators=["arcs"])
    assert type(model) == NewBN and all("a" in s for s, _ in model.arcs())

    estimated = hc.estimate(arc_set, pbn.BIC(df), ExtraNewBN(ABCD))
    assert type(estimated) is ExtraNewBN and estimated.extra_data == "extra"


def test_bde_score_and_discrete_hill_climb(pbn):
    """learning/scores/bde.cpp:5-50 against a direct restatement on pandas counts; the hill-climb of a DiscreteBN driven
    by it (counts from the device groupings, per-candidate Python score calls as for any derived Score)."""
    from math import lgamma

    df = discrete_dependent(5000, 3)
    bde = pbn.BDe(df, iss=2.0)
    model = pbn.DiscreteBN(ABCD)
    assert bde.compatible_bn(model) and not bde.compatible_bn(pbn.GaussianNetwork(ABCD)) and bde.has_variables(["a", "d"])

    def restated(variable, parents, iss=2.0):
        cards = {c: len(df[c].cat.categories) for c in df.columns}
        total = int(np.prod([cards[v] for v in [variable] + parents]))
        alpha = iss / total
        res = -total * lgamma(alpha)
        if not parents:
            counts = df[variable].value_counts().reindex(df[variable].cat.categories, fill_value=0)
            return res + sum(lgamma(m + alpha) for m in counts) + lgamma(iss) - lgamma(iss + len(df))
        import itertools

        for config in itertools.product(*[df[p].cat.categories for p in parents]):   # every configuration, also unseen ones
            rows = np.ones(len(df), dtype=bool)
            for p, value in zip(parents, config):
                rows &= (df[p] == value).to_numpy()
            counts = df[variable][rows].value_counts().reindex(df[variable].cat.categories, fill_value=0)
            res += sum(lgamma(m + alpha) for m in counts)
            res += lgamma(alpha * cards[variable]) - lgamma(alpha * cards[variable] + counts.sum())
        return res

    for variable, parents in (("a", []), ("b", ["a"]), ("c", ["b"]), ("c", ["a", "b"]), ("d", ["c", "a", "b"])):
        assert bde.local_score(model, variable, parents) == pytest.approx(restated(variable, parents), rel=1e-12)
    model.add_arc("a", "b")
    assert bde.local_score(model, "b") == bde.local_score(model, "b", ["a"])
    assert bde.score(model) == pytest.approx(sum(bde.local_score(model, v) for v in ABCD), rel=1e-12)
    with pytest.raises(ValueError, match="not valid for score BDe"):
        bde.local_score(pbn.GaussianNetwork(ABCD), "a", [])

    learned = pbn.GreedyHillClimbing().estimate(pbn.ArcOperatorSet(), bde, pbn.DiscreteBN(ABCD))
    assert type(learned) is pbn.DiscreteBN
    skeleton = {frozenset(a) for a in learned.arcs()}
    assert frozenset(("a", "b")) in skeleton and frozenset(("b", "c")) in skeleton
    assert not any("d" in e for e in skeleton)            # d is independent of everything
    learned.fit(df)
    assert learned.fitted() and np.isfinite(learned.slogl(df))
