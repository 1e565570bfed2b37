"""GPU tier: the behavioural assertions of the reference's own operator / operator-set / pool / tabu-set tests
(/root/reference/tests/learning/operators/{operators,operatorset,operatorpool,operatorstabuset}_test.py), re-typed
against this package on the same synthetic table (util_test.generate_normal_data(10000), here from the golden file)."""
import numpy as np
import pytest

from helpers import frame

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def df(golden):
    return frame(golden["train10k"])


def test_operators_create_apply_opposite(pbn):   # operators_test.py:4-104
    o = pbn.AddArc("a", "b", 1)
    assert (o.source(), o.target(), o.delta()) == ("a", "b", 1)
    o = pbn.ChangeNodeType("a", pbn.CKDEType(), 4)
    assert o.node() == "a" and o.node_type() == pbn.CKDEType() and o.delta() == 4
    gbn = pbn.GaussianNetwork(["a", "b", "c", "d"])
    pbn.AddArc("a", "b", 1).apply(gbn)
    assert gbn.num_arcs() == 1 and gbn.has_arc("a", "b")
    pbn.FlipArc("a", "b", 1).apply(gbn)
    assert gbn.num_arcs() == 1 and gbn.has_arc("b", "a") and not gbn.has_arc("a", "b")
    pbn.RemoveArc("b", "a", 1).apply(gbn)
    assert gbn.num_arcs() == 0
    with pytest.raises(ValueError, match="Wrong factor type"):
        pbn.ChangeNodeType("a", pbn.CKDEType(), 1).apply(gbn)
    spbn = pbn.SemiparametricBN(["a", "b", "c", "d"])
    assert spbn.node_type("a") == pbn.UnknownFactorType()
    pbn.ChangeNodeType("a", pbn.CKDEType(), 1).apply(spbn)
    assert spbn.node_type("a") == pbn.CKDEType()
    bn = pbn.SemiparametricBN(["a", "b"])
    oppo = pbn.AddArc("a", "b", 1).opposite(bn)
    assert type(oppo) == pbn.RemoveArc and (oppo.source(), oppo.target(), oppo.delta()) == ("a", "b", -1)
    oppo = pbn.RemoveArc("a", "b", 1).opposite(bn)
    assert type(oppo) == pbn.AddArc and oppo.delta() == -1
    oppo = pbn.FlipArc("a", "b", 1).opposite(bn)
    assert type(oppo) == pbn.FlipArc and (oppo.source(), oppo.target(), oppo.delta()) == ("b", "a", -1)
    bn.set_node_type("a", pbn.LinearGaussianCPDType())
    oppo = pbn.ChangeNodeType("a", pbn.CKDEType(), 1).opposite(bn)
    assert oppo.node() == "a" and oppo.node_type() == pbn.LinearGaussianCPDType() and oppo.delta() == -1


def test_operator_set_behaviour(pbn, df):   # operatorset_test.py:9-78
    gbn = pbn.GaussianNetwork(["a", "b", "c", "d"])
    with pytest.raises(ValueError, match="can only be used with non-homogeneous"):
        pbn.ChangeNodeTypeSet().cache_scores(gbn, pbn.CVLikelihood(df))
    bic = pbn.BIC(df)
    arc_op = pbn.ArcOperatorSet()
    arc_op.set_arc_blacklist([("b", "a")])
    arc_op.set_arc_whitelist([("b", "c")])
    arc_op.set_max_indegree(3)
    arc_op.set_type_whitelist([("a", pbn.LinearGaussianCPDType())])
    arc_op.cache_scores(gbn, bic)
    arc_op.set_arc_blacklist([("e", "a")])
    with pytest.raises(ValueError, match="not present in the graph"):
        arc_op.cache_scores(gbn, bic)
    arc_op.set_arc_blacklist([])
    arc_op.set_arc_whitelist([("e", "a")])
    with pytest.raises(ValueError, match="not present in the graph"):
        arc_op.cache_scores(gbn, bic)
    # check_max_score
    gbn = pbn.GaussianNetwork(["c", "d"])
    arc_op = pbn.ArcOperatorSet()
    arc_op.cache_scores(gbn, bic)
    op = arc_op.find_max(gbn)
    assert np.isclose(op.delta(), bic.local_score(gbn, "d", ["c"]) - bic.local_score(gbn, "d"))
    arc_op.set_arc_blacklist([(op.source(), op.target())])
    arc_op.cache_scores(gbn, bic)
    op2 = arc_op.find_max(gbn)
    assert op.source() == op2.target() and op.target() == op2.source() and type(op) == type(op2) == pbn.AddArc
    # nomax
    gbn = pbn.GaussianNetwork(["a", "b"])
    arc_op = pbn.ArcOperatorSet(whitelist=[("a", "b")])
    arc_op.cache_scores(gbn, bic)
    assert arc_op.find_max(gbn) is None


def test_operator_pool_and_tabu(pbn, df):   # operatorpool_test.py:8-39, operatorstabuset_test.py
    with pytest.raises(ValueError, match="cannot be empty"):
        pbn.OperatorPool([])
    small = df.iloc[:2000]
    spbn = pbn.SemiparametricBN(["a", "b", "c", "d"])
    cv = pbn.CVLikelihood(small, 5, 0)
    arcs, node_type = pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()
    arcs.cache_scores(spbn, cv)
    spbn.set_unknown_node_types(small)
    assert not spbn.has_unknown_node_types() and spbn.node_type("a") == pbn.LinearGaussianCPDType()
    node_type.cache_scores(spbn, cv)
    arcs_max, node_max = arcs.find_max(spbn), node_type.find_max(spbn)
    pool = pbn.OperatorPool([arcs, node_type])
    pool.cache_scores(spbn, cv)
    combined = pool.find_max(spbn)
    assert combined == (arcs_max if arcs_max.delta() >= node_max.delta() else node_max)
    tabu = pbn.OperatorTabuSet()
    assert tabu.empty() and not tabu.contains(pbn.AddArc("a", "b", 1))
    tabu.insert(pbn.AddArc("a", "b", 2))
    assert not tabu.empty() and tabu.contains(pbn.AddArc("a", "b", 3))
    assert not tabu.contains(pbn.RemoveArc("b", "c", 4))
    tabu.insert(pbn.RemoveArc("b", "c", 5))
    assert tabu.contains(pbn.RemoveArc("b", "c", 6)) and not tabu.contains(pbn.FlipArc("c", "d", 7))
    tabu.clear()
    assert tabu.empty()
