"""CPU tier: the graph-structure assertions of the reference's model tests re-typed
(/root/reference/tests/models/BayesianNetwork_test.py:9-220, SemiparametricBN_test.py:9-98): constructor overloads and
their errors, node indices, parents / children, arc edits with the acyclicity predicates."""
import pytest

import pybnesian_amd as pbn
from pybnesian_amd import BayesianNetwork, GaussianNetwork, SemiparametricBN

ABCD = ["a", "b", "c", "d"]
CHAIN3 = [("a", "c"), ("b", "d"), ("c", "d")]
CYCLE = [("a", "b"), ("b", "c"), ("c", "a")]


@pytest.mark.parametrize("cls", [GaussianNetwork, SemiparametricBN])
def test_create(cls):   # BayesianNetwork_test.py:9-44, SemiparametricBN_test.py:9-47
    for args, n_arcs, nodes in (((ABCD,), 0, ABCD), ((ABCD, [("a", "c")]), 1, ABCD), ((CHAIN3,), 3, ["a", "c", "b", "d"])):
        net = cls(*args)
        assert net.num_nodes() == 4 and net.num_arcs() == n_arcs and net.nodes() == nodes
        if cls is SemiparametricBN:
            assert all(net.node_type(n) == pbn.UnknownFactorType() for n in net.nodes())
    with pytest.raises(TypeError, match="incompatible constructor arguments"):
        cls(["a", "b", "c"], [("a", "c", "b")])
    with pytest.raises(IndexError, match="not present in the graph"):
        cls(["a", "b", "c"], [("a", "d")])
    with pytest.raises(ValueError, match="must be a DAG"):
        cls(CYCLE)
    with pytest.raises(ValueError, match="must be a DAG"):
        cls(ABCD, CYCLE)


def test_create_with_node_types():   # BayesianNetwork_test.py:41-44, SemiparametricBN_test.py:50-96
    with pytest.raises(ValueError, match="Wrong factor type"):
        BayesianNetwork(pbn.GaussianNetworkType(), ABCD, [], [("a", pbn.CKDEType())])
    types = [("a", pbn.CKDEType()), ("c", pbn.CKDEType())]
    expected = {"a": pbn.CKDEType(), "b": pbn.UnknownFactorType(), "c": pbn.CKDEType(), "d": pbn.UnknownFactorType()}
    for args, n_arcs, nodes in (((ABCD, types), 0, ABCD), ((ABCD, [("a", "c")], types), 1, ABCD), ((CHAIN3, types), 3, ["a", "c", "b", "d"])):
        spbn = SemiparametricBN(*args)
        assert spbn.num_nodes() == 4 and spbn.num_arcs() == n_arcs and spbn.nodes() == nodes
        assert spbn.node_types() == expected
    with pytest.raises(TypeError, match="incompatible constructor arguments"):
        SemiparametricBN(["a", "b", "c"], [("a", "c", "b")], types)
    with pytest.raises(IndexError, match="not present in the graph"):
        SemiparametricBN(["a", "b", "c"], [("a", "d")], types)
    with pytest.raises(ValueError, match="must be a DAG"):
        SemiparametricBN(CYCLE, types)
    with pytest.raises(ValueError, match="must be a DAG"):
        SemiparametricBN(ABCD, CYCLE, types)


def test_nodes_util():   # BayesianNetwork_test.py:55-76
    for gbn in (GaussianNetwork(ABCD), GaussianNetwork(CHAIN3), GaussianNetwork(ABCD, [("a", "b"), ("b", "c")])):
        nodes, indices = gbn.nodes(), gbn.indices()
        for v in ABCD:
            assert nodes[gbn.index(v)] == v and gbn.contains_node(v)
        for i in range(4):
            assert indices[gbn.name(i)] == i
        assert not gbn.contains_node("e")


def test_parent_children():   # BayesianNetwork_test.py:78-128
    gbn = GaussianNetwork(ABCD)
    assert all(gbn.num_parents(v) == 0 and gbn.parents(v) == [] and gbn.num_children(v) == 0 for v in ABCD)
    gbn = GaussianNetwork(CHAIN3)
    assert [gbn.num_parents(v) for v in ABCD] == [0, 0, 1, 2]
    assert gbn.parents("c") == ["a"] and set(gbn.parents("d")) == {"b", "c"}
    assert [gbn.num_children(v) for v in ABCD] == [1, 1, 1, 0]
    gbn = GaussianNetwork(ABCD, [("a", "b"), ("b", "c")])
    assert [gbn.num_parents(v) for v in ABCD] == [0, 1, 1, 0]
    assert gbn.parents("b") == ["a"] and gbn.parents("c") == ["b"]
    assert [gbn.num_children(v) for v in ABCD] == [1, 1, 0, 0]


def test_arcs():   # BayesianNetwork_test.py:130-220
    gbn = GaussianNetwork(ABCD)
    assert gbn.num_arcs() == 0 and gbn.arcs() == [] and not gbn.has_arc("a", "b")
    gbn.add_arc("a", "b")
    assert gbn.arcs() == [("a", "b")] and gbn.parents("b") == ["a"] and gbn.num_children("a") == 1 and gbn.has_arc("a", "b")
    gbn.add_arc("b", "c")
    gbn.add_arc("d", "c")
    assert set(gbn.arcs()) == {("a", "b"), ("b", "c"), ("d", "c")} and set(gbn.parents("c")) == {"b", "d"}
    assert gbn.has_path("a", "c") and not gbn.has_path("a", "d") and gbn.has_path("b", "c") and gbn.has_path("d", "c")
    assert not gbn.can_add_arc("c", "a")
    assert gbn.can_add_arc("b", "c")          # exists already: adding it again is allowed (and changes nothing)
    assert gbn.can_add_arc("d", "a")
    gbn.add_arc("b", "d")
    assert gbn.num_arcs() == 4 and gbn.parents("d") == ["b"] and gbn.num_children("b") == 2
    assert gbn.has_path("a", "d") and not gbn.can_add_arc("d", "a")
    assert not gbn.can_flip_arc("b", "c") and gbn.can_flip_arc("a", "b")
    assert gbn.can_flip_arc("d", "a")         # does not exist, but could be flipped if it did
    gbn.add_arc("b", "d")
    assert gbn.num_arcs() == 4 and gbn.parents("d") == ["b"]
    with pytest.raises(ValueError, match="Cannot add arc d -> a"):
        gbn.add_arc("d", "a")
    with pytest.raises(ValueError, match="Cannot flip arc b -> c"):
        gbn.flip_arc("b", "c")
    with pytest.raises(ValueError, match="not present in the graph"):
        gbn.add_arc("a", "zz")
    gbn.remove_arc("b", "c")
    assert set(gbn.arcs()) == {("a", "b"), ("d", "c"), ("b", "d")} and gbn.parents("c") == ["d"] and not gbn.has_arc("b", "c")
    assert gbn.can_add_arc("b", "c") and not gbn.can_add_arc("c", "b") and gbn.has_path("a", "c") and gbn.has_path("b", "c")
    gbn.remove_arc("d", "c")
    assert set(gbn.arcs()) == {("a", "b"), ("b", "d")} and gbn.parents("c") == [] and gbn.num_children("d") == 0
    assert gbn.can_add_arc("b", "c") and gbn.can_add_arc("c", "b") and not gbn.has_path("a", "c")
    gbn.flip_arc("a", "b")
    assert gbn.has_arc("b", "a") and not gbn.has_arc("a", "b") and gbn.num_arcs() == 2


def test_conditional_structure_errors():
    cbn = pbn.ConditionalGaussianNetwork(["a", "b"], ["c", "d"], [("c", "a")])
    assert cbn.arcs() == [("c", "a")] and cbn.is_interface("c") and not cbn.can_add_arc("a", "c")
    with pytest.raises(ValueError, match="Interface node cannot have parents"):
        cbn.add_arc("a", "c")
    with pytest.raises(ValueError, match="Interface node cannot have parents"):
        pbn.ConditionalGaussianNetwork(["a", "b"], ["c", "d"], [("a", "d")])
