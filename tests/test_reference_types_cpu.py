"""CPU tier (no device work): the type / extension-point / pickle assertions of the reference's tests re-typed -
/root/reference/tests/models/BayesianNetwork_type_test.py, HeterogeneousBN_test.py, factors/factor_type_test.py,
serialization/serialize_factor_type_test.py, serialize_models_type_test.py, serialize_factor_test.py (unfitted and
LinearGaussian parts) and serialize_models_test.py (structure and LinearGaussian-factor parts).  The fitted CKDE /
DiscreteFactor / hill-climbing halves of those tests need the device and live in test_reference_types_gpu.py."""
import itertools
import pickle

import pyarrow as pa
import pytest

pytestmark = pytest.mark.extra

import pybnesian_amd as pbn
from pybnesian_amd import (CKDE, BayesianNetwork, BayesianNetworkType, ConditionalBayesianNetwork, DiscreteBN, DiscreteFactor, Factor,
                           FactorType, GaussianNetwork, KDENetwork, LinearGaussianCPD, SemiparametricBN)

ABCD = ["a", "b", "c", "d"]


# ---- user-defined types (module level so that they pickle) ------------------------------------------------------------
class NewFactorType(FactorType):
    def __init__(self, factor_class=None):
        FactorType.__init__(self)
        self.factor_class = factor_class

    def new_factor(self, model, variable, evidence):
        return self.factor_class(variable, evidence)

    def __str__(self):
        return "NewType"


class OtherFactorType(FactorType):
    def __init__(self):
        FactorType.__init__(self)


class NewFactor(Factor):
    def __init__(self, variable, evidence):
        Factor.__init__(self, variable, evidence)
        self._fitted = False
        self.some_fit_data = None

    def fit(self, df):
        self.some_fit_data = "fitted"
        self._fitted = True

    def fitted(self):
        return self._fitted

    def type(self):
        return NewFactorType(NewFactor)

    def __getstate_extra__(self):
        return {"fitted": self._fitted, "some_fit_data": self.some_fit_data}

    def __setstate_extra__(self, d):
        self._fitted = d["fitted"]
        self.some_fit_data = d["some_fit_data"]


class NewFactorBis(Factor):
    def __init__(self, variable, evidence):
        Factor.__init__(self, variable, evidence)
        self._fitted = False
        self.some_fit_data = None

    def fit(self, df):
        self.some_fit_data = "fitted"
        self._fitted = True

    def fitted(self):
        return self._fitted

    def type(self):
        return NewFactorType(NewFactorBis)

    def __getstate__(self):
        return {"variable": self.variable(), "evidence": self.evidence(), "fitted": self._fitted, "some_fit_data": self.some_fit_data}

    def __setstate__(self, d):
        Factor.__init__(self, d["variable"], d["evidence"])
        self._fitted = d["fitted"]
        self.some_fit_data = d["some_fit_data"]


class MyRestrictedGaussianNetworkType(BayesianNetworkType):
    def __init__(self):
        BayesianNetworkType.__init__(self)

    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return pbn.LinearGaussianCPDType()

    def can_have_arc(self, model, source, target):
        return "a" in source

    def new_bn(self, nodes):
        return NewBN(nodes)

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalNewBN(nodes, interface_nodes)

    def __str__(self):
        return "MyRestrictedGaussianNetworkType"


class NewBN(BayesianNetwork):
    def __init__(self, variables, arcs=None):
        if arcs is None:
            BayesianNetwork.__init__(self, MyRestrictedGaussianNetworkType(), variables)
        else:
            BayesianNetwork.__init__(self, MyRestrictedGaussianNetworkType(), variables, arcs)


class ConditionalNewBN(ConditionalBayesianNetwork):
    def __init__(self, variables, interface, arcs=None):
        if arcs is None:
            ConditionalBayesianNetwork.__init__(self, MyRestrictedGaussianNetworkType(), variables, interface)
        else:
            ConditionalBayesianNetwork.__init__(self, MyRestrictedGaussianNetworkType(), variables, interface, arcs)


class NonHomogeneousType(BayesianNetworkType):
    def __init__(self):
        BayesianNetworkType.__init__(self)

    def is_homogeneous(self):
        return False

    def data_default_node_type(self, dt):
        if dt.equals(pa.float64()) or dt.equals(pa.float32()):
            return [pbn.LinearGaussianCPDType()]
        raise ValueError("Data type not compatible with NonHomogeneousType")

    def new_bn(self, nodes):
        return OtherBN(nodes)

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalOtherBN(nodes, interface_nodes)

    def __str__(self):
        return "NonHomogeneousType"


class OtherBN(BayesianNetwork):
    def __init__(self, variables, arcs=None, node_types=None):
        args = [a for a in (arcs, node_types) if a is not None]
        BayesianNetwork.__init__(self, NonHomogeneousType(), variables, *args)
        self.extra_info = "extra"

    def __getstate_extra__(self):
        return self.extra_info

    def __setstate_extra__(self, t):
        self.extra_info = t


class ConditionalOtherBN(ConditionalBayesianNetwork):
    def __init__(self, variables, interface, arcs=None, node_types=None):
        args = [a for a in (arcs, node_types) if a is not None]
        ConditionalBayesianNetwork.__init__(self, NonHomogeneousType(), variables, interface, *args)
        self.extra_info = "extra"

    def __getstate_extra__(self):
        return self.extra_info

    def __setstate_extra__(self, t):
        self.extra_info = t


class NewBNType(BayesianNetworkType):
    def __init__(self):
        BayesianNetworkType.__init__(self)

    def __str__(self):
        return "NewType"


class OtherBNType(BayesianNetworkType):
    def __init__(self):
        BayesianNetworkType.__init__(self)
        self.some_useful_info = "info"

    def __str__(self):
        return "OtherType"


class DynamicNewBN(pbn.DynamicBayesianNetwork):
    def __init__(self, variables, markovian_order):
        pbn.DynamicBayesianNetwork.__init__(self, MyRestrictedGaussianNetworkType(), variables, markovian_order)


class DynamicOtherBN(pbn.DynamicBayesianNetwork):
    def __init__(self, variables, markovian_order, static_bn=None, transition_bn=None):
        if static_bn is None or transition_bn is None:
            pbn.DynamicBayesianNetwork.__init__(self, NonHomogeneousType(), variables, markovian_order)
        else:
            pbn.DynamicBayesianNetwork.__init__(self, variables, markovian_order, static_bn, transition_bn)
        self.extra_info = "extra"

    def __getstate_extra__(self):
        return self.extra_info

    def __setstate_extra__(self, t):
        self.extra_info = t


def roundtrip(x):
    return pickle.loads(pickle.dumps(x))


# ---- factors/factor_type_test.py ------------------------------------------------------------------------------------------
def test_factor_type():
    for cls, t in ((LinearGaussianCPD, pbn.LinearGaussianCPDType()), (CKDE, pbn.CKDEType()), (DiscreteFactor, pbn.DiscreteFactorType())):
        f1, f2, f3 = cls("a", []), cls("b", ["a"]), cls("c", ["b", "a"])
        assert f1.type() == t and f1.type() == f2.type() == f3.type()
    assert LinearGaussianCPD("a", []).type() != CKDE("a", []).type()
    assert LinearGaussianCPD("a", []).type() != DiscreteFactor("a", []).type()
    assert CKDE("a", []).type() != DiscreteFactor("a", []).type()


def test_new_factor_type():
    class A(FactorType):
        def __init__(self):
            FactorType.__init__(self)

    class B(FactorType):
        def __init__(self):
            FactorType.__init__(self)

    assert A() == A() and B() == B() and A() != B()


def test_factor_defined_factor_type():
    class F_type(FactorType):
        def __init__(self):
            FactorType.__init__(self)

        def __str__(self):
            return "FType"

    class F(Factor):
        def __init__(self, variable, evidence):
            Factor.__init__(self, variable, evidence)

        def type(self):
            return F_type()

    f1, f2, f3 = F("a", []), F("b", ["a"]), F("c", ["a", "b"])
    assert f1.type() == f2.type() == f3.type()
    assert str(f1.type()) == str(f2.type()) == str(f3.type()) == "FType"
    dummy_network = pbn.GaussianNetwork(ABCD)
    with pytest.raises(RuntimeError, match='Tried to call pure virtual function "FactorType::new_factor"'):
        f1.type().new_factor(dummy_network, "d", ["a", "b", "c"])

    class G_type(FactorType):
        def __init__(self):
            FactorType.__init__(self)

        def new_factor(self, model, variable, evidence):
            return G(variable, evidence)

        def __str__(self):
            return "GType"

    class G(Factor):
        def __init__(self, variable, evidence):
            Factor.__init__(self, variable, evidence)

        def type(self):
            return G_type()

    g1 = G("a", [])
    assert g1.type() == G("b", ["a"]).type() and f1.type() != g1.type() and str(g1.type()) == "GType"
    g4 = g1.type().new_factor(dummy_network, "d", ["a", "b", "c"])
    assert g1.type() == g4.type() and g4.variable() == "d" and g4.evidence() == ["a", "b", "c"]


# ---- models/BayesianNetwork_type_test.py -----------------------------------------------------------------------------------
def test_bn_type():
    nets = []
    for cls, t in ((GaussianNetwork, pbn.GaussianNetworkType()), (SemiparametricBN, pbn.SemiparametricBNType()),
                   (KDENetwork, pbn.KDENetworkType()), (DiscreteBN, pbn.DiscreteBNType())):
        n1, n2, n3 = cls(ABCD), cls(ABCD), cls(ABCD)
        assert n1.type() == t and n1.type() == n2.type() == n3.type()
        nets.append(n1)
    for x, y in itertools.combinations(nets, 2):
        assert x.type() != y.type()


def test_new_bn_type():
    class MyGaussianNetworkType(BayesianNetworkType):
        def __init__(self):
            BayesianNetworkType.__init__(self)

        def is_homogeneous(self):
            return True

        def default_node_type(self):
            return pbn.LinearGaussianCPDType()

        def can_have_arc(self, model, source, target):
            return source == "a"

    class MySemiparametricBNType(BayesianNetworkType):
        def __init__(self):
            BayesianNetworkType.__init__(self)

    a1, b1 = MyGaussianNetworkType(), MySemiparametricBNType()
    assert a1 == MyGaussianNetworkType() and b1 == MySemiparametricBNType() and a1 != b1
    mybn = BayesianNetwork(a1, ABCD)
    assert mybn.can_add_arc("a", "b")
    assert not mybn.can_add_arc("b", "a")
    assert not mybn.can_add_arc("c", "d")


def test_new_specific_bn_type_structure():   # BayesianNetwork_type_test.py:118-170 without the hill-climb (GPU tier)
    sp1, sp2, sp3 = NewBN(ABCD), NewBN(ABCD, [("a", "b")]), NewBN(ABCD)
    assert sp1.type() == sp2.type() == sp3.type()
    assert sp1.can_add_arc("a", "b") and not sp1.can_add_arc("b", "a") and not sp1.can_add_arc("c", "d")
    assert sp1.num_arcs() == sp3.num_arcs() == 0 and sp2.arcs() == [("a", "b")]
    csp1, csp2 = ConditionalNewBN(["a", "b"], ["c", "d"]), ConditionalNewBN(["a", "b"], ["c", "d"], [("a", "b")])
    assert csp1.type() == csp2.type()
    assert csp1.can_add_arc("a", "b") and not csp1.can_add_arc("b", "a") and not csp1.can_add_arc("c", "d")
    assert csp1.num_arcs() == 0 and csp2.arcs() == [("a", "b")]
    assert isinstance(sp1.conditional_bn(["a", "b"], ["c", "d"]), ConditionalNewBN)
    assert isinstance(csp1.unconditional_bn(), NewBN)
    clone = sp2.clone()
    assert type(clone) is NewBN and clone.arcs() == [("a", "b")]
    clone.remove_arc("a", "b")
    assert sp2.arcs() == [("a", "b")]


# ---- models/HeterogeneousBN_test.py -----------------------------------------------------------------------------------------
def test_heterogeneous_type_equality():
    ck, lg, dd = pbn.CKDEType(), pbn.LinearGaussianCPDType(), pbn.DiscreteFactorType()
    het_single = pbn.HeterogeneousBN([ck, lg], ABCD)
    assert het_single.type() == pbn.HeterogeneousBN([ck, lg], ABCD).type()
    assert het_single.type() != pbn.HeterogeneousBN([lg, ck], ABCD).type()
    dict_t = pa.dictionary(pa.int8(), pa.string())
    het_dt = pbn.HeterogeneousBN({pa.float64(): [ck, lg], pa.float32(): [ck, lg], dict_t: [dd]}, ABCD)
    het2_dt = pbn.HeterogeneousBN({dict_t: [dd], pa.float32(): [ck, lg], pa.float64(): [ck, lg]}, ABCD)
    assert het_dt.type() == het2_dt.type()                       # the order of the map is not relevant
    het3_dt = pbn.HeterogeneousBN({dict_t: [dd], pa.float32(): [lg, ck], pa.float64(): [ck, lg]}, ABCD)
    assert het_dt.type() != het3_dt.type()                       # the order of the defaults is
    assert het_single.type() != pbn.HeterogeneousBN({pa.float64(): [ck, lg]}, ABCD).type()
    assert het_dt.type().data_default_node_type(pa.float32()) == [ck, lg]
    assert het_single.node_types() == {n: pbn.UnknownFactorType() for n in ABCD}
    hom = pbn.HomogeneousBN(ck, ABCD, [("a", "b")])
    assert hom.type() == pbn.HomogeneousBNType(ck) != pbn.HomogeneousBNType(lg) and hom.node_type("a") == ck


# ---- serialization/serialize_factor_type_test.py, serialize_models_type_test.py ---------------------------------------------
def test_serialization_factor_type():
    types = [pbn.LinearGaussianCPDType(), pbn.CKDEType(), pbn.DiscreteFactorType(), NewFactorType(), OtherFactorType()]
    for t in types:
        assert roundtrip(t) == type(t)()
    for x, y in itertools.combinations(types, 2):
        assert x != y


def test_serialization_bn_type():
    types = [pbn.GaussianNetworkType(), pbn.SemiparametricBNType(), pbn.KDENetworkType(), pbn.DiscreteBNType(), NewBNType(), OtherBNType()]
    loaded = [roundtrip(t) for t in types]
    for t, l in zip(types, loaded):
        assert l == type(t)()
    assert loaded[-1].some_useful_info == "info"
    for x, y in itertools.combinations(loaded, 2):
        assert x != y


# ---- serialization/serialize_factor_test.py (no device) ---------------------------------------------------------------------
def test_serialization_unfitted_factor():
    loaded = {}
    for cls, t in ((LinearGaussianCPD, pbn.LinearGaussianCPDType()), (CKDE, pbn.CKDEType()), (DiscreteFactor, pbn.DiscreteFactorType())):
        f = roundtrip(cls("c", ["a", "b"]))
        assert f.variable() == "c" and set(f.evidence()) == {"a", "b"} and not f.fitted() and f.type() == t
        loaded[cls] = f
    dummy_network = GaussianNetwork(ABCD)
    new = roundtrip(NewFactor("c", ["a", "b"]))
    assert new.variable() == "c" and set(new.evidence()) == {"a", "b"} and not new.fitted()
    assert type(new.type()) == NewFactorType and new.type() == NewFactor("a", []).type()
    assert type(new.type().new_factor(dummy_network, "a", [])) == NewFactor
    bis = roundtrip(NewFactorBis("c", ["a", "b"]))
    assert bis.variable() == "c" and not bis.fitted() and bis.type() == NewFactorBis("a", []).type()
    assert type(bis.type().new_factor(dummy_network, "a", [])) == NewFactorBis
    assert bis.type() == new.type()
    for x, y in itertools.combinations([loaded[LinearGaussianCPD], loaded[CKDE], loaded[DiscreteFactor], new], 2):
        assert x.type() != y.type()


def test_serialization_fitted_factor_host_side():
    lg = roundtrip(LinearGaussianCPD("c", ["a", "b"], [1, 2, 3], 0.5))
    assert lg.variable() == "c" and set(lg.evidence()) == {"a", "b"} and lg.fitted()
    assert list(lg.beta) == [1, 2, 3] and lg.variance == 0.5
    for cls in (NewFactor, NewFactorBis):
        n = cls("c", ["a", "b"])
        n.fit(None)
        l = roundtrip(n)
        assert l.fitted() and l.some_fit_data == "fitted" and l.type() == cls("a", []).type()


# ---- serialization/serialize_models_test.py ----------------------------------------------------------------------------------
def test_serialization_bn_model():
    arcs = [("a", "b")]
    for net, t in ((GaussianNetwork(ABCD, arcs), pbn.GaussianNetworkType()), (KDENetwork(ABCD, arcs), pbn.KDENetworkType()),
                   (DiscreteBN(ABCD, arcs), pbn.DiscreteBNType()),
                   (BayesianNetwork(MyRestrictedGaussianNetworkType(), ABCD, arcs), MyRestrictedGaussianNetworkType()),
                   (NewBN(ABCD, arcs), MyRestrictedGaussianNetworkType())):
        l = roundtrip(net)
        assert set(l.nodes()) == set(ABCD) and l.arcs() == arcs and l.type() == t and type(l) is type(net)
    s = roundtrip(SemiparametricBN(ABCD, arcs, [("b", pbn.CKDEType())]))
    assert s.type() == pbn.SemiparametricBNType() and s.arcs() == arcs
    assert s.node_types() == {"a": pbn.UnknownFactorType(), "b": pbn.CKDEType(), "c": pbn.UnknownFactorType(), "d": pbn.UnknownFactorType()}
    o = roundtrip(OtherBN(ABCD, arcs, [("b", pbn.LinearGaussianCPDType()), ("c", pbn.CKDEType()), ("d", pbn.DiscreteFactorType())]))
    assert o.arcs() == arcs and o.type() == NonHomogeneousType() and o.extra_info == "extra"
    assert o.node_types() == {"a": pbn.UnknownFactorType(), "b": pbn.LinearGaussianCPDType(), "c": pbn.CKDEType(), "d": pbn.DiscreteFactorType()}
    assert roundtrip(NewBN(ABCD, arcs)).type() != o.type()


def test_serialization_fitted_bn_linear_gaussian():
    gaussian = GaussianNetwork(ABCD, [("a", "b")])
    gaussian.add_cpds([LinearGaussianCPD("b", ["a"], [1, 2], 2)])
    gaussian.include_cpd = True
    partial = roundtrip(gaussian)
    assert not partial.fitted()
    cpd = partial.cpd("b")
    assert cpd.variable() == "b" and cpd.evidence() == ["a"] and list(cpd.beta) == [1, 2] and cpd.variance == 2

    gaussian = GaussianNetwork(ABCD, [("a", "b")])
    gaussian.add_cpds([LinearGaussianCPD("a", [], [0], 0.5), LinearGaussianCPD("b", ["a"], [1, 2], 2),
                       LinearGaussianCPD("c", [], [2], 1), LinearGaussianCPD("d", [], [3], 1.5)])
    assert not roundtrip(gaussian).fitted()
    gaussian.include_cpd = True
    fitted = roundtrip(gaussian)
    assert fitted.fitted()
    for v, ev, beta, var in (("a", [], [0], 0.5), ("b", ["a"], [1, 2], 2), ("c", [], [2], 1), ("d", [], [3], 1.5)):
        cpd = fitted.cpd(v)
        assert cpd.variable() == v and cpd.evidence() == ev and list(cpd.beta) == beta and cpd.variance == var

    other = OtherBN(ABCD, [("a", "b")], [("b", pbn.LinearGaussianCPDType()), ("c", pbn.CKDEType()), ("d", pbn.DiscreteFactorType())])
    other.add_cpds([LinearGaussianCPD("b", ["a"], [1, 2], 2)])
    other.include_cpd = True
    lo = roundtrip(other)
    assert not lo.fitted() and lo.cpd("b").variance == 2 and lo.extra_info == "extra"


def test_serialization_conditional_bn_model():
    arcs = [("a", "c")]
    for net, t in ((pbn.ConditionalGaussianNetwork(["c", "d"], ["a", "b"], arcs), pbn.GaussianNetworkType()),
                   (pbn.ConditionalKDENetwork(["c", "d"], ["a", "b"], arcs), pbn.KDENetworkType()),
                   (pbn.ConditionalDiscreteBN(["c", "d"], ["a", "b"], arcs), pbn.DiscreteBNType()),
                   (ConditionalBayesianNetwork(MyRestrictedGaussianNetworkType(), ["c", "d"], ["a", "b"], arcs), MyRestrictedGaussianNetworkType()),
                   (ConditionalNewBN(["c", "d"], ["a", "b"], arcs), MyRestrictedGaussianNetworkType())):
        l = roundtrip(net)
        assert set(l.nodes()) == {"c", "d"} and set(l.interface_nodes()) == {"a", "b"} and l.arcs() == arcs
        assert l.type() == t and type(l) is type(net)
    s = roundtrip(pbn.ConditionalSemiparametricBN(["c", "d"], ["a", "b"], arcs, [("c", pbn.CKDEType())]))
    assert s.type() == pbn.SemiparametricBNType() and s.node_type("c") == pbn.CKDEType() and s.node_type("d") == pbn.UnknownFactorType()
    o = roundtrip(ConditionalOtherBN(["c", "d"], ["a", "b"], arcs, [("c", pbn.CKDEType()), ("d", pbn.DiscreteFactorType())]))
    assert o.type() == NonHomogeneousType() and o.extra_info == "extra" and o.node_type("d") == pbn.DiscreteFactorType()

    cg = pbn.ConditionalGaussianNetwork(["c", "d"], ["a", "b"], arcs)
    cg.add_cpds([LinearGaussianCPD("c", ["a"], [1, 2], 2)])
    cg.include_cpd = True
    l = roundtrip(cg)
    assert not l.fitted() and list(l.cpd("c").beta) == [1, 2] and l.cpd("c").evidence() == ["a"]
    cg.add_cpds([LinearGaussianCPD("d", [], [3], 1.5)])
    assert roundtrip(cg).fitted()


def test_serialization_dbn_model():
    def arcs(d, static, transition):
        d.static_bn().add_arc(*static)
        d.transition_bn().add_arc(*transition)
        return d

    st, tr = ("a_t_2", "d_t_1"), ("c_t_2", "b_t_0")
    for cls, t in ((pbn.DynamicGaussianNetwork, pbn.GaussianNetworkType()), (pbn.DynamicSemiparametricBN, pbn.SemiparametricBNType()),
                   (pbn.DynamicKDENetwork, pbn.KDENetworkType()), (pbn.DynamicDiscreteBN, pbn.DiscreteBNType())):
        d = arcs(cls(ABCD, 2), st, tr)
        if cls is pbn.DynamicSemiparametricBN:
            d.transition_bn().set_node_type("b_t_0", pbn.CKDEType())
        l = roundtrip(d)
        assert set(l.variables()) == set(ABCD) and l.static_bn().arcs() == [st] and l.transition_bn().arcs() == [tr]
        assert l.type() == t and type(l) is cls
        if cls is pbn.DynamicSemiparametricBN:
            want = {v + "_t_0": pbn.UnknownFactorType() for v in ABCD}
            want["b_t_0"] = pbn.CKDEType()
            got = l.transition_bn().node_types()
            assert {k: got[k] for k in want} == want
    tr2 = ("a_t_2", "b_t_0")
    for d in (arcs(pbn.DynamicBayesianNetwork(MyRestrictedGaussianNetworkType(), ABCD, 2), st, tr2), arcs(DynamicNewBN(ABCD, 2), st, tr2)):
        l = roundtrip(d)
        assert l.static_bn().arcs() == [st] and l.transition_bn().arcs() == [tr2] and l.type() == MyRestrictedGaussianNetworkType()
        with pytest.raises(ValueError):
            l.transition_bn().add_arc("c_t_2", "b_t_0")      # the user type's can_have_arc survives the round trip
    other = arcs(DynamicOtherBN(ABCD, 2), st, tr2)
    other.static_bn().set_node_type("c_t_1", pbn.DiscreteFactorType())
    other.static_bn().set_node_type("d_t_1", pbn.CKDEType())
    other.transition_bn().set_node_type("d_t_0", pbn.CKDEType())
    l = roundtrip(other)
    assert l.type() == NonHomogeneousType() and l.extra_info == "extra" and type(l) is DynamicOtherBN
    assert l.static_bn().node_type("c_t_1") == pbn.DiscreteFactorType() and l.static_bn().node_type("d_t_1") == pbn.CKDEType()
    assert l.transition_bn().node_type("d_t_0") == pbn.CKDEType()


def test_serialization_partially_fitted_dbn():
    g = pbn.DynamicGaussianNetwork(ABCD, 2)
    g.static_bn().add_arc("a_t_2", "d_t_1")
    g.transition_bn().add_arc("c_t_2", "b_t_0")
    g.static_bn().add_cpds([LinearGaussianCPD("d_t_1", ["a_t_2"], [1, 2], 2)])
    g.transition_bn().add_cpds([LinearGaussianCPD("b_t_0", ["c_t_2"], [3, 4], 5)])
    g.include_cpd = True
    l = roundtrip(g)
    assert not l.fitted() and not l.static_bn().fitted() and not l.transition_bn().fitted()
    cpd = l.static_bn().cpd("d_t_1")
    assert cpd.evidence() == ["a_t_2"] and list(cpd.beta) == [1, 2] and cpd.variance == 2
    cpd = l.transition_bn().cpd("b_t_0")
    assert cpd.evidence() == ["c_t_2"] and list(cpd.beta) == [3, 4] and cpd.variance == 5

    variables = ABCD
    static_nodes = [v + "_t_" + str(m) for v in variables for m in range(1, 3)]
    transition_nodes = [v + "_t_0" for v in variables]
    other_static = OtherBN(static_nodes, [("a_t_2", "d_t_1")], [("b_t_1", pbn.DiscreteFactorType()), ("c_t_1", pbn.CKDEType()),
                                                                ("d_t_1", pbn.LinearGaussianCPDType())])
    other_static.add_cpds([LinearGaussianCPD("d_t_1", ["a_t_2"], [1, 2], 2)])
    other_transition = ConditionalOtherBN(transition_nodes, static_nodes, [("a_t_2", "d_t_0")],
                                          [("b_t_0", pbn.DiscreteFactorType()), ("c_t_0", pbn.CKDEType()), ("d_t_0", pbn.LinearGaussianCPDType())])
    other_transition.add_cpds([LinearGaussianCPD("d_t_0", ["a_t_2"], [3, 4], 1.5)])
    assert other_static.type() == other_transition.type()
    dyn_other = DynamicOtherBN(variables, 2, other_static, other_transition)
    dyn_other.include_cpd = True
    l = roundtrip(dyn_other)
    assert not l.fitted()
    assert l.static_bn().node_type("b_t_1") == pbn.DiscreteFactorType() and l.static_bn().node_type("c_t_1") == pbn.CKDEType()
    assert l.transition_bn().node_type("b_t_0") == pbn.DiscreteFactorType() and l.transition_bn().node_type("d_t_0") == pbn.LinearGaussianCPDType()
    assert list(l.static_bn().cpd("d_t_1").beta) == [1, 2] and l.transition_bn().cpd("d_t_0").variance == 1.5


def test_discrete_factor_data_type():   # factors/discrete/DiscreteFactor_test.py:11-35
    import numpy as np
    import pandas as pd

    a = pbn.DiscreteFactor("A", [])
    with pytest.raises(ValueError, match="DiscreteFactor factor not fitted."):
        a.data_type()
    for ncat, index_type in ((2, pa.int8()), (128, pa.int8()), (129, pa.int16())):
        categories = np.asarray(["a" + str(i) for i in range(1, ncat + 1)])
        values = pd.Categorical(categories[np.random.RandomState(ncat).randint(len(categories), size=100)], categories=categories, ordered=False)
        a.fit(pd.DataFrame({"A": values}))
        assert a.data_type() == pa.dictionary(index_type, pa.string())


def test_documented_names_and_small_helpers(tmp_path):
    """Names of the reference's API pages (docs/source/api) that user code reaches for around the hot path: operator and
    network base classes, the MLE estimators and their parameter classes, the SaveModel callback."""
    import numpy as np
    import pandas as pd

    assert isinstance(pbn.RemoveArc("a", "b", 0.0), pbn.ArcOperator) and isinstance(pbn.FlipArc("a", "b", 0.0), pbn.Operator)
    assert not isinstance(pbn.RemoveArc("a", "b", 0.0), pbn.AddArc)
    assert issubclass(pbn.ArcOperatorSet, pbn.OperatorSet)
    assert isinstance(GaussianNetwork(ABCD), pbn.BayesianNetworkBase)
    assert isinstance(pbn.ConditionalGaussianNetwork(["a"], ["b"]), pbn.ConditionalBayesianNetworkBase)
    assert isinstance(pbn.DynamicGaussianNetwork(["a", "b"], 1), pbn.DynamicBayesianNetworkBase)
    with pytest.raises(ValueError, match="MLE not available"):   # mle_test.py:26-31
        pbn.MLE(pbn.CKDEType())
    assert isinstance(pbn.MLE(pbn.LinearGaussianCPDType()), pbn.MLELinearGaussianCPD)
    rng = np.random.RandomState(0)   # pybindings_parameters.cpp:151-165
    df = pd.DataFrame({"variable": rng.choice(["a1", "a2", "a3"], size=50, p=[0.5, 0.3, 0.2]),
                       "evidence": rng.choice(["b1", "b2"], size=50, p=[0.5, 0.5])}, dtype="category")
    params = pbn.MLE(pbn.DiscreteFactorType()).estimate(df, "variable", ["evidence"])
    assert isinstance(params, pbn.DiscreteFactorParams) and params.logprob.shape == (3, 2) and list(params.cardinality) == [3, 2]
    assert np.allclose(np.exp(params.logprob).sum(axis=0), 1.0)
    p = pbn.LinearGaussianParams([1, 2], 0.5)
    assert list(p.beta) == [1, 2] and p.variance == 0.5
    cb = pbn.SaveModel(str(tmp_path))
    cb.call(GaussianNetwork(ABCD, [("a", "b")]), None, None, 7)
    assert pbn.load(str(tmp_path / "000007.pickle")).arcs() == [("a", "b")]
    with pytest.raises(NotImplementedError, match="DynamicScore::static_score"):
        pbn.DynamicScore().static_score()


def test_factor_save_and_extra_state(tmp_path):
    """Factor.save (factors.hpp:150-152) and the extra-state protocol (docs/source/extending.rst:196-255): only variable,
    evidence and __getstate_extra__() travel; __setstate_extra__() restores."""
    n = NewFactor("c", ["a", "b"])
    n.fit(None)
    n.scratch = "not part of the extra state"
    n.save(str(tmp_path / "factor"))
    loaded = pbn.load(str(tmp_path / "factor.pickle"))
    assert type(loaded) is NewFactor and loaded.variable() == "c" and loaded.evidence() == ["a", "b"]
    assert loaded.fitted() and loaded.some_fit_data == "fitted" and not hasattr(loaded, "scratch")
    lg = LinearGaussianCPD("b", ["a"], [1.0, 2.0], 0.5)
    lg.save(str(tmp_path / "lg.pickle"))
    again = pbn.load(str(tmp_path / "lg.pickle"))
    assert list(again.beta) == [1.0, 2.0] and again.variance == 0.5
