"""GPU tier: the walkthrough of the reference's extension guide (/root/reference/docs/source/extending.rst:117-490,
535-640, 700-900) re-typed - a user-defined linear-Gaussian factor inside HomogeneousBN / HeterogeneousBN next to the
library's factors, a user-defined network type with arc restrictions, and a user network class with its own pickle."""
import pickle

import numpy as np
import pandas as pd
import pyarrow as pa
import pytest
from scipy.stats import norm

import pybnesian_amd as pbn
from pybnesian_amd import BayesianNetwork, BayesianNetworkType, ConditionalBayesianNetwork, Factor, FactorType

pytestmark = pytest.mark.gpu
ABCD = ["a", "b", "c", "d"]


class MyLGType(FactorType):
    def __init__(self):
        FactorType.__init__(self)

    def __str__(self):
        return "MyLGType"

    def new_factor(self, model, variable, evidence, *args, **kwargs):
        return MyLG(variable, evidence)


class MyLG(Factor):   # extending.rst:142-194: least squares on the RecordBatch the model hands over
    def __init__(self, variable, evidence):
        Factor.__init__(self, variable, evidence)
        self._fitted = False
        self.beta = np.empty((1 + len(evidence),))
        self.variance = -1

    def data_type(self):
        return pa.float64()

    def fit(self, df):
        pandas_df = df.to_pandas()
        restricted = pandas_df.loc[:, [self.variable()] + self.evidence()].dropna()
        y = restricted.loc[:, self.variable()].to_numpy()
        x = restricted.loc[:, self.evidence()].to_numpy()
        design = np.column_stack((np.ones(x.shape[0]), x))
        self.beta, res, _, _ = np.linalg.lstsq(design, y, rcond=None)
        self.variance = res[0] / (design.shape[0] - 1 - len(self.evidence()))
        self._fitted = True

    def fitted(self):
        return self._fitted

    def logl(self, df):
        pandas_df = df.to_pandas()
        means = self.beta[0] + np.sum(self.beta[1:] * pandas_df.loc[:, self.evidence()], axis=1)
        return norm.logpdf(pandas_df.loc[:, self.variable()], means, np.sqrt(self.variance))

    def slogl(self, df):
        return self.logl(df).sum()

    def type(self):
        return MyLGType()

    def __getstate_extra__(self):
        return {"fitted": self._fitted, "beta": self.beta, "variance": self.variance}

    def __setstate_extra__(self, extra):
        self._fitted, self.beta, self.variance = extra["fitted"], extra["beta"], extra["variance"]


class MyRestrictedGaussianType(BayesianNetworkType):   # extending.rst:541-580
    def __init__(self):
        BayesianNetworkType.__init__(self)

    def __str__(self):
        return "MyRestrictedGaussianType"

    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return MyLGType()

    def can_have_arc(self, model, source, target):
        return "a" in source.lower()

    def new_bn(self, nodes):
        return BayesianNetwork(MyRestrictedGaussianType(), nodes)

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalBayesianNetwork(MyRestrictedGaussianType(), nodes, interface_nodes)


class MyRestrictedBN(BayesianNetwork):   # extending.rst:700-800, 870-895
    def __init__(self, nodes, arcs=None):
        if arcs is None:
            BayesianNetwork.__init__(self, MyRestrictedGaussianType(), nodes)
        else:
            BayesianNetwork.__init__(self, MyRestrictedGaussianType(), nodes, arcs)
        self.extra_data = "extra"
        self.log = []

    def add_arc(self, source, target):
        self.log.append(f"Adding arc {source} -> {target}")
        BayesianNetwork.add_arc(self, source, target)

    def __getstate__(self):
        d = {"graph": self.graph(), "type": self.type(), "factor_types": list(self.node_types().items()), "extra_data": self.extra_data}
        if self.include_cpd:
            d["factors"] = [self.cpd(n) for n in self.nodes()]
        return d

    def __setstate__(self, d):
        BayesianNetwork.__init__(self, d["type"], d["graph"], d["factor_types"])
        if "factors" in d:
            self.add_cpds(d["factors"])
        self.extra_data = d["extra_data"]
        self.log = []


def sample_data(size, seed=0):
    rng = np.random.RandomState(seed)
    a = rng.normal(3, 0.5, size=size)
    b = rng.normal(2.5, 2, size=size)
    c = -4.2 + 1.2 * a + 3.2 * b + rng.normal(0, 0.75, size=size)
    d = 1.5 - 0.3 * c + rng.normal(0, 0.5, size=size)
    return pd.DataFrame({"a": a, "b": b, "c": c, "d": d})


def same_parameters(cpd1, cpd2):
    assert np.all(np.isclose(cpd1.beta, cpd2.beta)) and np.isclose(cpd1.variance, cpd2.variance)


def test_user_factor_in_generic_networks():   # extending.rst:376-450
    df, df_test = sample_data(300), sample_data(20, seed=1)
    with pytest.raises(ValueError, match="Wrong factor type"):
        pbn.GaussianNetwork(ABCD).set_node_type("a", MyLGType())
    homo = pbn.HomogeneousBN(MyLGType(), ABCD, [("a", "c")])
    homo.fit(df)
    gbn = pbn.GaussianNetwork(ABCD, [("a", "c")])
    gbn.fit(df)
    for v in ABCD:
        assert type(homo.cpd(v)) is MyLG
        same_parameters(homo.cpd(v), gbn.cpd(v))
    assert np.all(np.isclose(homo.logl(df_test), gbn.logl(df_test))) and np.isclose(homo.slogl(df_test), gbn.slogl(df_test))

    het = pbn.HeterogeneousBN([MyLGType()], ABCD, [("a", "c")])
    het.set_node_type("a", pbn.CKDEType())
    het.fit(df)
    spbn = pbn.SemiparametricBN(ABCD, [("a", "c")], [("a", pbn.CKDEType())])
    spbn.fit(df)
    assert type(het.cpd("a")) is pbn.CKDE
    for v in "bcd":
        same_parameters(het.cpd(v), spbn.cpd(v))
    assert np.all(np.isclose(het.logl(df_test), spbn.logl(df_test))) and np.isclose(het.slogl(df_test), spbn.slogl(df_test))
    het.include_cpd = True
    loaded = pickle.loads(pickle.dumps(het))
    assert loaded.fitted() and np.allclose(loaded.logl(df_test), het.logl(df_test))


def test_heterogeneous_defaults_per_data_type():   # extending.rst:452-489
    rng = np.random.RandomState(0)
    size = 20
    a = rng.normal(3, 0.5, size=size)
    cats = np.asarray(["b1", "b2"])
    b = cats[rng.choice(cats.size, size, p=[0.5, 0.5])]
    c = -4.2 + 1.2 * a + rng.normal(0, 0.75, size=size)
    d = 1.5 - 0.3 * c + rng.normal(0, 0.5, size=size)
    df = pd.DataFrame({"a": a, "b": pd.Series(b, dtype="category"), "c": c, "d": d})
    het = pbn.HeterogeneousBN({pa.float64(): [MyLGType()], pa.float32(): [MyLGType()],
                               pa.dictionary(pa.int8(), pa.utf8()): [pbn.DiscreteFactorType()]}, ABCD, [("a", "c")])
    het.set_node_type("a", pbn.CKDEType())
    het.fit(df)
    assert het.node_type("a") == pbn.CKDEType() and het.node_type("b") == pbn.DiscreteFactorType()
    assert het.node_type("c") == MyLGType() and het.node_type("d") == MyLGType()
    assert het.fitted() and np.isfinite(het.slogl(df))


def test_user_network_type_and_class():   # extending.rst:625-640, 786-806, 840-895
    g = BayesianNetwork(MyRestrictedGaussianType(), ABCD)
    g.add_arc("a", "b")
    with pytest.raises(ValueError, match="Cannot add arc b -> c."):
        g.add_arc("b", "c")
    with pytest.raises(ValueError, match="Cannot add arc c -> a."):
        g.add_arc("c", "a")
    with pytest.raises(ValueError, match="Cannot flip arc a -> b."):
        g.flip_arc("a", "b")
    g1, g2 = BayesianNetwork(pbn.GaussianNetworkType(), ABCD), BayesianNetwork(MyRestrictedGaussianType(), ABCD)
    assert type(g1) == type(g2) and type(MyRestrictedBN(ABCD)) != type(g1)

    bn = MyRestrictedBN(ABCD)
    bn.add_arc("a", "c")
    assert bn.log == ["Adding arc a -> c"] and bn.has_arc("a", "c")
    df = sample_data(500)
    bn.fit(df)
    assert all(type(bn.cpd(v)) is MyLG for v in ABCD)
    bn.include_cpd = True
    loaded = pickle.loads(pickle.dumps(bn))
    assert type(loaded) is MyRestrictedBN and loaded.extra_data == "extra" and loaded.arcs() == [("a", "c")] and loaded.fitted()
    same_parameters(loaded.cpd("c"), bn.cpd("c"))
    clone = loaded.clone()
    assert type(clone) is MyRestrictedBN and clone.extra_data == "extra" and clone.arcs() == [("a", "c")]
