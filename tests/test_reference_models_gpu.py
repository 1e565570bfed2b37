"""GPU tier: the fit / cpd / add_cpds / logl assertions of the reference's model tests
(/root/reference/tests/models/SemiparametricBN_test.py:99-240, BayesianNetwork_test.py:221-330) re-typed."""
import numpy as np
import pytest

from helpers import frame

pytestmark = pytest.mark.gpu
FULL = [("a", "b"), ("a", "c"), ("a", "d"), ("b", "c"), ("b", "d"), ("c", "d")]


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


@pytest.fixture(scope="module")
def df(golden):
    return frame(golden["train10k"])


def test_spbn_node_type_and_fit(pbn, df):   # SemiparametricBN_test.py:99-153
    spbn = pbn.SemiparametricBN(["a", "b", "c", "d"])
    assert all(spbn.node_type(n) == pbn.UnknownFactorType() for n in spbn.nodes())
    spbn.set_node_type("b", pbn.CKDEType())
    assert spbn.node_type("b") == pbn.CKDEType()
    spbn.set_node_type("b", pbn.LinearGaussianCPDType())
    assert spbn.node_type("b") == pbn.LinearGaussianCPDType()

    spbn = pbn.SemiparametricBN(FULL)
    with pytest.raises(ValueError, match="not added"):
        spbn.cpd("a")
    spbn.fit(df)
    for n in spbn.nodes():
        cpd = spbn.cpd(n)
        assert cpd.type() == pbn.LinearGaussianCPDType() and type(cpd) == pbn.LinearGaussianCPD
        assert cpd.variable() == n and set(cpd.evidence()) == set(spbn.parents(n))
    spbn.fit(df)
    spbn.remove_arc("a", "b")
    cpd_b = spbn.cpd("b")
    assert type(cpd_b) == pbn.LinearGaussianCPD and cpd_b.evidence() != spbn.parents("b")
    spbn.fit(df)
    cpd_b = spbn.cpd("b")
    assert type(cpd_b) == pbn.LinearGaussianCPD and cpd_b.evidence() == spbn.parents("b")
    spbn.set_node_type("c", pbn.CKDEType())
    with pytest.raises(ValueError, match="not added"):
        spbn.cpd("c")
    spbn.fit(df)
    assert spbn.cpd("c").type() == spbn.node_type("c") == pbn.CKDEType()


def test_spbn_cpd_and_add_cpds(pbn, df):   # SemiparametricBN_test.py:155-203
    spbn = pbn.SemiparametricBN(FULL, [("d", pbn.CKDEType())])
    with pytest.raises(ValueError, match="not added"):
        spbn.cpd("a")
    spbn.fit(df)
    assert [spbn.cpd(v).type() for v in "abcd"] == [pbn.LinearGaussianCPDType()] * 3 + [pbn.CKDEType()]
    assert all(spbn.cpd(v).fitted() for v in "abcd")

    spbn = pbn.SemiparametricBN(FULL, [("d", pbn.CKDEType())])
    assert spbn.node_type("a") == pbn.UnknownFactorType()
    spbn.add_cpds([pbn.CKDE("a", [])])
    assert spbn.node_type("a") == pbn.CKDEType()
    with pytest.raises(ValueError, match="Bayesian network expects type"):
        spbn.add_cpds([pbn.LinearGaussianCPD("d", ["a", "b", "c"])])
    lg = pbn.LinearGaussianCPD("b", ["a"], [2.5, 1.65], 4)
    ckde = pbn.CKDE("d", ["a", "b", "c"])
    assert lg.fitted() and not ckde.fitted()
    spbn.add_cpds([lg, ckde])
    spbn.set_node_type("a", pbn.UnknownFactorType())
    with pytest.raises(ValueError, match='CPD of variable "a" not added. Call add_cpds\\(\\) or fit\\(\\) to add the CPD.'):
        spbn.cpd("a")
    assert spbn.cpd("b").fitted()
    with pytest.raises(ValueError, match='CPD of variable "c" not added'):
        spbn.cpd("c")
    assert not spbn.cpd("d").fitted()


@pytest.mark.parametrize("kind", ["spbn", "gbn", "mixed"])
def test_network_logl_is_sum_of_factors(pbn, df, golden, kind):   # SemiparametricBN_test.py:205-230, BayesianNetwork_test.py:300-322
    if kind == "gbn":
        net = pbn.GaussianNetwork(FULL)
    elif kind == "spbn":
        net = pbn.SemiparametricBN(FULL)
    else:
        net = pbn.SemiparametricBN(FULL, [("a", pbn.CKDEType()), ("c", pbn.CKDEType())])
    net.fit(df)
    test_df = frame(golden["train500"])
    ll, sll = net.logl(test_df), net.slogl(test_df)
    sum_ll, sum_sll = np.zeros(test_df.shape[0]), 0.0
    for n in net.nodes():
        cpd = net.cpd(n)
        l, s = cpd.logl(test_df), cpd.slogl(test_df)
        assert np.isclose(s, l.sum())
        sum_ll += l
        sum_sll += s
    assert np.all(np.isclose(ll, sum_ll)) and np.isclose(sll, ll.sum()) and sll == pytest.approx(sum_sll, rel=1e-12)
    s = net.sample(100, 0, ordered=True)   # BayesianNetwork_test.py:324-340: shape and column order
    assert s.num_rows == 100 and s.schema.names == net.nodes()


def test_lg_cdf_and_sample(pbn, df, golden):   # LinearGaussianCPD_test.py:211-290
    from scipy.stats import norm

    test_df = frame(golden["train500"])
    np.random.seed(0)
    tn = test_df.copy()
    for c in "abcd":
        tn.loc[tn.index[np.random.randint(0, tn.shape[0], size=20)], c] = np.nan
    for variable, evidence in [("a", []), ("b", ["a"]), ("c", ["a", "b"]), ("d", ["a", "b", "c"])]:
        cpd = pbn.LinearGaussianCPD(variable, evidence)
        cpd.fit(df)
        want = lambda t: norm.cdf(t[variable], cpd.beta[0] + (t[evidence].to_numpy() @ cpd.beta[1:] if evidence else 0.0), np.sqrt(cpd.variance))
        assert np.all(np.isclose(cpd.cdf(test_df), want(test_df)))
        got = cpd.cdf(tn)
        nulls = tn[[variable] + evidence].isna().any(axis=1).to_numpy()
        assert np.array_equal(np.isnan(got), nulls) and np.all(np.isclose(got[~nulls], want(tn)[~nulls]))
        s = cpd.sample(1000, test_df.iloc[np.arange(1000) % test_df.shape[0]][evidence] if evidence else None, 0)
        assert len(s) == 1000 and str(s.type) == "double"
    a, b = pbn.LinearGaussianCPD("d", ["a", "b", "c"]), pbn.LinearGaussianCPD("d", ["c", "a", "b"])
    a.fit(df)
    b.fit(df)
    assert np.all(np.isclose(a.cdf(test_df), b.cdf(test_df)))


@pytest.mark.parametrize("kind", ["gbn", "mixed"])
def test_network_logl_vs_oracle_factors(pbn, df, golden, kind):
    """BayesianNetwork::logl / slogl (models/BayesianNetwork.hpp:930-958: the sum of the factors' log-likelihoods) against the
    ORACLE's factors on the golden table - LinearGaussianCPD by the restated MLE (mle_LinearGaussianCPD.hpp:11-193) + normal
    log-density, CKDE by the restated normal-reference bandwidth + per-pair sums - not against the product's own factor objects
    (test_network_logl_is_sum_of_factors above)."""
    from oracle import oracle

    types = [("a", pbn.CKDEType()), ("c", pbn.CKDEType())] if kind == "mixed" else []
    net = pbn.SemiparametricBN(FULL, types) if kind == "mixed" else pbn.GaussianNetwork(FULL)
    net.fit(df)
    test_df = frame(golden["train500"])
    tr, te = df, test_df
    want = np.zeros(te.shape[0])
    for v in net.nodes():
        cols = [v] + list(net.parents(v))
        if net.node_type(v) == pbn.CKDEType():
            H = oracle.nr_bandwidth(tr[cols].to_numpy())
            want += oracle.ckde_logl(tr[cols].to_numpy(), H, te[cols].to_numpy())
        else:
            beta, var = oracle.lg_fit(tr[cols].to_numpy())
            want += oracle.lg_logl(te[cols].to_numpy(), beta, var)
    got = net.logl(test_df)
    assert np.allclose(got, want, rtol=1e-8, atol=1e-8)
    assert abs(net.slogl(test_df) - want.sum()) <= 1e-9 * abs(want.sum())


def test_network_sample_order_is_pinned(pbn, df):
    """BNGeneric::sample (BayesianNetwork.hpp:960-994) seeds node i of the topological order with seed + i.  The reference's order
    comes out of libstdc++ unordered_sets (roots, children: generic_graph.hpp:2659-2710) and depends on the graph's edit history;
    here the order is the documented index-ordered one (models.BayesianNetwork.topological_sort).  This pins it: the joint sample
    is reproduced column by column from the factors' own samplers with seed + position in THAT order - per-factor parity with the
    reference's samplers is tested in test_sampling_gpu.py, the order is a documented divergence."""
    import pyarrow as pa

    net = pbn.GaussianNetwork(["a", "b", "c", "d"], [("c", "a"), ("a", "b"), ("d", "b")])
    net.fit(df)
    order = net.topological_sort()
    assert order == ["d", "c", "a", "b"]          # stack-driven Kahn: roots in index order (c, d) pushed, d popped first
    s = net.sample(257, 11)
    assert s.schema.names == order                # unordered output: columns in sampling order
    cols = {}
    for i, v in enumerate(order):
        ev = pa.RecordBatch.from_arrays([pa.array(cols[p]) for p in net.parents(v)], names=list(net.parents(v))) if net.parents(v) else None
        cols[v] = np.asarray(net.cpd(v).sample(257, ev, 11 + i))
        assert np.array_equal(s.column(i).to_numpy(), cols[v]), v


@pytest.mark.parametrize("kind", ["gbn", "spbn", "kdebn"])
def test_shared_upload_gives_the_per_factor_values(pbn, df, golden, kind):
    """The one-upload scope of BayesianNetwork.fit / logl / slogl (dataset.shared_upload) addresses columns of the whole table by index:
    parameters, per-row values and sums are bit-identical to factors fitted / evaluated one by one on their own uploads; columns with
    nulls fall back to the per-factor path."""
    import pyarrow as pa

    from pybnesian_amd.dataset import DeviceTable, as_record_batch, default_context, shared_upload

    nodes = ["d", "b", "a", "c"]   # node order differs from column order
    arcs = [("a", "b"), ("a", "c"), ("b", "d"), ("c", "d")]
    types = {"gbn": [], "spbn": [("d", pbn.CKDEType()), ("c", pbn.CKDEType())], "kdebn": []}[kind]
    make = {"gbn": lambda: pbn.GaussianNetwork(nodes, arcs), "spbn": lambda: pbn.SemiparametricBN(nodes, arcs, types),
            "kdebn": lambda: pbn.KDENetwork(nodes, arcs)}[kind]
    rb = as_record_batch(df)
    test = as_record_batch(frame(golden["train500"]))
    net = make()
    net.fit(rb)
    lone = {}
    for n in nodes:                    # the same factors, each on its own upload
        f = type(net.cpd(n))(n, net.parents(n))
        f.fit(rb)
        lone[n] = f
    with shared_upload(default_context(), rb) as table:
        assert table is not None and DeviceTable.from_dataframe(default_context(), rb, ["c", "a"])[0] is table
    assert DeviceTable.from_dataframe(default_context(), rb, ["c", "a"])[0].names == ["c", "a"]   # outside the scope: its own table
    total = np.zeros(test.num_rows)
    for n in nodes:
        f = net.cpd(n)
        if hasattr(f, "beta"):
            assert np.array_equal(f.beta, lone[n].beta) and f.variance == lone[n].variance
        total += lone[n].logl(test)
    assert np.array_equal(net.logl(test), total)
    assert net.slogl(test) == pytest.approx(sum(lone[n].slogl(test) for n in nodes), rel=1e-14)
    # nulls in one column: that column leaves the shared table, everything still agrees with the per-factor path
    a = test.column(test.schema.get_field_index("a")).to_numpy().copy()
    mask = np.zeros(a.size, bool)
    mask[::7] = True
    holes = pa.RecordBatch.from_arrays([pa.array(a, mask=mask) if f.name == "a" else test.column(i) for i, f in enumerate(test.schema)],
                                       names=test.schema.names)
    total = np.zeros(holes.num_rows)
    for n in nodes:
        total += lone[n].logl(holes)
    got = net.logl(holes)
    assert np.array_equal(np.isnan(got), np.isnan(total)) and np.array_equal(got[~np.isnan(total)], total[~np.isnan(total)])
