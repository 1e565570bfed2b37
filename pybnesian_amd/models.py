"""Minimal Bayesian-network model surface needed by the score / hill-climbing path.

Only what `Score.local_score`, the operator sets and `GreedyHillClimbing` touch in the reference
(/root/reference/pybnesian/models/BayesianNetwork.hpp:29-145,571-577,662-681, SemiparametricBN.hpp:93-119,
graph/generic_graph.hpp:2711-2745): an ordered node list, a DAG with the reference's add / flip legality
predicates, and a node-type table (LinearGaussianCPD / CKDE).  Everything else of `models/` is out of scope.
"""


class FactorType:
    _name = "FactorType"

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self).__name__)

    def __str__(self):
        return self._name

    __repr__ = __str__


class LinearGaussianCPDType(FactorType):
    _name = "LinearGaussianFactor"


class CKDEType(FactorType):
    _name = "CKDEFactor"


class DiscreteFactorType(FactorType):
    _name = "DiscreteFactor"


class UnknownFactorType(FactorType):
    """Node type of a heterogeneous network before it has seen data (factors/factors.hpp:60-90): resolved from the
    column's data type by set_unknown_node_types / fit / the hill-climb (hillclimbing.hpp:81-93)."""

    _name = "UnknownFactor"


class BayesianNetworkType:
    _name = "BayesianNetworkType"
    homogeneous = True
    default_type = LinearGaussianCPDType()

    def __eq__(self, other):
        return type(self) is type(other)

    def __str__(self):
        return self._name


class GaussianNetworkType(BayesianNetworkType):
    _name = "GaussianNetworkType"


class KDENetworkType(BayesianNetworkType):
    _name = "KDENetworkType"
    default_type = CKDEType()


class SemiparametricBNType(BayesianNetworkType):
    _name = "SemiparametricBNType"
    homogeneous = False
    default_type = UnknownFactorType()


class CLGNetworkType(BayesianNetworkType):
    """models/CLGNetwork.hpp: discrete nodes are DiscreteFactor, continuous nodes LinearGaussianCPD (conditional
    on their discrete parents); no continuous -> discrete arcs (CLGNetwork.hpp:84-89)."""

    _name = "CLGNetworkType"
    homogeneous = False
    default_type = UnknownFactorType()


class BayesianNetwork:
    """BayesianNetwork / ConditionalBayesianNetwork (models/BayesianNetwork.hpp): with `interface_nodes` the network is
    conditional - interface nodes can be parents of the nodes but have no parents, factors or scores of their own."""

    def __init__(self, bn_type, nodes, arcs=(), node_types=(), interface_nodes=()):
        if nodes and isinstance(nodes[0], (tuple, list)) and not arcs:
            # constructed from arcs only: nodes in order of first appearance
            arcs = nodes
            seen = []
            for s, t in arcs:
                for x in (s, t):
                    if x not in seen:
                        seen.append(x)
            nodes = seen
        self._type = bn_type
        self._nodes = list(nodes)
        self._interface = list(interface_nodes)
        joint = self._nodes + self._interface
        if len(set(joint)) != len(joint):
            raise ValueError("Nodes must be unique.")
        self._index = {n: i for i, n in enumerate(joint)}
        self._parents = {n: [] for n in joint}
        self._children = {n: [] for n in joint}
        self._types = {n: bn_type.default_type for n in joint}
        for n, t in node_types:
            self.set_node_type(n, t)
        for s, t in arcs:
            self.add_arc(s, t)

    # -- structure -----------------------------------------------------------------------------------
    def type(self):
        return self._type

    def nodes(self):
        return list(self._nodes)

    def num_nodes(self):
        return len(self._nodes)

    def index(self, node):
        return self._index[node]

    def name(self, idx):
        return self._nodes[idx]

    def contains_node(self, node):
        return node in self._index and node not in self._interface

    def _reindex(self):
        self._index = {n: i for i, n in enumerate(self._nodes + self._interface)}

    def add_node(self, node):
        """BNGeneric::add_node (BayesianNetwork.hpp:330-345): a new isolated node with the network's default type."""
        if node in self._index:
            raise ValueError(f"Cannot add node {node} because a node with the same name already exists.")
        self._nodes.append(node)
        self._parents[node], self._children[node] = [], []
        self._types[node] = self._type.default_type
        self._reindex()
        if getattr(self, "_cpds", None):
            self._cpds.pop(node, None)
        return self._index[node]

    def remove_node(self, node):
        """BNGeneric::remove_node: the node, its arcs and the factors that conditioned on it go."""
        if node not in self._index or node in self._interface:
            raise ValueError(f"Node {node} not present in the Bayesian network.")
        for c in list(self._children[node]):
            self._parents[c].remove(node)
            if getattr(self, "_cpds", None):
                self._cpds.pop(c, None)
        for p in list(self._parents[node]):
            self._children[p].remove(node)
        self._nodes.remove(node)
        for d in (self._parents, self._children, self._types):
            d.pop(node)
        if getattr(self, "_cpds", None):
            self._cpds.pop(node, None)
        self._reindex()

    def add_interface_node(self, node):
        if node in self._index:
            raise ValueError(f"Cannot add node {node} because a node with the same name already exists.")
        self._interface.append(node)
        self._parents[node], self._children[node] = [], []
        self._types[node] = self._type.default_type
        self._reindex()

    def remove_interface_node(self, node):
        if node not in self._interface:
            raise ValueError(f"Interface node {node} not present in the Bayesian network.")
        for c in list(self._children[node]):
            self._parents[c].remove(node)
            if getattr(self, "_cpds", None):
                self._cpds.pop(c, None)
        self._interface.remove(node)
        for d in (self._parents, self._children, self._types):
            d.pop(node)
        self._reindex()

    # -- conditional networks (ConditionalBayesianNetworkBase, BayesianNetwork.hpp:140-222) ---------------------
    def interface_nodes(self):
        return list(self._interface)

    def joint_nodes(self):
        return self._nodes + self._interface

    def num_interface_nodes(self):
        return len(self._interface)

    def num_joint_nodes(self):
        return len(self._nodes) + len(self._interface)

    def is_interface(self, node):
        return node in self._interface

    def contains_interface_node(self, node):
        return node in self._interface

    def contains_joint_node(self, node):
        return node in self._index

    def conditional_bn(self, nodes=None, interface_nodes=None):
        """BNGeneric::conditional_bn (BayesianNetwork.hpp:1064-1100): the same arcs and node types over a new split of
        the variables into nodes and interface nodes (arcs into interface nodes are dropped)."""
        if nodes is None:
            nodes, interface_nodes = self._nodes, self._interface
        interface_nodes = list(interface_nodes or [])
        keep = set(nodes) | set(interface_nodes)
        arcs = [(s, t) for s, t in self.arcs() if s in keep and t in set(nodes)]
        types = [(n, t) for n, t in self._types.items() if n in keep]
        return BayesianNetwork(self._type, list(nodes), arcs, [] if self._type.homogeneous else types, interface_nodes)

    def unconditional_bn(self):
        types = list(self._types.items())
        return BayesianNetwork(self._type, self.joint_nodes(), self.arcs(), [] if self._type.homogeneous else types)

    def arcs(self):
        return [(p, n) for n in self._nodes for p in self._parents[n]]

    def num_arcs(self):
        return sum(len(v) for v in self._parents.values())

    def parents(self, node):
        return list(self._parents[node])

    def num_parents(self, node):
        return len(self._parents[node])

    def children(self, node):
        return list(self._children[node])

    def has_arc(self, source, target):
        return source in self._parents[target]

    def has_path(self, source, target):
        seen, stack = {source}, [source]
        while stack:
            u = stack.pop()
            for c in self._children[u]:
                if c == target:
                    return True
                if c not in seen:
                    seen.add(c)
                    stack.append(c)
        return False

    def can_have_arc(self, source, target):
        """BayesianNetworkType::can_have_arc (SemiparametricBN.hpp:93-98, CLGNetwork.hpp:84-89)."""
        return not (self._types[target] == DiscreteFactorType() and self._types[source] != DiscreteFactorType())

    def can_add_arc(self, source, target):
        return source != target and target not in self._interface and self.can_have_arc(source, target) and (
            not self._parents[source] or not self._children[target] or not self.has_path(target, source)
        )

    def can_flip_arc(self, source, target):
        if source == target or source in self._interface or target in self._interface or not self.can_have_arc(target, source):
            return False
        if self.has_arc(source, target):
            if len(self._parents[target]) == 1 or len(self._children[source]) == 1:
                return True
            self._children[source].remove(target)
            try:
                return not self.has_path(source, target)
            finally:
                self._children[source].append(target)
        if not self._parents[target] or not self._children[source]:
            return True
        return not self.has_path(source, target)

    def add_arc(self, source, target):
        if source not in self._index or target not in self._index:
            raise ValueError("Node not present in the Bayesian network.")
        if target in self._interface:
            raise ValueError(f"Interface node {target} cannot have parents.")
        if self.has_arc(source, target):
            return
        if not self.can_add_arc(source, target):
            raise ValueError(f"Cannot add arc {source} -> {target}: it would create a cycle.")
        self._parents[target].append(source)
        self._children[source].append(target)

    def remove_arc(self, source, target):
        if self.has_arc(source, target):
            self._parents[target].remove(source)
            self._children[source].remove(target)

    def flip_arc(self, source, target):
        if self.has_arc(source, target):
            self.remove_arc(source, target)
            self.add_arc(target, source)

    # -- node types -----------------------------------------------------------------------------------
    def node_type(self, node):
        return self._types[node]

    def node_types(self):
        return dict(self._types)

    def has_unknown_node_types(self):
        return any(t == UnknownFactorType() for t in self._types.values())

    def underlying_node_type(self, df, node):
        """BNGeneric::underlying_node_type (BayesianNetwork.hpp:662-681): the node's type, or for an unknown one the
        first default of the network type for the column's data type (SemiparametricBN.hpp:60-75, CLGNetwork.hpp:55-68:
        float -> LinearGaussianCPD, dictionary -> DiscreteFactor)."""
        import pyarrow as pa

        from .dataset import as_record_batch

        t = self._types[node]
        if t != UnknownFactorType():
            return t
        rb = as_record_batch(df)
        f = rb.schema.field(node)
        if pa.types.is_dictionary(f.type):
            return DiscreteFactorType()
        if pa.types.is_floating(f.type):
            return LinearGaussianCPDType()
        raise ValueError(f"There is no underlying FactorType for node {node} as there is no valid FactorType for DataType {f.type}")

    def set_unknown_node_types(self, df, type_blacklist=()):
        """BNGeneric::set_unknown_node_types (BayesianNetwork.hpp:700-745)."""
        import pyarrow as pa

        from .dataset import as_record_batch

        rb = as_record_batch(df)
        black = {(n, t) for n, t in type_blacklist}
        for node, t in list(self._types.items()):
            if t != UnknownFactorType():
                continue
            f = rb.schema.field(node)
            if pa.types.is_dictionary(f.type):
                options = [DiscreteFactorType()]
            elif isinstance(self._type, CLGNetworkType):
                options = [LinearGaussianCPDType()]
            else:
                options = [LinearGaussianCPDType(), CKDEType()]
            options = [o for o in options if (node, o) not in black]
            if not options:
                raise ValueError(f"There is no valid FactorType for node {node} (all the defaults are blacklisted).")
            self._types[node] = options[0]

    def set_node_type(self, node, node_type):
        if isinstance(self._type, CLGNetworkType) and node_type == CKDEType():
            raise ValueError(f"Wrong factor type \"{node_type}\" for node \"{node}\" in Bayesian network type \"{self._type}\".")
        if self._type.homogeneous and node_type != self._type.default_type:
            raise ValueError(f"Wrong factor type \"{node_type}\" for node \"{node}\" in Bayesian network type \"{self._type}\".")
        if self._types.get(node) != node_type and getattr(self, "_cpds", None):
            self._cpds.pop(node, None)   # a factor of the old type no longer belongs to the node (BayesianNetwork.hpp:770-790)
        self._types[node] = node_type

    def clone(self):
        return BayesianNetwork(self._type, self._nodes, self.arcs(), list(self._types.items()), self._interface)

    # -- parameters: BayesianNetwork::fit / logl / slogl (models/BayesianNetwork.hpp:960-994) -------------------
    def _new_factor(self, df, node):
        """FactorType::new_factor (CKDE.cpp:15-41, LinearGaussianCPD.cpp:20-46): discrete evidence selects the
        DiscreteAdaptator variant."""
        import pyarrow as pa

        from .dataset import as_record_batch
        from .factors import CKDE, HCKDE, CLinearGaussianCPD, DiscreteFactor, LinearGaussianCPD

        rb = as_record_batch(df)
        parents = self.parents(node)
        is_disc = lambda v: pa.types.is_dictionary(rb.schema.field(v).type)
        nt = DiscreteFactorType() if is_disc(node) else self._types[node]
        if nt == UnknownFactorType():
            nt = LinearGaussianCPDType()
        if nt == DiscreteFactorType():
            return DiscreteFactor(node, parents)
        hybrid = any(is_disc(p) for p in parents)
        if nt == CKDEType():
            return HCKDE(node, parents) if hybrid else CKDE(node, parents)
        return CLinearGaussianCPD(node, parents) if hybrid else LinearGaussianCPD(node, parents)

    def _cpd_valid(self, node):
        """must_construct_cpd (BayesianNetwork.hpp:905-935): a factor is kept when its type and evidence still match."""
        f = getattr(self, "_cpds", {}).get(node)
        if f is None:
            return False
        t = self._types[node]
        if t != UnknownFactorType() and f.type() != t and not (t == DiscreteFactorType()):
            return False
        return sorted(f.evidence()) == sorted(self._parents[node])

    def fit(self, df):
        """BNGeneric::fit (BayesianNetwork.hpp:960-994): unknown node types are resolved from the data, factors that are
        missing, of the wrong type, with stale evidence or unfitted are (re)built and fitted; the others are kept."""
        if getattr(self, "_cpds", None) is None:
            self._cpds = {}
        if self.has_unknown_node_types():
            self.set_unknown_node_types(df)
        for n in self._nodes:
            if not self._cpd_valid(n):
                self._cpds[n] = self._new_factor(df, n)
            if not self._cpds[n].fitted():
                self._cpds[n].fit(df)

    def fitted(self):
        cp = getattr(self, "_cpds", None)
        return cp is not None and all(n in cp and self._cpd_valid(n) and cp[n].fitted() for n in self._nodes)

    def cpd(self, node):
        if node not in self._index or node in self._interface:
            raise ValueError(f"Node {node} not present in the Bayesian network.")
        cp = getattr(self, "_cpds", None)
        if not cp or node not in cp:
            raise ValueError(f"CPD of variable \"{node}\" not added. Call add_cpds() or fit() to add the CPD.")
        return cp[node]

    def logl(self, df):
        if not self.fitted():
            raise ValueError("Model not fitted.")  # BayesianNetwork.hpp check_fitted
        import numpy as np

        out = None
        for n in self._nodes:
            ll = self._cpds[n].logl(df)
            out = ll if out is None else out + ll
        return out if out is not None else np.zeros(0)

    def slogl(self, df):
        if not self.fitted():
            raise ValueError("Model not fitted.")
        return float(sum(self._cpds[n].slogl(df) for n in self._nodes))

    # -- graph queries of models/BayesianNetwork.hpp / graph/generic_graph.hpp the callers of the hot path use -------
    def num_children(self, node):
        return len(self._children[node])

    def roots(self):
        return {n for n in self._nodes if not self._parents[n]}

    def leaves(self):
        return {n for n in self._nodes if not self._children[n]}

    def indices(self):
        return dict(self._index)

    collapsed_indices = indices

    def collapsed_index(self, node):
        return self._index[node]

    def collapsed_name(self, idx):
        return self._nodes[idx]

    def is_valid(self, idx):
        return 0 <= idx < len(self._nodes)

    def topological_sort(self):
        """DagImpl::topological_sort (graph/generic_graph.hpp:2659-2710): stack-driven Kahn order.  The reference
        walks libstdc++ unordered_sets (roots, children), whose iteration order depends on the graph's edit history;
        here roots and children are visited in node-index order, so the result is a valid, deterministic order that
        need not be the reference's."""
        incoming = {n: sum(1 for p in self._parents[n] if p not in self._interface) for n in self._nodes}
        stack = [n for n in self._nodes if incoming[n] == 0]
        order = []
        while stack:
            u = stack.pop()
            order.append(u)
            for c in sorted(self._children[u], key=self._index.__getitem__):
                if c not in incoming:
                    continue
                incoming[c] -= 1
                if incoming[c] == 0:
                    stack.append(c)
        if len(order) != len(self._nodes):
            raise ValueError("Graph must be a DAG to obtain a topological sort.")
        return order

    def add_cpds(self, cpds):
        """BayesianNetwork::add_cpds (BayesianNetwork.hpp:845-905): fitted factors whose evidence equals the parents."""
        for f in cpds:
            v = f.variable()
            if v not in self._index:
                raise ValueError(f"CPD defined on variable which is not present in the model:\n{f}")
            if sorted(f.evidence()) != sorted(self._parents[v]):
                raise ValueError(f"CPD do not have the model's parent set as evidence:\n{f}")
            t = self._types[v]
            if t != UnknownFactorType() and f.type() != t:
                raise ValueError(f"Bayesian network expects type {t} for node {v}, but {f.type()} was provided.")
        if getattr(self, "_cpds", None) is None:
            self._cpds = {}
        for f in cpds:
            if self._types[f.variable()] == UnknownFactorType():
                self._types[f.variable()] = f.type()
            self._cpds[f.variable()] = f

    def sample(self, n, seed=None, ordered=False, concat_evidence=False):
        """BNGeneric::sample (BayesianNetwork.hpp:1023-1062): ancestral sampling in topological order, node i of that
        order with seed + i; returns a pyarrow.RecordBatch (columns in topological order, or in nodes() order when
        `ordered`).  Conditional networks take the evidence table of the interface nodes instead of `n`
        (ConditionalBayesianNetwork::sample(evidence, seed, concat_evidence, ordered), :1165-1215)."""
        import pyarrow as pa

        from .factors import _random_seed

        names, arrays = [], []
        if self._interface:
            from .dataset import as_record_batch

            ev = as_record_batch(n)
            if any(ev.schema.get_field_index(v) < 0 for v in self._interface):
                raise ValueError("Evidence DataFrame does not contain all the interface nodes.")
            n = ev.num_rows
            names = list(self._interface)
            arrays = [ev.column(ev.schema.get_field_index(v)) for v in names]
        elif not isinstance(n, int):
            raise ValueError("n should be an integer number of samples")
        if n < 0:
            raise ValueError("n should be a non-negative number")
        if not self.fitted():
            raise ValueError("Model not fitted.")
        seed = _random_seed() if seed is None else int(seed)
        for i, node in enumerate(self.topological_sort()):
            parents = pa.RecordBatch.from_arrays(arrays, names=names) if arrays else None
            arrays.append(self._cpds[node].sample(n, parents, (seed + i) & 0xFFFFFFFF))
            names.append(node)
        pos = {nm: i for i, nm in enumerate(names)}
        out = list(self._nodes) if ordered else [nm for nm in names if nm not in self._interface]
        if self._interface and concat_evidence:
            out = out + list(self._interface)
        return pa.RecordBatch.from_arrays([arrays[pos[nm]] for nm in out], names=out)

    def save(self, name, include_cpd=False):
        """BayesianNetwork::save (BayesianNetwork.hpp:643, util/pickle.hpp): pickle to `name`.pickle."""
        import pickle

        self._include_cpd = bool(include_cpd)
        with open(name if name.endswith(".pickle") else name + ".pickle", "wb") as f:
            pickle.dump(self, f, protocol=2)

    def __getstate__(self):
        state = {"type": self._type, "nodes": self._nodes, "arcs": self.arcs(), "types": list(self._types.items()),
                 "interface": self._interface}
        if getattr(self, "_include_cpd", False) and self.fitted():
            state["cpds"] = [self._cpds[n] for n in self._nodes]
        return state

    def __setstate__(self, state):
        self.__init__(state["type"], state["nodes"], state["arcs"], state["types"], state.get("interface", ()))
        if "cpds" in state:
            self._cpds = {f.variable(): f for f in state["cpds"]}
            self._include_cpd = True

    @property
    def include_cpd(self):
        return getattr(self, "_include_cpd", False)

    @include_cpd.setter
    def include_cpd(self, value):
        self._include_cpd = bool(value)

    def __str__(self):
        return f"{self._type} with {self.num_nodes()} nodes and {self.num_arcs()} arcs"


def GaussianNetwork(nodes, arcs=()):
    return BayesianNetwork(GaussianNetworkType(), nodes, arcs)


def KDENetwork(nodes, arcs=()):
    return BayesianNetwork(KDENetworkType(), nodes, arcs)


def SemiparametricBN(nodes, arcs=(), node_types=()):
    if arcs and isinstance(arcs[0], (tuple, list)) and len(arcs[0]) == 2 and isinstance(arcs[0][1], FactorType):
        node_types, arcs = arcs, ()
    return BayesianNetwork(SemiparametricBNType(), nodes, arcs, node_types)


def CLGNetwork(nodes, arcs=(), node_types=()):
    return BayesianNetwork(CLGNetworkType(), nodes, arcs, node_types)


def ConditionalBayesianNetwork(bn_type, nodes, interface_nodes, arcs=(), node_types=()):
    return BayesianNetwork(bn_type, list(nodes), arcs, node_types, list(interface_nodes))


def ConditionalGaussianNetwork(nodes, interface_nodes, arcs=()):
    return BayesianNetwork(GaussianNetworkType(), list(nodes), arcs, (), list(interface_nodes))


def ConditionalKDENetwork(nodes, interface_nodes, arcs=()):
    return BayesianNetwork(KDENetworkType(), list(nodes), arcs, (), list(interface_nodes))


def ConditionalSemiparametricBN(nodes, interface_nodes, arcs=(), node_types=()):
    if arcs and isinstance(arcs[0], (tuple, list)) and len(arcs[0]) == 2 and isinstance(arcs[0][1], FactorType):
        node_types, arcs = arcs, ()
    return BayesianNetwork(SemiparametricBNType(), list(nodes), arcs, node_types, list(interface_nodes))


def ConditionalCLGNetwork(nodes, interface_nodes, arcs=(), node_types=()):
    return BayesianNetwork(CLGNetworkType(), list(nodes), arcs, node_types, list(interface_nodes))


def load(name):
    """pybnesian.load (pybindings/lib.cpp, util/pickle.hpp): un-pickle a saved model / factor."""
    import pickle

    with open(name, "rb") as f:
        return pickle.load(f)
