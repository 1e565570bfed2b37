"""The Bayesian-network model surface around the score / hill-climbing path.

What `Score.local_score`, the operator sets, `GreedyHillClimbing` / `MMHC` and the model-level fit / logl / sample /
pickle calls touch in the reference (/root/reference/pybnesian/models/BayesianNetwork.hpp:29-300,571-1260,
SemiparametricBN.hpp, CLGNetwork.hpp, HomogeneousBN.hpp, HeterogeneousBN.hpp, graph/generic_graph.hpp:2659-2745): an
ordered node list, a DAG with the reference's add / flip legality predicates, a node-type table, and the
BayesianNetworkType protocol (subclassable, as through the reference's trampolines) that decides node types and arc
legality.  The numbers all come from the factors (pybnesian_amd.factors), i.e. from the device.
"""


class FactorType:
    """factors/factors.hpp:20-58.  Subclassable; two factor types are equal when they are of the same class."""

    _name = None

    def __init__(self):
        pass

    def new_factor(self, model, variable, evidence, *args, **kwargs):
        raise RuntimeError('Tried to call pure virtual function "FactorType::new_factor"')

    def __eq__(self, other):
        return type(self) is type(other)

    def __hash__(self):
        return hash(type(self).__name__)

    def __str__(self):
        return self._name if self._name is not None else type(self).__name__

    def __repr__(self):
        return self.__str__()


def _has_discrete_evidence(model, evidence):
    return model is not None and any(model.node_type(e) == DiscreteFactorType() for e in evidence)


class LinearGaussianCPDType(FactorType):
    _name = "LinearGaussianFactor"

    def new_factor(self, model, variable, evidence, *args, **kwargs):
        """LinearGaussianCPDType::new_factor (LinearGaussianCPD.cpp:20-46)."""
        from .factors import CLinearGaussianCPD, LinearGaussianCPD

        if _has_discrete_evidence(model, evidence):
            return CLinearGaussianCPD(variable, list(evidence))
        return LinearGaussianCPD(variable, list(evidence), *args, **kwargs)


class CKDEType(FactorType):
    _name = "CKDEFactor"

    def new_factor(self, model, variable, evidence, *args, **kwargs):
        """CKDEType::new_factor (CKDE.cpp:15-41)."""
        from .factors import CKDE, HCKDE

        if _has_discrete_evidence(model, evidence):
            return HCKDE(variable, list(evidence))
        return CKDE(variable, list(evidence), *args, **kwargs)


class DiscreteFactorType(FactorType):
    _name = "DiscreteFactor"

    def new_factor(self, model, variable, evidence, *args, **kwargs):
        from .factors import DiscreteFactor

        return DiscreteFactor(variable, list(evidence))


class UnknownFactorType(FactorType):
    """Node type of a heterogeneous network before it has seen data (factors/factors.hpp:60-90): resolved from the
    column's data type by set_unknown_node_types / fit / the hill-climb (hillclimbing.hpp:81-93)."""

    _name = "UnknownFactor"


class BayesianNetworkType:
    """models/BayesianNetwork.hpp:224-300.  Subclass it to define a new family of networks: `is_homogeneous`,
    `default_node_type` (homogeneous) or `data_default_node_type(dt)` (heterogeneous), and optionally
    `compatible_node_type(model, node, type)`, `can_have_arc(model, source, target)`, `alternative_node_type(model,
    node)`, `new_bn(nodes)`, `new_cbn(nodes, interface_nodes)`.  Two types are equal when they are of the same class
    (the reference hashes the Python type object)."""

    def __init__(self):
        pass

    def is_homogeneous(self):
        raise NotImplementedError(f"{type(self).__name__}.is_homogeneous() is not implemented.")

    def default_node_type(self):
        raise RuntimeError(f"default_node_type() for {self} is not defined.")

    def data_default_node_type(self, dt):
        raise RuntimeError(f"data_default_node_type() for {self} is not defined.")

    def compatible_node_type(self, model, node, node_type):
        return True

    def can_have_arc(self, model, source, target):
        return True

    def alternative_node_type(self, model, node):
        return []

    def new_bn(self, nodes):
        return BayesianNetwork(self, list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalBayesianNetwork(self, list(nodes), list(interface_nodes))

    def _key(self):
        return (type(self),)

    def __eq__(self, other):
        return isinstance(other, BayesianNetworkType) and self._key() == other._key()

    def __ne__(self, other):
        return not self.__eq__(other)

    def __hash__(self):
        return hash(self._key())

    def __str__(self):
        return type(self).__name__

    def __repr__(self):
        return self.__str__()

    # the attribute spelling the engine binding reads
    @property
    def homogeneous(self):
        return bool(self.is_homogeneous())


def _is_float(dt):
    import pyarrow as pa

    return pa.types.is_floating(dt)


def _is_dict(dt):
    import pyarrow as pa

    return pa.types.is_dictionary(dt)


class GaussianNetworkType(BayesianNetworkType):
    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return LinearGaussianCPDType()

    def new_bn(self, nodes):
        return GaussianNetwork(list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalGaussianNetwork(list(nodes), list(interface_nodes))


class KDENetworkType(BayesianNetworkType):
    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return CKDEType()

    def new_bn(self, nodes):
        return KDENetwork(list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalKDENetwork(list(nodes), list(interface_nodes))


class DiscreteBNType(BayesianNetworkType):
    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return DiscreteFactorType()

    def new_bn(self, nodes):
        return DiscreteBN(list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalDiscreteBN(list(nodes), list(interface_nodes))


def _discrete_parents_ok(model, node, node_type):
    if node_type == DiscreteFactorType():
        return all(model.is_interface(p) or model.node_type(p) == DiscreteFactorType() for p in model.parents(node))
    return True


class SemiparametricBNType(BayesianNetworkType):
    """models/SemiparametricBN.hpp:30-125."""

    def is_homogeneous(self):
        return False

    def default_node_type(self):
        raise RuntimeError("default_node_type() for SemiparametricBN is not defined.")

    def data_default_node_type(self, dt):
        if _is_float(dt):
            return [LinearGaussianCPDType(), CKDEType()]
        if _is_dict(dt):
            return [DiscreteFactorType()]
        raise ValueError(f"Data type [{dt}] not compatible with SemiparametricBNType")

    def compatible_node_type(self, model, node, node_type):
        if node_type not in (LinearGaussianCPDType(), CKDEType(), DiscreteFactorType()):
            return False
        return _discrete_parents_ok(model, node, node_type)

    def can_have_arc(self, model, source, target):
        return model.node_type(target) != DiscreteFactorType() or model.node_type(source) == DiscreteFactorType()

    def alternative_node_type(self, model, node):
        t = model.node_type(node)
        if t == LinearGaussianCPDType():
            return [CKDEType()]
        if t == CKDEType():
            return [LinearGaussianCPDType()]
        return []

    def new_bn(self, nodes):
        return SemiparametricBN(list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalSemiparametricBN(list(nodes), list(interface_nodes))

    def __str__(self):
        return "SemiparametricNetworkType"


class CLGNetworkType(BayesianNetworkType):
    """models/CLGNetwork.hpp: discrete nodes are DiscreteFactor, continuous nodes LinearGaussianCPD (conditional
    on their discrete parents); no continuous -> discrete arcs (CLGNetwork.hpp:84-89)."""

    def is_homogeneous(self):
        return False

    def default_node_type(self):
        raise RuntimeError("default_node_type() for CLGNetwork is not defined.")

    def data_default_node_type(self, dt):
        if _is_float(dt):
            return [LinearGaussianCPDType()]
        if _is_dict(dt):
            return [DiscreteFactorType()]
        raise ValueError(f"Data type [{dt}] not compatible with CLGNetworkType")

    def compatible_node_type(self, model, node, node_type):
        if node_type not in (LinearGaussianCPDType(), DiscreteFactorType()):
            return False
        return _discrete_parents_ok(model, node, node_type)

    def can_have_arc(self, model, source, target):
        return model.node_type(target) == LinearGaussianCPDType() or model.node_type(source) != LinearGaussianCPDType()

    def new_bn(self, nodes):
        return CLGNetwork(list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalCLGNetwork(list(nodes), list(interface_nodes))


class HomogeneousBNType(BayesianNetworkType):
    """models/HomogeneousBN.hpp: every node has the one factor type given at construction."""

    def __init__(self, default_factor_type):
        if default_factor_type is None:
            raise ValueError("Default factor_type cannot be null.")
        self._ft = default_factor_type

    def is_homogeneous(self):
        return True

    def default_node_type(self):
        return self._ft

    def new_bn(self, nodes):
        return HomogeneousBN(self._ft, list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalHomogeneousBN(self._ft, list(nodes), list(interface_nodes))

    def _key(self):
        return (type(self), self._ft)

    def __str__(self):
        return f"HomogeneousType({self._ft})"


class HeterogeneousBNType(BayesianNetworkType):
    """models/HeterogeneousBN.hpp:28-150: default factor types either as one list for every column or as a map
    pyarrow data type -> list (order of the lists matters for equality, order of the map does not)."""

    def __init__(self, default_factor_types):
        if isinstance(default_factor_types, dict):
            self._map = {k: list(v) for k, v in default_factor_types.items() if len(v)}
            self._single = None
            if not self._map:
                raise ValueError("Default factor_type cannot be empty.")
            if any(f is None for v in self._map.values() for f in v):
                raise ValueError("Default factor_type cannot contain null FactorType.")
        else:
            self._single = list(default_factor_types)
            self._map = None
            if not self._single:
                raise ValueError("Default factor_type cannot be empty.")
            if any(f is None for f in self._single):
                raise ValueError("Default factor_type cannot contain null FactorType.")

    def is_homogeneous(self):
        return False

    def default_node_type(self):
        raise RuntimeError("default_node_type() for HeterogeneousBN is not defined.")

    def data_default_node_type(self, dt):
        if self._single is not None:
            return list(self._single)
        for k, v in self._map.items():
            if k.equals(dt):
                return list(v)
        raise ValueError(f"Not valid FactorType for DataType {dt}")

    def single_default(self):
        return self._single is not None

    def default_node_types(self):
        return {None: list(self._single)} if self._single is not None else {k: list(v) for k, v in self._map.items()}

    def new_bn(self, nodes):
        return HeterogeneousBN(self._single if self._single is not None else self._map, list(nodes))

    def new_cbn(self, nodes, interface_nodes):
        return ConditionalHeterogeneousBN(self._single if self._single is not None else self._map, list(nodes), list(interface_nodes))

    def _key(self):
        if self._single is not None:
            return (type(self), tuple(self._single))
        return (type(self), frozenset((str(k), tuple(v)) for k, v in self._map.items()))

    def __str__(self):
        return "HeterogeneousBNType"


def _is_arc_list(x):
    return len(x) > 0 and all(isinstance(a, (tuple, list)) and len(a) == 2 and isinstance(a[0], str) and isinstance(a[1], str) for a in x)


def _is_type_list(x):
    return len(x) > 0 and all(isinstance(a, (tuple, list)) and len(a) == 2 and isinstance(a[1], FactorType) for a in x)


def _split_ctor_args(nodes, arcs, node_types):
    """Disentangle the reference's constructor overloads: the first list holds node names or arcs, the second arcs or
    (node, FactorType) pairs."""
    nodes = list(nodes) if nodes is not None else []
    arcs = list(arcs) if arcs is not None else []
    node_types = list(node_types) if node_types is not None else []
    if _is_type_list(arcs) and not node_types:
        arcs, node_types = [], arcs
    if _is_arc_list(nodes) and not arcs:
        arcs, seen = nodes, []
        for s, t in arcs:
            for x in (s, t):
                if x not in seen:
                    seen.append(x)
        nodes = seen
    return nodes, arcs, node_types


class Dag:
    """The directed graph of a network as a value (graph/generic_graph.hpp Dag / ConditionalDag, the part the model
    constructors and pickles use): `BayesianNetwork(type, bn.graph()[, node_types])` rebuilds the structure."""

    def __init__(self, nodes, arcs=(), interface_nodes=()):
        self._nodes, self._arcs, self._interface = list(nodes), [tuple(a) for a in arcs], list(interface_nodes)

    def nodes(self):
        return list(self._nodes)

    def interface_nodes(self):
        return list(self._interface)

    def arcs(self):
        return list(self._arcs)

    def num_nodes(self):
        return len(self._nodes)

    def num_arcs(self):
        return len(self._arcs)

    def has_arc(self, source, target):
        return (source, target) in self._arcs

    def parents(self, node):
        return [s for s, t in self._arcs if t == node]

    def children(self, node):
        return [t for s, t in self._arcs if s == node]

    def __eq__(self, other):
        return isinstance(other, Dag) and (self._nodes, self._interface, sorted(self._arcs)) == (other._nodes, other._interface, sorted(other._arcs))


ConditionalDag = Dag


class BayesianNetworkBase:
    """models/BayesianNetwork.hpp:29-145: the abstract interface of every network (isinstance checks of user code)."""


class ConditionalBayesianNetworkBase(BayesianNetworkBase):
    """models/BayesianNetwork.hpp:147-222."""


class BayesianNetwork(BayesianNetworkBase):
    """BayesianNetwork / ConditionalBayesianNetwork (models/BayesianNetwork.hpp): with `interface_nodes` the network is
    conditional - interface nodes can be parents of the nodes but have no parents, factors or scores of their own."""

    def __init__(self, bn_type, nodes, arcs=(), node_types=(), interface_nodes=()):
        """The reference's overloads (pybindings_models.cpp:2211-2390): (type, nodes), (type, nodes, node_types),
        (type, arcs), (type, arcs, node_types), (type, nodes, arcs), (type, nodes, arcs, node_types)."""
        if bn_type is None:
            raise ValueError("Type of Bayesian network must be non-null.")
        if isinstance(nodes, Dag):   # (type, graph[, node_types])
            if arcs and not node_types:
                node_types = arcs
            if nodes.interface_nodes() and not interface_nodes:
                interface_nodes = nodes.interface_nodes()
            nodes, arcs = nodes.nodes(), nodes.arcs()
        nodes, arcs, node_types = _split_ctor_args(nodes, arcs, node_types)
        self._type = bn_type
        self._nodes = list(nodes)
        self._interface = list(interface_nodes)
        joint = self._nodes + self._interface
        if len(set(joint)) != len(joint):
            raise ValueError("Nodes must be unique.")
        self._index = {n: i for i, n in enumerate(joint)}
        self._parents = {n: [] for n in joint}
        self._children = {n: [] for n in joint}
        default = bn_type.default_node_type() if bn_type.is_homogeneous() else UnknownFactorType()
        self._types = {n: default for n in joint}
        for n, t in node_types:
            if n not in self._index:
                raise ValueError(f"Node {n} not present in the Bayesian network.")
            self.set_node_type(n, t)
        # the graph constructor (generic_graph.hpp:120-160, 2659-2710): arcs by name, then one acyclicity check
        for arc in arcs:
            if not (isinstance(arc, (tuple, list)) and len(arc) == 2 and all(isinstance(x, str) for x in arc)):
                raise TypeError("BayesianNetwork(): incompatible constructor arguments: arcs must be (source, target) pairs of node names.")
            for x in arc:
                if x not in self._index:
                    raise IndexError(f"Node {x} not present in the graph.")
            if arc[1] in self._interface:
                raise ValueError("Interface node cannot have parents.")
            if arc[0] not in self._parents[arc[1]]:
                self._parents[arc[1]].append(arc[0])
                self._children[arc[0]].append(arc[1])
        if arcs:
            self.topological_sort()   # raises "Graph must be a DAG to obtain a topological sort."

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

    def _default_type(self):
        return self._type.default_node_type() if self._type.is_homogeneous() else UnknownFactorType()

    def _reindex(self):
        self._index = {n: i for i, n in enumerate(self._nodes + self._interface)}

    def add_node(self, node):
        """BNGeneric::add_node (BayesianNetwork.hpp:330-345): a new isolated node with the network's default type."""
        if node in self._index:
            raise ValueError(f"Cannot add node {node} because a node with the same name already exists.")
        self._nodes.append(node)
        self._parents[node], self._children[node] = [], []
        self._types[node] = self._default_type()
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
        self._types[node] = self._default_type()
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

    def _filled(self, net, arcs, types):
        """Copy node types and arcs onto a network just made by the type's new_bn / new_cbn."""
        if not self._type.is_homogeneous():
            for n, t in types:
                if net.contains_joint_node(n):
                    net._types[n] = t
        for s, t in arcs:
            net.add_arc(s, t)
        return net

    def conditional_bn(self, nodes=None, interface_nodes=None):
        """BNGeneric::conditional_bn (BayesianNetwork.hpp:1064-1100): the same arcs and node types over a new split of
        the variables into nodes and interface nodes (arcs into interface nodes are dropped); the network object comes
        from the type's new_cbn."""
        if nodes is None:
            nodes, interface_nodes = self._nodes, self._interface
        interface_nodes = list(interface_nodes or [])
        keep = set(nodes) | set(interface_nodes)
        arcs = [(s, t) for s, t in self.arcs() if s in keep and t in set(nodes)]
        return self._filled(self._type.new_cbn(list(nodes), interface_nodes), arcs, list(self._types.items()))

    def unconditional_bn(self):
        return self._filled(self._type.new_bn(self.joint_nodes()), self.arcs(), list(self._types.items()))

    def arcs(self):
        return [(p, n) for n in self._nodes for p in self._parents[n]]

    def graph(self):
        return Dag(self._nodes, self.arcs(), self._interface)

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
        """BayesianNetworkType::can_have_arc (SemiparametricBN.hpp:93-98, CLGNetwork.hpp:84-89, user types)."""
        return bool(self._type.can_have_arc(self, source, target))

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

    def _check_node(self, node):
        if node not in self._index:
            raise ValueError(f"Node {node} not present in the graph.")   # generic_graph.hpp:488-495

    def add_arc(self, source, target):
        """BNGeneric::add_arc (BayesianNetwork.hpp:549-555): allowed when the graph stays acyclic and the network type
        accepts the arc; adding an arc that is already there changes nothing."""
        self._check_node(source)
        self._check_node(target)
        if target in self._interface:
            raise ValueError("Interface node cannot have parents.")       # generic_graph.hpp:952-956
        if not self.can_add_arc(source, target):
            raise ValueError(f"Cannot add arc {source} -> {target}.")
        if not self.has_arc(source, target):
            self._parents[target].append(source)
            self._children[source].append(target)

    def remove_arc(self, source, target):
        self._check_node(source)
        self._check_node(target)
        if self.has_arc(source, target):
            self._parents[target].remove(source)
            self._children[source].remove(target)

    def flip_arc(self, source, target):
        """BNGeneric::flip_arc (BayesianNetwork.hpp:561-567)."""
        self._check_node(source)
        self._check_node(target)
        if not self.can_flip_arc(source, target):
            raise ValueError(f"Cannot flip arc {source} -> {target}.")
        if self.has_arc(source, target):
            self._parents[target].remove(source)
            self._children[source].remove(target)
            self._parents[source].append(target)
            self._children[target].append(source)

    # -- node types -----------------------------------------------------------------------------------
    def node_type(self, node):
        return self._types[node]

    def node_types(self):
        return dict(self._types)

    def has_unknown_node_types(self):
        if self._type.is_homogeneous():
            return False
        return any(self._types[n] == UnknownFactorType() for n in self._nodes)

    def underlying_node_type(self, df, node):
        """BNGeneric::underlying_node_type (BayesianNetwork.hpp:662-681): the node's type, or for an unknown one the
        first default of the network type for the column's data type (SemiparametricBN.hpp:43-55, CLGNetwork.hpp:40-50:
        float -> LinearGaussianCPD, dictionary -> DiscreteFactor)."""
        from .dataset import as_record_batch

        t = self._types[node]
        if t != UnknownFactorType():
            return t
        rb = as_record_batch(df)
        f = rb.schema.field(node)
        options = self._type.data_default_node_type(f.type)
        if not options:
            raise ValueError(f"There is no underlying FactorType for node {node} as there is no valid FactorType for DataType {f.type}")
        return options[0]

    def set_unknown_node_types(self, df, type_blacklist=()):
        """BNGeneric::set_unknown_node_types (BayesianNetwork.hpp:720-748): the first default of the column's data
        type that is not blacklisted."""
        from .dataset import as_record_batch

        if self._type.is_homogeneous():
            return
        rb = as_record_batch(df)
        black = {(n, t) for n, t in type_blacklist}
        new_types = []
        for node in self._nodes:
            if self._types[node] != UnknownFactorType():
                continue
            if rb.schema.get_field_index(node) < 0:
                raise ValueError(f"Column {node} not present in the DataFrame.")
            options = [o for o in self._type.data_default_node_type(rb.schema.field(node).type) if (node, o) not in black]
            if not options:
                raise ValueError(f"A valid FactorType for node {node} could not be inferred.")
            new_types.append((node, options[0]))
        self.force_type_whitelist(new_types)

    def force_type_whitelist(self, type_whitelist):
        """BNGeneric::force_type_whitelist (BayesianNetwork.hpp:761-790)."""
        for n, t in type_whitelist:
            self.set_node_type(n, t)

    def force_whitelist(self, arc_whitelist):
        """BNGeneric::force_whitelist: every whitelisted arc present, flipping the opposite arc when needed."""
        for s, t in arc_whitelist:
            if self.has_arc(s, t):
                continue
            if self.has_arc(t, s):
                raise ValueError(f"Arc {t} -> {s} in whitelist, but the opposite arc is present in the Bayesian network.")
            if not self.can_add_arc(s, t):
                raise ValueError(f"Arc {s} -> {t} in whitelist can not be added to the Bayesian network.")
            self.add_arc(s, t)

    def check_blacklist(self, arc_blacklist):
        for s, t in arc_blacklist:
            if self.has_arc(s, t):
                raise ValueError(f"Arc {s} -> {t} in blacklist, but it is present in the Bayesian Network.")

    def set_node_type(self, node, node_type):
        """BNGeneric::set_node_type (BayesianNetwork.hpp:701-718)."""
        wrong = f"Wrong factor type \"{node_type}\" for node \"{node}\" in Bayesian network type \"{self._type}\"."
        if self._type.is_homogeneous():
            if node_type != self._type.default_node_type():
                raise ValueError(wrong)
            return
        if node not in self._index:
            raise ValueError(f"Node {node} not present in the Bayesian network.")
        if node_type != UnknownFactorType() and not self._type.compatible_node_type(self, node, node_type):
            raise ValueError(wrong)
        if self._types.get(node) != node_type and getattr(self, "_cpds", None):
            self._cpds.pop(node, None)   # a factor of the old type no longer belongs to the node
        self._types[node] = node_type

    def can_have_cpd(self, node):
        return self.contains_node(node)

    def check_compatible_cpd(self, cpd):
        v = cpd.variable()
        if not self.contains_node(v):
            raise ValueError(f"CPD defined on variable which is not present in the model:\n{cpd}")
        if sorted(cpd.evidence()) != sorted(self._parents[v]):
            raise ValueError(f"CPD do not have the model's parent set as evidence:\n{cpd}")
        t = self._types[v]
        if t != UnknownFactorType() and cpd.type() != t:
            raise ValueError(f"Bayesian network expects type {t} for node {v}, but {cpd.type()} was provided.")

    def clone(self):
        """BayesianNetworkBase::clone (BayesianNetwork.hpp:105-119): same class (also a Python-derived one, with its
        extra attributes), structure, node types and factors; the containers are the clone's own."""
        import copy

        new = copy.copy(self)
        new._nodes, new._interface = list(self._nodes), list(self._interface)
        new._index, new._types = dict(self._index), dict(self._types)
        new._parents = {k: list(v) for k, v in self._parents.items()}
        new._children = {k: list(v) for k, v in self._children.items()}
        if getattr(self, "_cpds", None) is not None:
            new._cpds = dict(self._cpds)
        return new

    def _set_structure(self, arcs, node_types=()):
        """Replace arcs (and, for heterogeneous networks, node types) in place: how the learning algorithms write their
        result onto a clone of the start model."""
        for n in self._parents:
            self._parents[n], self._children[n] = [], []
        if not self._type.is_homogeneous():
            for n, t in node_types:
                if self._types.get(n) != t:
                    self._types[n] = t
        for s, t in arcs:
            self._parents[t].append(s)
            self._children[s].append(t)
        if getattr(self, "_cpds", None):
            for n in list(self._cpds):
                if not self._cpd_valid(n):
                    self._cpds.pop(n)
        return self

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
        nt = self._types[node]
        if nt == UnknownFactorType():
            nt = self.underlying_node_type(rb, node)
        if is_disc(node) and nt in (LinearGaussianCPDType(), CKDEType()):
            nt = DiscreteFactorType()
        if hasattr(nt, "new_factor") and type(nt) not in (LinearGaussianCPDType, CKDEType, DiscreteFactorType):
            return nt.new_factor(self, node, parents)   # FactorType::new_factor of a user-defined type
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
        from .dataset import as_record_batch

        df = as_record_batch(df)   # factors - also Python-derived ones - see a pyarrow.RecordBatch, as in the reference
        if getattr(self, "_cpds", None) is None:
            self._cpds = {}
        if self.has_unknown_node_types():
            self.set_unknown_node_types(df)
        from .dataset import default_context, shared_upload

        with shared_upload(default_context(), df, self._upload_columns()):   # one PCIe pass over the model's columns for all factors
            for n in self._nodes:
                if not self._cpd_valid(n):
                    self._cpds[n] = self._new_factor(df, n)
                if not self._cpds[n].fitted():
                    self._cpds[n].fit(df)

    def _upload_columns(self):
        """The columns the factors of this model read: its nodes and, for a conditional network, its interface nodes."""
        return list(self._nodes) + list(getattr(self, "_interface", ()))

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

        from .dataset import as_record_batch, default_context, shared_upload

        df = as_record_batch(df)
        out = None
        with shared_upload(default_context(), df, self._upload_columns()):
            for n in self._nodes:
                ll = np.asarray(self._cpds[n].logl(df), dtype=np.float64)
                out = ll if out is None else out + ll
        return out if out is not None else np.zeros(0)

    def slogl(self, df):
        if not self.fitted():
            raise ValueError("Model not fitted.")
        from .dataset import as_record_batch, default_context, shared_upload

        df = as_record_batch(df)
        with shared_upload(default_context(), df, self._upload_columns()):
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
            self.check_compatible_cpd(f)
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
        """BNGeneric::__getstate__ (BayesianNetwork.hpp:1217-1260): structure, type, node types, and - with include_cpd -
        the factors added so far; a Python-derived class adds `__getstate_extra__()`."""
        state = {"type": self._type, "nodes": self._nodes, "arcs": self.arcs(), "types": list(self._types.items()),
                 "interface": self._interface, "include_cpd": bool(getattr(self, "_include_cpd", False))}
        if state["include_cpd"] and getattr(self, "_cpds", None):
            state["cpds"] = [self._cpds[n] for n in self._nodes if n in self._cpds]
        extra = getattr(self, "__getstate_extra__", None)
        if extra is not None:
            state["extra"] = extra()
        return state

    def __setstate__(self, state):
        BayesianNetwork.__init__(self, state["type"], state["nodes"], (), (), state.get("interface", ()))
        if not self._type.is_homogeneous():
            self._types.update(dict(state["types"]))
        for s, t in state["arcs"]:
            self._parents[t].append(s)
            self._children[s].append(t)
        self._include_cpd = bool(state.get("include_cpd", False))
        if "cpds" in state:
            self._cpds = {f.variable(): f for f in state["cpds"]}
            self._include_cpd = True
        if "extra" in state:
            self.__setstate_extra__(state["extra"])

    @property
    def include_cpd(self):
        return getattr(self, "_include_cpd", False)

    @include_cpd.setter
    def include_cpd(self, value):
        self._include_cpd = bool(value)

    def __str__(self):
        return f"{self._type} with {self.num_nodes()} nodes and {self.num_arcs()} arcs"

    __repr__ = __str__



class ConditionalBayesianNetwork(BayesianNetwork, ConditionalBayesianNetworkBase):
    """models/BayesianNetwork.hpp ConditionalBayesianNetwork: (type, nodes, interface_nodes[, arcs][, node_types])."""

    def __init__(self, bn_type, nodes, interface_nodes, arcs=(), node_types=()):
        BayesianNetwork.__init__(self, bn_type, list(nodes), arcs, node_types, list(interface_nodes))


def _typed(name, bn_type_cls, conditional, takes_types, doc):
    base = ConditionalBayesianNetwork if conditional else BayesianNetwork
    if conditional:
        def __init__(self, nodes, interface_nodes, arcs=(), node_types=()):
            if not takes_types and node_types:
                raise TypeError(f"{name} does not take node types")
            ConditionalBayesianNetwork.__init__(self, bn_type_cls(), nodes, interface_nodes, arcs, node_types)
    else:
        def __init__(self, nodes, arcs=(), node_types=()):
            if not takes_types and (node_types or _is_type_list(list(arcs))):
                raise TypeError(f"{name} does not take node types")
            BayesianNetwork.__init__(self, bn_type_cls(), nodes, arcs, node_types)
    return type(name, (base,), {"__init__": __init__, "__doc__": doc, "__module__": __name__})


GaussianNetwork = _typed("GaussianNetwork", GaussianNetworkType, False, False, "models/GaussianNetwork.hpp: every node a LinearGaussianCPD.")
KDENetwork = _typed("KDENetwork", KDENetworkType, False, False, "models/KDENetwork.hpp: every node a CKDE.")
DiscreteBN = _typed("DiscreteBN", DiscreteBNType, False, False, "models/DiscreteBN.hpp: every node a DiscreteFactor.")
SemiparametricBN = _typed("SemiparametricBN", SemiparametricBNType, False, True, "models/SemiparametricBN.hpp: LinearGaussianCPD / CKDE (/ DiscreteFactor) per node.")
CLGNetwork = _typed("CLGNetwork", CLGNetworkType, False, True, "models/CLGNetwork.hpp: DiscreteFactor and (conditional) LinearGaussianCPD nodes.")
ConditionalGaussianNetwork = _typed("ConditionalGaussianNetwork", GaussianNetworkType, True, False, "Conditional GaussianNetwork.")
ConditionalKDENetwork = _typed("ConditionalKDENetwork", KDENetworkType, True, False, "Conditional KDENetwork.")
ConditionalDiscreteBN = _typed("ConditionalDiscreteBN", DiscreteBNType, True, False, "Conditional DiscreteBN.")
ConditionalSemiparametricBN = _typed("ConditionalSemiparametricBN", SemiparametricBNType, True, True, "Conditional SemiparametricBN.")
ConditionalCLGNetwork = _typed("ConditionalCLGNetwork", CLGNetworkType, True, True, "Conditional CLGNetwork.")


class HomogeneousBN(BayesianNetwork):
    """models/HomogeneousBN.hpp: HomogeneousBN(factor_type, nodes | arcs[, arcs])."""

    def __init__(self, factor_type, nodes, arcs=()):
        BayesianNetwork.__init__(self, HomogeneousBNType(factor_type), nodes, arcs)


class ConditionalHomogeneousBN(ConditionalBayesianNetwork):
    def __init__(self, factor_type, nodes, interface_nodes, arcs=()):
        ConditionalBayesianNetwork.__init__(self, HomogeneousBNType(factor_type), nodes, interface_nodes, arcs)


class HeterogeneousBN(BayesianNetwork):
    """models/HeterogeneousBN.hpp: HeterogeneousBN(factor_types (list | {DataType: list}), nodes | arcs[, arcs][, node_types])."""

    def __init__(self, factor_types, nodes, arcs=(), node_types=()):
        BayesianNetwork.__init__(self, HeterogeneousBNType(factor_types), nodes, arcs, node_types)


class ConditionalHeterogeneousBN(ConditionalBayesianNetwork):
    def __init__(self, factor_types, nodes, interface_nodes, arcs=(), node_types=()):
        ConditionalBayesianNetwork.__init__(self, HeterogeneousBNType(factor_types), nodes, interface_nodes, arcs, node_types)


def load(name):
    """pybnesian.load (pybindings/lib.cpp, util/pickle.hpp): un-pickle a saved model / factor."""
    import pickle

    with open(name, "rb") as f:
        return pickle.load(f)
