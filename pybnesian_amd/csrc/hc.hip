// Greedy hill-climbing with the delta-score cache — host logic of libpbn_hip.
//
// Mirrors, decision for decision, /root/reference/pybnesian/learning/operators/operators.{hpp,cpp}
// (ArcOperatorSet :19-132,296-363 / operators.hpp:489-525,580-623; ChangeNodeTypeSet operators.cpp:439-599;
// OperatorPool operators.hpp:836-904; Operators :89-244; OperatorTabuSet :258-293; LocalScoreCache :295-338),
// learning/algorithms/hillclimbing.hpp:46-199 (estimate_hc) and the DAG legality predicates of
// graph/generic_graph.hpp:2711-2745.  What changes is WHEN scores are computed: every place where the
// reference calls Score::local_score inside a loop (cache_scores' n(n-1) cells, update_scores' <= 2(n-1)
// cells per changed node, the local caches) first collects all (variable, node type, parent list) requests
// of that step and submits them as ONE batch to the score callback - the unit that is spread over the CUs of
// a GPU and over the GPUs of a node.  Cell arithmetic, candidate parent orderings (swap_remove / push_back
// exactly as the reference manipulates them), the std::sort-based find_max with its persistent index
// vector, the stop rule, patience and tabu handling are the reference's.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <limits>
#include <memory>
#include <set>
#include <string>
#include <vector>

#include "common.hpp"

using namespace pbn;

namespace {

constexpr double MACHINE_TOL = 1.4901161193847656e-08;  // util/math_constants.hpp:30
const double LOWEST = std::numeric_limits<double>::lowest();

enum OpKind { OP_ADD = 0, OP_REMOVE = 1, OP_FLIP = 2, OP_TYPE = 3 };

struct Op {
    int kind = -1;
    int source = -1, target = -1;  // arcs; OP_TYPE: source = node
    int new_type = -1;             // OP_TYPE
    double delta = 0;
    bool valid() const { return kind >= 0; }
    bool same(const Op& o) const {  // hash/eq of operators.hpp:108-119: type + source + target (+ node type)
        return kind == o.kind && source == o.source && target == o.target && new_type == o.new_type;
    }
};

struct Model {
    int n = 0;   // nodes
    int ni = 0;  // interface nodes of a conditional network: ids n .. n+ni-1, sources only (ConditionalBayesianNetwork)
    int bn_type = PBN_BN_GAUSSIAN;
    std::vector<std::vector<int>> parents, children;
    std::vector<char> adj;  // adj[s + t*J] = arc s -> t, J = n + ni
    std::vector<int> node_type;
    int J() const { return n + ni; }
    bool is_interface(int v) const { return v >= n; }
    void reset_graph() {
        parents.assign(J(), {}); children.assign(J(), {}); adj.assign((size_t)J() * J(), 0);
    }
    bool has_arc(int s, int t) const { return adj[s + (size_t)t * J()] != 0; }
    void add_arc(int s, int t) {
        if (has_arc(s, t)) return;
        adj[s + (size_t)t * J()] = 1;
        parents[t].push_back(s);
        children[s].push_back(t);
    }
    static void swap_remove(std::vector<int>& v, int x) {  // util::swap_remove_v
        auto it = std::find(v.begin(), v.end(), x);
        if (it != v.end()) { *it = v.back(); v.pop_back(); }
    }
    void remove_arc(int s, int t) {
        if (!has_arc(s, t)) return;
        adj[s + (size_t)t * J()] = 0;
        swap_remove(parents[t], s);
        swap_remove(children[s], t);
    }
    bool has_path(int from, int to, int skip_s = -1, int skip_t = -1) const {  // follows children
        std::vector<char> seen(J(), 0);
        std::vector<int> stack{from};
        seen[from] = 1;
        while (!stack.empty()) {
            int u = stack.back();
            stack.pop_back();
            for (int c : children[u]) {
                if (u == skip_s && c == skip_t) continue;
                if (c == to) return true;
                if (!seen[c]) { seen[c] = 1; stack.push_back(c); }
            }
        }
        return false;
    }
    // BayesianNetworkType::can_have_arc (SemiparametricBN.hpp:93-98, CLGNetwork.hpp:84-89): no continuous -> discrete arcs
    bool can_have_arc(int s, int t) const { return !(node_type[t] == PBN_NODE_DISCRETE && node_type[s] != PBN_NODE_DISCRETE); }
    bool can_add_arc(int s, int t) const {  // generic_graph.hpp:2711-2718 && BayesianNetwork.hpp:571-577
        return s != t && !is_interface(t) && can_have_arc(s, t) && (parents[s].empty() || children[t].empty() || !has_path(t, s));
    }
    bool can_flip_arc(int s, int t) const {  // generic_graph.hpp:2721-2745; the flipped arc is t -> s
        if (s == t || is_interface(s) || is_interface(t) || !can_have_arc(t, s)) return false;
        if (has_arc(s, t)) {
            if (parents[t].size() == 1 || children[s].size() == 1) return true;
            return !has_path(s, t, s, t);
        }
        if (parents[t].empty() || children[s].empty()) return true;
        return !has_path(s, t);
    }
    int alternative_type(int node) const {  // SemiparametricBN.hpp:106-119
        if (bn_type != PBN_BN_SEMIPARAMETRIC || node_type[node] == PBN_NODE_DISCRETE) return -1;
        return node_type[node] == PBN_NODE_LG ? PBN_NODE_CKDE : PBN_NODE_LG;
    }
    void apply(const Op& op) {
        switch (op.kind) {
            case OP_ADD: add_arc(op.source, op.target); break;
            case OP_REMOVE: remove_arc(op.source, op.target); break;
            case OP_FLIP: remove_arc(op.source, op.target); add_arc(op.target, op.source); break;
            case OP_TYPE: node_type[op.source] = op.new_type; break;
        }
    }
    Op opposite(const Op& op) const {  // operators.hpp:89-244, evaluated AFTER apply (as estimate_hc does)
        Op o = op;
        switch (op.kind) {
            case OP_ADD: o.kind = OP_REMOVE; o.delta = -op.delta; break;
            case OP_REMOVE: o.kind = OP_ADD; o.delta = -op.delta; break;
            case OP_FLIP: o.source = op.target; o.target = op.source; o.delta = -op.delta; break;
            // ChangeNodeType::opposite(m) = ChangeNodeType(node, m.node_type(node), -delta) (operators.hpp:214-216); estimate_hc
            // calls it AFTER apply, so the tabu entry names the NEW type (a reference quirk that is reproduced, not fixed)
            case OP_TYPE: o.new_type = node_type[op.source]; o.delta = -op.delta; break;
        }
        return o;
    }
    std::vector<int> nodes_changed(const Op& op) const {
        if (op.kind == OP_FLIP) return {op.source, op.target};
        if (op.kind == OP_TYPE) return {op.source};
        return {op.target};
    }
};

// One batch of Score::local_score requests.
struct Batch {
    std::vector<int> var, ntype, off{0}, par;
    int add(int v, int t, const std::vector<int>& p) {
        var.push_back(v);
        ntype.push_back(t);
        par.insert(par.end(), p.begin(), p.end());
        off.push_back((int)par.size());
        return (int)var.size() - 1;
    }
    int size() const { return (int)var.size(); }
};

struct Scorer {
    pbn_hc_score_fn fn;
    void* user;
    int64_t evals = 0;
    std::vector<double> run(const Batch& b, int validated) {
        std::vector<double> out(b.size());
        if (b.size() == 0) return out;
        evals += b.size();
        int rc = fn(user, validated, b.size(), b.var.data(), b.ntype.data(), b.off.data(), b.par.data(), out.data());
        if (rc != 0) throw invalid_error(std::string("score callback failed: ") + pbn_last_error());
        return out;
    }
};

struct ArcSet {
    int n = 0, J = 0;
    std::vector<double> delta;   // col-major: delta[s + t*J], s over the joint nodes, t over the nodes
    std::vector<char> valid_op;
    mutable std::vector<int> sorted_idx;
    int max_indegree = 0;
    std::vector<std::pair<int, int>> blacklist, whitelist;

    void update_valid_ops(const Model& m) {  // operators.cpp:19-69 (cells the reference leaves uninitialised start at lowest())
        // conditional networks (operators.cpp:153-210): rows run over the joint nodes; an interface source has no
        // reverse cell
        n = m.n; J = m.J();
        delta.assign((size_t)J * n, LOWEST);
        valid_op.assign((size_t)J * n, 1);
        for (auto& a : whitelist) {
            valid_op[a.first + (size_t)a.second * J] = 0;
            if (!m.is_interface(a.first)) valid_op[a.second + (size_t)a.first * J] = 0;
        }
        for (auto& a : blacklist) valid_op[a.first + (size_t)a.second * J] = 0;
        for (int i = 0; i < n; ++i) valid_op[i + (size_t)i * J] = 0;
        sorted_idx.clear();
        for (int i = 0; i < J; ++i)
            for (int j = 0; j < n; ++j)
                if (valid_op[i + (size_t)j * J]) sorted_idx.push_back(i + j * J);
    }
};

struct TypeSet {
    std::vector<double> delta;  // one alternative type per node (SPBN); LOWEST when unavailable
    std::vector<char> has;      // delta[i] has an entry (alt type exists)
    std::vector<char> whitelisted;
    std::set<std::pair<int, int>> type_blacklist;
};

struct Engine {
    Model cur;
    Scorer scorer;
    bool use_arcs = false, use_types = false;
    int order_arcs_first = 1;
    ArcSet arcs;
    TypeSet types;
    std::vector<double> local;  // LocalScoreCache of the pool / op set
    int64_t cells_scored = 0;

    // ---- ArcOperatorSet::cache_scores (operators.cpp:100-132 + cache_score_operation :71-98) ---------
    struct CCell { int s, t; int a, b; int kind; };  // kind 0: d = S[a]-local[t]; 1: S[a]+S[b]-local[s]-local[t]
    void arcs_collect_cache(Batch& bt, std::vector<CCell>& cells) {
        arcs.update_valid_ops(cur);
        const int n = cur.n, J = cur.J();
        for (int t = 0; t < n; ++t) {
            std::vector<int> pt = cur.parents[t];
            for (int s = 0; s < J; ++s) {
                if (!arcs.valid_op[s + (size_t)t * J] || !cur.can_have_arc(s, t)) continue;
                if (cur.has_arc(s, t)) {   // also cache_score_interface (operators.cpp:134-151) for an interface source
                    Model::swap_remove(pt, s);
                    int a = bt.add(t, cur.node_type[t], pt);
                    pt.push_back(s);
                    cells.push_back({s, t, a, -1, 0});
                } else if (cur.has_arc(t, s)) {
                    std::vector<int> ps = cur.parents[s];
                    Model::swap_remove(ps, t);
                    pt.push_back(s);
                    int a = bt.add(s, cur.node_type[s], ps);
                    int b = bt.add(t, cur.node_type[t], pt);
                    pt.pop_back();
                    cells.push_back({s, t, a, b, 1});
                } else {
                    pt.push_back(s);
                    int a = bt.add(t, cur.node_type[t], pt);
                    pt.pop_back();
                    cells.push_back({s, t, a, -1, 0});
                }
            }
        }
    }
    void arcs_apply_cache(const std::vector<CCell>& cells, const std::vector<double>& S) {
        const int n = cur.J();
        for (const CCell& c : cells) {
            double d;
            if (c.kind == 0) d = S[c.a] - local[c.t];
            else d = S[c.a] + S[c.b] - local[c.s] - local[c.t];
            arcs.delta[c.s + (size_t)c.t * n] = d;
        }
        cells_scored += (int64_t)cells.size();
    }

    // ---- ArcOperatorSet::update_incoming_arcs_scores (operators.cpp:296-347), requests only -------------
    struct UCell { int s, t; int a, b; int kind; };  // kind 0 remove/add: S[a]-local[t]; 1 flip-of-existing: cell(t,s) = d_remove + S[b]-local[s]; 2 flip: S[a]+S[b]-local[s]-local[t]
    void arcs_collect_updates(int t, Batch& bt, std::vector<UCell>& cells) {
        const int n = cur.J();   // stride of the delta matrix; interface sources (operators.cpp:365-418) never flip
        std::vector<int> parents = cur.parents[t];
        for (int s = 0; s < n; ++s) {
            if (!arcs.valid_op[s + (size_t)t * n]) continue;
            if (cur.has_arc(s, t)) {
                Model::swap_remove(parents, s);
                int a = bt.add(t, cur.node_type[t], parents);
                parents.push_back(s);
                int b = -1;
                if (!cur.is_interface(s) && cur.can_have_arc(t, s) && arcs.valid_op[t + (size_t)s * n]) {
                    std::vector<int> ps = cur.parents[s];
                    ps.push_back(t);
                    b = bt.add(s, cur.node_type[s], ps);
                }
                cells.push_back({s, t, a, b, 1});
            } else if (!cur.is_interface(s) && cur.has_arc(t, s) && cur.can_have_arc(s, t)) {
                std::vector<int> ps = cur.parents[s];
                Model::swap_remove(ps, t);
                parents.push_back(s);
                int a = bt.add(s, cur.node_type[s], ps);
                int b = bt.add(t, cur.node_type[t], parents);
                parents.pop_back();
                cells.push_back({s, t, a, b, 2});
            } else if (cur.can_have_arc(s, t)) {
                parents.push_back(s);
                int a = bt.add(t, cur.node_type[t], parents);
                parents.pop_back();
                cells.push_back({s, t, a, -1, 0});
            }
        }
    }
    void arcs_apply_updates(const std::vector<UCell>& cells, const std::vector<double>& S) {
        const int n = cur.J();
        for (const UCell& c : cells) {
            if (c.kind == 0) {
                arcs.delta[c.s + (size_t)c.t * n] = S[c.a] - local[c.t];
                ++cells_scored;
            } else if (c.kind == 1) {
                const double d = S[c.a] - local[c.t];
                arcs.delta[c.s + (size_t)c.t * n] = d;
                ++cells_scored;
                if (c.b >= 0) {
                    arcs.delta[c.t + (size_t)c.s * n] = d + S[c.b] - local[c.s];
                    ++cells_scored;
                }
            } else {
                arcs.delta[c.s + (size_t)c.t * n] = S[c.a] + S[c.b] - local[c.s] - local[c.t];
                ++cells_scored;
            }
        }
    }

    // ---- ArcOperatorSet::find_max_indegree (operators.hpp:489-525, tabu :580-623) --------------------------
    Op arcs_find_max(const std::vector<Op>* tabu) const {
        const int n = cur.J();
        const double* dp = arcs.delta.data();
        std::sort(arcs.sorted_idx.begin(), arcs.sorted_idx.end(), [dp](int i1, int i2) { return dp[i1] > dp[i2]; });
        auto in_tabu = [&](const Op& o) {
            if (!tabu) return false;
            for (const Op& x : *tabu)
                if (x.same(o)) return true;
            return false;
        };
        const bool limited = arcs.max_indegree > 0;
        for (int idx : arcs.sorted_idx) {
            const int s = idx % n, t = idx / n;
            Op o;
            if (cur.has_arc(s, t)) {
                o.kind = OP_REMOVE; o.source = s; o.target = t; o.delta = dp[idx];
                if (!in_tabu(o)) return o;
            } else if (cur.is_interface(s)) {
                // operators.hpp:547-556: one direction only, no cycle possible
                if (limited && (int)cur.parents[t].size() >= arcs.max_indegree) continue;
                if (cur.can_have_arc(s, t)) {
                    o.kind = OP_ADD; o.source = s; o.target = t; o.delta = dp[idx];
                    if (!in_tabu(o)) return o;
                }
            } else if (cur.has_arc(t, s) && cur.can_flip_arc(t, s)) {
                if (limited && (int)cur.parents[t].size() >= arcs.max_indegree) continue;
                o.kind = OP_FLIP; o.source = t; o.target = s; o.delta = dp[idx];
                if (!in_tabu(o)) return o;
            } else if (cur.can_add_arc(s, t)) {
                if (limited && (int)cur.parents[t].size() >= arcs.max_indegree) continue;
                o.kind = OP_ADD; o.source = s; o.target = t; o.delta = dp[idx];
                if (!in_tabu(o)) return o;
            }
        }
        return Op{};
    }

    // ---- ChangeNodeTypeSet (operators.cpp:439-599) ---------------------------------------------------------
    void types_collect(const std::vector<int>& nodes, Batch& bt, std::vector<std::pair<int, int>>& req) {
        for (int v : nodes) {
            if (types.whitelisted[v]) continue;
            const int alt = cur.alternative_type(v);
            if (alt < 0) { types.has[v] = 0; continue; }
            types.has[v] = 1;
            if (types.type_blacklist.count({v, alt})) {
                types.delta[v] = LOWEST;
                continue;
            }
            req.push_back({v, bt.add(v, alt, cur.parents[v])});
        }
    }
    void types_apply(const std::vector<std::pair<int, int>>& req, const std::vector<double>& S) {
        for (auto& r : req) {
            types.delta[r.first] = S[r.second] - local[r.first];
            ++cells_scored;
        }
    }
    Op types_find_max(const std::vector<Op>* tabu) const {
        double max_score = LOWEST;
        int max_node = -1;
        for (int i = 0; i < cur.n; ++i) {
            if (types.whitelisted[i] || !types.has[i]) continue;
            if (types.delta[i] > max_score) {
                if (tabu) {
                    Op o; o.kind = OP_TYPE; o.source = i; o.new_type = cur.alternative_type(i);
                    bool hit = false;
                    for (const Op& x : *tabu) if (x.same(o)) { hit = true; break; }
                    if (hit) continue;
                }
                max_score = types.delta[i];
                max_node = i;
            }
        }
        if (max_score > LOWEST) {
            Op o; o.kind = OP_TYPE; o.source = max_node; o.new_type = cur.alternative_type(max_node); o.delta = max_score;
            return o;
        }
        return Op{};
    }

    // ---- pool-level operations (OperatorPool, operators.hpp:836-904) ----------------------------------------
    // All local_score requests of a step travel in ONE batch (local cache + every delta cell): the values
    // are the ones the reference computes call by call, only their evaluation is batched.
    void cache_scores() {
        Batch bt;
        for (int v = 0; v < cur.n; ++v) bt.add(v, cur.node_type[v], cur.parents[v]);
        std::vector<CCell> cells;
        std::vector<std::pair<int, int>> treq;
        if (use_arcs) arcs_collect_cache(bt, cells);
        if (use_types) {
            types.delta.assign(cur.n, LOWEST);
            types.has.assign(cur.n, 0);
            std::vector<int> all(cur.n);
            for (int i = 0; i < cur.n; ++i) all[i] = i;
            types_collect(all, bt, treq);
        }
        std::vector<double> S = scorer.run(bt, 0);
        local.assign(S.begin(), S.begin() + cur.n);
        if (use_arcs) arcs_apply_cache(cells, S);
        if (use_types) types_apply(treq, S);
    }

    // ---- near ties (round 6, pbn_hc_config.near_tie_abs; not in the reference) -----------------------------------------------------
    bool bears_ckde(const Op& o) const {   // does the operator's delta rest on a CKDE local score?
        if (o.kind == OP_TYPE) return o.new_type == PBN_NODE_CKDE || cur.node_type[o.source] == PBN_NODE_CKDE;
        if (cur.node_type[o.target] == PBN_NODE_CKDE) return true;
        return o.kind == OP_FLIP && cur.node_type[o.source] == PBN_NODE_CKDE;
    }
    // the runner-up of find_max: the best valid operator other than `best`; the persistent sort order is put back (the second std::sort must
    // not change which of two EQUAL deltas a later find_max meets first)
    Op runner_up(const std::vector<Op>* tabu, const Op& best) const {
        std::vector<Op> tb;
        if (tabu) tb = *tabu;
        tb.push_back(best);
        const std::vector<int> keep = arcs.sorted_idx;
        Op second = find_max(&tb);
        arcs.sorted_idx = keep;
        return second;
    }
    // the operator's delta from local scores asked for at full precision (callback flag 2): new local scores of the nodes it changes minus
    // their current ones, all four / two evaluated in one batch
    double precise_delta(const Op& o) {
        Batch b;
        auto with = [&](int t, int s, bool add) {
            std::vector<int> p = cur.parents[t];
            if (add) p.push_back(s); else p.erase(std::remove(p.begin(), p.end(), s), p.end());
            return p;
        };
        std::vector<int> plus, minus;   // indices into the batch
        if (o.kind == OP_TYPE) {
            plus.push_back(b.add(o.source, o.new_type, cur.parents[o.source]));
            minus.push_back(b.add(o.source, cur.node_type[o.source], cur.parents[o.source]));
        } else if (o.kind == OP_ADD) {
            plus.push_back(b.add(o.target, cur.node_type[o.target], with(o.target, o.source, true)));
            minus.push_back(b.add(o.target, cur.node_type[o.target], cur.parents[o.target]));
        } else if (o.kind == OP_REMOVE) {
            plus.push_back(b.add(o.target, cur.node_type[o.target], with(o.target, o.source, false)));
            minus.push_back(b.add(o.target, cur.node_type[o.target], cur.parents[o.target]));
        } else {   // flip source -> target into target -> source
            plus.push_back(b.add(o.target, cur.node_type[o.target], with(o.target, o.source, false)));
            plus.push_back(b.add(o.source, cur.node_type[o.source], with(o.source, o.target, true)));
            minus.push_back(b.add(o.target, cur.node_type[o.target], cur.parents[o.target]));
            minus.push_back(b.add(o.source, cur.node_type[o.source], cur.parents[o.source]));
        }
        const std::vector<double> S = scorer.run(b, 2);
        double d = 0.0;
        for (int i : plus) d += S[(size_t)i];
        for (int i : minus) d -= S[(size_t)i];
        return d;
    }

    Op find_max(const std::vector<Op>* tabu) const {
        if (tabu && tabu->empty()) tabu = nullptr;
        double max_delta = LOWEST;
        Op best;
        auto consider = [&](const Op& o) {
            if (o.valid() && o.delta > max_delta) { best = o; max_delta = o.delta; }
        };
        if (order_arcs_first) {
            if (use_arcs) consider(arcs_find_max(tabu));
            if (use_types) consider(types_find_max(tabu));
        } else {
            if (use_types) consider(types_find_max(tabu));
            if (use_arcs) consider(arcs_find_max(tabu));
        }
        return best;
    }

    void update_scores(const std::vector<int>& changed) {
        Batch bt;
        for (int v : changed) bt.add(v, cur.node_type[v], cur.parents[v]);  // refreshed local scores first
        std::vector<UCell> cells;
        std::vector<std::pair<int, int>> treq;
        if (use_arcs)
            for (int t : changed) arcs_collect_updates(t, bt, cells);
        if (use_types) types_collect(changed, bt, treq);
        std::vector<double> S = scorer.run(bt, 0);
        for (size_t i = 0; i < changed.size(); ++i) local[changed[i]] = S[i];
        if (use_arcs) arcs_apply_updates(cells, S);
        if (use_types) types_apply(treq, S);
    }
};

}  // namespace

static void init_engine(Engine& e, const pbn_hc_config* cfg, pbn_hc_score_fn fn, void* user) {
    const int n = cfg->n_nodes;
    if (n <= 0) throw invalid_error("pbn_hc_estimate: empty model");
    e.scorer = Scorer{fn, user};
    Model& m = e.cur;
    if (cfg->n_interface < 0) throw invalid_error("pbn_hc_estimate: negative number of interface nodes");
    m.n = n; m.ni = cfg->n_interface; m.bn_type = cfg->bn_type;
    m.reset_graph();
    m.node_type.assign(m.J(), cfg->bn_type == PBN_BN_KDE ? PBN_NODE_CKDE : PBN_NODE_LG);
    if (cfg->bn_type < PBN_BN_GAUSSIAN || cfg->bn_type > PBN_BN_CLG) throw invalid_error("pbn_hc_estimate: unknown network type");
    if (cfg->node_types)
        for (int i = 0; i < m.J(); ++i) m.node_type[i] = cfg->node_types[i];
    auto check_node = [&](int v) { if (v < 0 || v >= n) throw invalid_error("pbn_hc_estimate: node index out of range"); };
    auto check_source = [&](int v) { if (v < 0 || v >= m.J()) throw invalid_error("pbn_hc_estimate: node index out of range"); };
    // force type whitelist (hillclimbing.hpp:77)
    for (int i = 0; i < cfg->n_type_whitelist; ++i) {
        check_node(cfg->type_whitelist[2 * i]);
        m.node_type[cfg->type_whitelist[2 * i]] = cfg->type_whitelist[2 * i + 1];
    }
    for (int i = 0; i < cfg->n_arcs; ++i) {
        check_source(cfg->arcs[2 * i]); check_node(cfg->arcs[2 * i + 1]);
        m.add_arc(cfg->arcs[2 * i], cfg->arcs[2 * i + 1]);
    }
    // check_blacklist / force_whitelist (hillclimbing.hpp:95-96)
    for (int i = 0; i < cfg->n_arc_blacklist; ++i) {
        const int s = cfg->arc_blacklist[2 * i], t = cfg->arc_blacklist[2 * i + 1];
        check_source(s); check_node(t);
        if (m.has_arc(s, t)) throw invalid_error("Arc in the blacklist is present in the starting Bayesian network.");
        e.arcs.blacklist.push_back({s, t});
    }
    for (int i = 0; i < cfg->n_arc_whitelist; ++i) {
        const int s = cfg->arc_whitelist[2 * i], t = cfg->arc_whitelist[2 * i + 1];
        check_source(s); check_node(t);
        if (!m.has_arc(s, t)) {
            if (!m.is_interface(s) && m.has_arc(t, s)) m.remove_arc(t, s);
            if (!m.can_add_arc(s, t)) throw invalid_error("Arc whitelist creates a cycle in the starting Bayesian network.");
            m.add_arc(s, t);
        }
        e.arcs.whitelist.push_back({s, t});
    }
    e.use_arcs = cfg->op_arcs != 0;
    e.use_types = cfg->op_node_type != 0;
    e.order_arcs_first = cfg->arcs_first;
    if (!e.use_arcs && !e.use_types) throw invalid_error("pbn_hc_estimate: no operator set");
    if (e.use_types && cfg->bn_type != PBN_BN_SEMIPARAMETRIC)
        throw invalid_error("ChangeNodeTypeSet can only be used with non-homogeneous Bayesian networks.");
    e.arcs.max_indegree = cfg->max_indegree;
    e.types.whitelisted.assign(n, 0);
    for (int i = 0; i < cfg->n_type_whitelist; ++i) e.types.whitelisted[cfg->type_whitelist[2 * i]] = 1;
    for (int i = 0; i < cfg->n_type_blacklist; ++i)
        e.types.type_blacklist.insert({cfg->type_blacklist[2 * i], cfg->type_blacklist[2 * i + 1]});
}

struct pbn_hc {
    std::recursive_mutex mu;   // the handle's own state; its score callback takes the lock of whatever context it touches
    Engine e;
    bool cached = false;
};
static std::recursive_mutex& hc_mu(pbn_hc* h) { return h ? h->mu : pbn::api_mutex(); }

extern "C" {

// ---- stateful operator-set interface: OperatorSet::{cache_scores,find_max,find_max_tabu,update_scores} and
// LocalScoreCache of learning/operators/operators.hpp:295-355 for callers that drive their own search loop ----
int pbn_hc_create(const pbn_hc_config* cfg, pbn_hc_score_fn fn, void* user, pbn_hc** out) {
    return guarded([&] {
        if (!cfg || !fn || !out) throw invalid_error("pbn_hc_create: null argument");
        auto h = std::make_unique<pbn_hc>();
        init_engine(h->e, cfg, fn, user);
        *out = h.release();
    });
}

void pbn_hc_destroy(pbn_hc* h) { delete h; }   // (the caller owns the handle: nobody else may be inside it)

int pbn_hc_set_model(pbn_hc* h, int n_arcs, const int* arcs, const int* node_types) {
    return guarded(hc_mu(h), [&] {
        if (!h || (n_arcs > 0 && !arcs)) throw invalid_error("pbn_hc_set_model: null argument");
        Model& m = h->e.cur;
        const int n = m.n;
        m.reset_graph();
        if (node_types)
            for (int i = 0; i < m.J(); ++i) m.node_type[i] = node_types[i];
        for (int i = 0; i < n_arcs; ++i) {
            const int s = arcs[2 * i], t = arcs[2 * i + 1];
            if (s < 0 || s >= m.J() || t < 0 || t >= n) throw invalid_error("pbn_hc_set_model: node index out of range");
            m.add_arc(s, t);
        }
    });
}

int pbn_hc_cache_scores(pbn_hc* h) {
    return guarded(hc_mu(h), [&] {
        if (!h) throw invalid_error("pbn_hc_cache_scores: null handle");
        h->e.cache_scores();
        h->cached = true;
    });
}

int pbn_hc_find_max(pbn_hc* h, int n_tabu, const int* tabu, int* op, double* delta) {
    return guarded(hc_mu(h), [&] {
        if (!h || !op || !delta) throw invalid_error("pbn_hc_find_max: null argument");
        if (!h->cached) throw invalid_error("Local cache not initialized. Call cache_scores() before find_max()");
        std::vector<Op> tb((size_t)n_tabu);
        for (int i = 0; i < n_tabu; ++i) {
            tb[i].kind = tabu[4 * i];
            tb[i].source = tabu[4 * i + 1];
            if (tb[i].kind == OP_TYPE) tb[i].new_type = tabu[4 * i + 2]; else tb[i].target = tabu[4 * i + 2];
        }
        Op o = h->e.find_max(n_tabu > 0 ? &tb : nullptr);
        op[0] = o.kind;
        op[1] = o.source;
        op[2] = o.kind == OP_TYPE ? o.new_type : o.target;
        *delta = o.delta;
    });
}

int pbn_hc_update_scores(pbn_hc* h, int n, const int* nodes) {
    return guarded(hc_mu(h), [&] {
        if (!h || (n > 0 && !nodes)) throw invalid_error("pbn_hc_update_scores: null argument");
        if (!h->cached) throw invalid_error("Local cache not initialized. Call cache_scores() before update_scores()");
        std::vector<int> changed(nodes, nodes + n);
        for (int v : changed)
            if (v < 0 || v >= h->e.cur.n) throw invalid_error("pbn_hc_update_scores: node index out of range");
        h->e.update_scores(changed);
    });
}

int pbn_hc_get(pbn_hc* h, double* local, double* delta_arcs, double* delta_types) {
    return guarded(hc_mu(h), [&] {
        if (!h) throw invalid_error("pbn_hc_get: null handle");
        const int n = h->e.cur.n;
        if (local && (int)h->e.local.size() == n) std::memcpy(local, h->e.local.data(), n * sizeof(double));
        const size_t cells = (size_t)h->e.cur.J() * n;
        if (delta_arcs && h->e.arcs.delta.size() == cells) std::memcpy(delta_arcs, h->e.arcs.delta.data(), cells * sizeof(double));
        if (delta_types && (int)h->e.types.delta.size() == n) std::memcpy(delta_types, h->e.types.delta.data(), n * sizeof(double));
    });
}

}  // extern "C"

extern "C" int pbn_hc_estimate(const pbn_hc_config* cfg, pbn_hc_score_fn fn, void* user, int* out_arcs, int* out_n_arcs,
                               int* out_node_types, pbn_hc_stats* stats) {
    std::recursive_mutex own;   // a search holds no shared state of the library: its callbacks lock the contexts they use
    return guarded(own, [&] {
        if (!cfg || !fn || !out_arcs || !out_n_arcs || !out_node_types) throw invalid_error("pbn_hc_estimate: null argument");
        Engine e;
        init_engine(e, cfg, fn, user);
        Model& m = e.cur;
        const int n = m.n;

        // ---- estimate_hc (hillclimbing.hpp:62-199) ------------------------------------------------------------
        const bool validated = cfg->validated != 0;
        const bool zero_patience = cfg->patience == 0;
        Model prev = m;   // prev_current_model
        Model best = m;   // value copy; `best_is_current` models the reference's pointer aliasing
        bool best_is_current = true;
        std::vector<double> vlocal;
        if (validated) {
            Batch b;
            for (int v = 0; v < n; ++v) b.add(v, m.node_type[v], m.parents[v]);
            vlocal = e.scorer.run(b, 1);
        }
        e.cache_scores();
        int p = 0;
        double accumulated_offset = 0;
        std::vector<Op> tabu;
        int iter = 0;
        int64_t near_tie_redos = 0;
        std::vector<int> trace;
        auto notify = [&](const Model& mod, const Op* op, int iteration) {  // Callback::call
            if (!cfg->on_iter) return;
            std::vector<int> arcs;
            for (int t = 0; t < n; ++t)
                for (int s : mod.parents[t]) { arcs.push_back(s); arcs.push_back(t); }
            int o[3] = {-1, 0, 0};
            if (op) { o[0] = op->kind; o[1] = op->source; o[2] = op->kind == OP_TYPE ? op->new_type : op->target; }
            if (cfg->on_iter(cfg->on_iter_user, iteration, o, op ? op->delta : 0.0, (int)arcs.size() / 2, arcs.data(),
                             mod.node_type.data()) != 0)
                throw device_error("pbn_hc_estimate: the iteration callback failed");
        };
        notify(m, nullptr, 0);
        while (iter < cfg->max_iters) {
            ++iter;
            Op best_op = zero_patience ? e.find_max(nullptr) : e.find_max(&tabu);
            if (cfg->near_tie_abs > 0.0 && best_op.valid()) {
                const Op second = e.runner_up(zero_patience ? nullptr : &tabu, best_op);
                if (second.valid() && best_op.delta - second.delta < cfg->near_tie_abs && (e.bears_ckde(best_op) || e.bears_ckde(second))) {
                    // a gap the sum-only sweeps' error budget could have decided: both deltas again from full-precision local scores
                    const double d1 = e.precise_delta(best_op), d2 = e.precise_delta(second);
                    ++near_tie_redos;
                    if (d2 > d1) { best_op = second; best_op.delta = d2; } else best_op.delta = d1;
                }
            }
            if (!best_op.valid() || (best_op.delta - cfg->epsilon) < MACHINE_TOL) break;
            m.apply(best_op);
            std::vector<int> changed = m.nodes_changed(best_op);
            double validation_delta = best_op.delta;
            if (validated) {  // validation_delta_score (hillclimbing.hpp:46-60)
                Batch b;
                for (int v : changed) b.add(v, m.node_type[v], m.parents[v]);
                std::vector<double> S = e.scorer.run(b, 1);
                double prev_s = 0, new_s = 0;
                for (size_t i = 0; i < changed.size(); ++i) {
                    prev_s += vlocal[changed[i]];
                    vlocal[changed[i]] = S[i];
                    new_s += vlocal[changed[i]];
                }
                validation_delta = new_s - prev_s;
            }
            if ((validation_delta + accumulated_offset) > MACHINE_TOL) {
                if (!zero_patience) {
                    if (p > 0) { best_is_current = true; p = 0; accumulated_offset = 0; }
                    tabu.clear();
                }
            } else {
                if (zero_patience) {
                    best = prev; best_is_current = false;
                    break;
                } else {
                    if (p == 0) { best = prev; best_is_current = false; }
                    if (++p > cfg->patience) break;
                    accumulated_offset += validation_delta;
                    tabu.push_back(m.opposite(best_op));
                }
            }
            prev.apply(best_op);
            if (stats && stats->trace && (int)trace.size() / 4 < stats->trace_capacity) {
                trace.push_back(best_op.kind);
                trace.push_back(best_op.source);
                trace.push_back(best_op.kind == OP_TYPE ? best_op.new_type : best_op.target);
                trace.push_back(0);
                if (stats->trace_delta) stats->trace_delta[trace.size() / 4 - 1] = best_op.delta;
            }
            notify(m, &best_op, iter);
            e.update_scores(changed);
        }
        const Model& res = best_is_current ? m : best;
        notify(res, nullptr, iter);
        int na = 0;
        for (int t = 0; t < n; ++t)
            for (int s : res.parents[t]) { out_arcs[2 * na] = s; out_arcs[2 * na + 1] = t; ++na; }
        *out_n_arcs = na;
        for (int i = 0; i < n; ++i) out_node_types[i] = res.node_type[i];
        if (stats) {
            stats->iterations = iter;
            stats->cells_scored = e.cells_scored;
            stats->local_score_evals = e.scorer.evals;
            stats->trace_len = (int)trace.size() / 4;
            stats->near_tie_redos = near_tie_redos;
            if (stats->trace && !trace.empty()) std::memcpy(stats->trace, trace.data(), trace.size() * sizeof(int));   // (an empty vector's data() may be null: UB for memcpy even with size 0 - found by the host sanitizer run)
        }
    });
}
