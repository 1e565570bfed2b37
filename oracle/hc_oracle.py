"""ORACLE — TEST INFRASTRUCTURE ONLY.  Pure-Python, strictly serial restatement of the reference's greedy
hill-climbing: ArcOperatorSet / ChangeNodeTypeSet / OperatorPool (learning/operators/operators.{hpp,cpp}),
estimate_hc (learning/algorithms/hillclimbing.hpp:46-199) and the DAG predicates
(graph/generic_graph.hpp:2711-2745).  One local_score call per cell, exactly where the reference makes it.

`score(var, node_type, parents)` and `vscore(...)` are callables over node indices (node_type: 0 LG, 1 CKDE, 2 discrete; bn_type 0 gaussian, 1 semiparametric, 2 kde, 3 clg).
Returns (arcs, node_types, trace) with trace = [(kind, a, b, delta)], kind 0 add, 1 remove, 2 flip, 3 type.
"""
import ctypes as C
import sys

import numpy as np

from . import oracle

MACHINE_TOL = 1.4901161193847656e-08
LOWEST = -sys.float_info.max


class Model:
    def __init__(self, n, bn_type, node_types=None, arcs=(), ni=0):
        self.n, self.bn_type, self.ni = n, bn_type, ni  # 0 gaussian, 1 semiparametric, 2 kde; ni interface nodes n..n+ni-1
        self.parents = [[] for _ in range(n + ni)]
        self.children = [[] for _ in range(n + ni)]
        self.node_type = list(node_types) if node_types is not None else [1 if bn_type == 2 else 0] * (n + ni)
        for s, t in arcs:
            self.add_arc(s, t)

    def is_interface(self, v):
        return v >= self.n

    def clone(self):
        m = Model(self.n, self.bn_type, self.node_type, ni=self.ni)
        m.parents = [list(p) for p in self.parents]
        m.children = [list(c) for c in self.children]
        return m

    def has_arc(self, s, t):
        return s in self.parents[t]

    @staticmethod
    def swap_remove(v, x):
        if x in v:
            i = v.index(x)
            v[i] = v[-1]
            v.pop()

    def add_arc(self, s, t):
        if not self.has_arc(s, t):
            self.parents[t].append(s)
            self.children[s].append(t)

    def remove_arc(self, s, t):
        if self.has_arc(s, t):
            self.swap_remove(self.parents[t], s)
            self.swap_remove(self.children[s], t)

    def has_path(self, a, b, skip=None):
        seen, stack = {a}, [a]
        while stack:
            u = stack.pop()
            for c in self.children[u]:
                if skip is not None and (u, c) == skip:
                    continue
                if c == b:
                    return True
                if c not in seen:
                    seen.add(c)
                    stack.append(c)
        return False

    def can_have_arc(self, s, t):
        return not (self.node_type[t] == 2 and self.node_type[s] != 2)

    def can_add_arc(self, s, t):
        return s != t and t < self.n and self.can_have_arc(s, t) and (not self.parents[s] or not self.children[t] or not self.has_path(t, s))

    def can_flip_arc(self, s, t):
        if s == t or not self.can_have_arc(t, s):
            return False
        if self.has_arc(s, t):
            if len(self.parents[t]) == 1 or len(self.children[s]) == 1:
                return True
            return not self.has_path(s, t, skip=(s, t))
        if not self.parents[t] or not self.children[s]:
            return True
        return not self.has_path(s, t)

    def alt_type(self, v):
        return -1 if (self.bn_type != 1 or self.node_type[v] == 2) else (1 if self.node_type[v] == 0 else 0)

    def apply(self, op):
        k, a, b, _ = op
        if k == 0:
            self.add_arc(a, b)
        elif k == 1:
            self.remove_arc(a, b)
        elif k == 2:
            self.remove_arc(a, b)
            self.add_arc(b, a)
        else:
            self.node_type[a] = b


def _opposite(op, model):
    """Operator::opposite(model) evaluated AFTER apply, as estimate_hc does (hillclimbing.hpp:175).  For
    ChangeNodeType it reads model.node_type(node), i.e. the NEW type (operators.hpp:214-216)."""
    k, a, b, d = op
    if k == 0:
        return (1, a, b, -d)
    if k == 1:
        return (0, a, b, -d)
    if k == 2:
        return (2, b, a, -d)
    return (3, a, model.node_type[a], -d)


def _same(o1, o2):
    return o1[:3] == o2[:3]


def estimate(n, bn_type, score, vscore=None, node_types=None, arcs=(), arc_blacklist=(), arc_whitelist=(), type_blacklist=(),
             type_whitelist=(), op_arcs=True, op_types=False, arcs_first=True, max_indegree=0, max_iters=2 ** 31 - 1,
             epsilon=0.0, patience=0, n_interface=0, follow=None, tie_tol=1e-9):
    """follow: an operator sequence [(kind, a, b), ...] (e.g. the product's trace) to REPLAY - at every iteration the
    restatement still takes its own greedy decision from its own deltas, but then applies follow[i]; whenever the two differ
    it records (iteration, own op, followed op, |own delta - delta of the followed op|) in info["flips"], and raises
    AssertionError if that gap exceeds tie_tol * max(1, |delta|): the followed sequence is then not a greedy sequence
    under this restatement's scores.  This is the tie-flip accounting of SURVEY.md §7 hard part (a): score-equivalent
    orientations (a -> b vs b -> a under BIC / BGe) tie mathematically, so which one wins is decided in the last ulps.
    n_interface > 0: conditional network (operators.cpp:134-256,365-437; operators.hpp:526-578) - ids n .. n+ni-1 are
    interface nodes; the delta matrix is (n + ni) x n."""
    ni = n_interface
    J = n + ni
    m = Model(n, bn_type, node_types, arcs, ni)
    for v, t in type_whitelist:
        m.node_type[v] = t
    for s, t in arc_whitelist:
        if not m.has_arc(s, t):
            if s < n:
                m.remove_arc(t, s)
            m.add_arc(s, t)
    validated = vscore is not None
    zero_patience = patience == 0
    type_wl = {v for v, _ in type_whitelist}
    type_bl = set(type_blacklist)

    def local_of(mod, v):
        return score(v, mod.node_type[v], list(mod.parents[v]))

    # ---- op-set state -------------------------------------------------------------------------------------
    delta = np.full(J * n, LOWEST)
    valid = np.ones(J * n, dtype=bool)
    for s, t in arc_whitelist:
        valid[s + t * J] = False
        if s < n:
            valid[t + s * J] = False
    for s, t in arc_blacklist:
        valid[s + t * J] = False
    for i in range(n):
        valid[i + i * J] = False
    sorted_idx = np.array([i + j * J for i in range(J) for j in range(n) if valid[i + j * J]], dtype=np.int32)
    tdelta = [LOWEST] * n
    thas = [False] * n
    cells = [0]

    local = [local_of(m, v) for v in range(n)]

    def cache_arcs():
        for t in range(n):
            pt = list(m.parents[t])
            for s in range(J):
                if not valid[s + t * J] or not m.can_have_arc(s, t):
                    continue
                if m.has_arc(s, t):
                    Model.swap_remove(pt, s)
                    d = score(t, m.node_type[t], list(pt)) - local[t]
                    pt.append(s)
                elif m.has_arc(t, s):
                    ps = list(m.parents[s])
                    Model.swap_remove(ps, t)
                    pt.append(s)
                    d = score(s, m.node_type[s], ps) + score(t, m.node_type[t], list(pt)) - local[s] - local[t]
                    pt.pop()
                else:
                    pt.append(s)
                    d = score(t, m.node_type[t], list(pt)) - local[t]
                    pt.pop()
                delta[s + t * J] = d
                cells[0] += 1

    def update_types(nodes):
        for v in nodes:
            if v in type_wl:
                continue
            alt = m.alt_type(v)
            if alt < 0:
                thas[v] = False
                continue
            thas[v] = True
            if (v, alt) in type_bl:
                tdelta[v] = LOWEST
            else:
                tdelta[v] = score(v, alt, list(m.parents[v])) - local[v]
                cells[0] += 1

    def update_arcs(t):
        parents = list(m.parents[t])
        for s in range(J):
            if not valid[s + t * J]:
                continue
            if m.has_arc(s, t):
                Model.swap_remove(parents, s)
                d = score(t, m.node_type[t], list(parents)) - local[t]
                parents.append(s)
                delta[s + t * J] = d
                cells[0] += 1
                if s < n and m.can_have_arc(t, s) and valid[t + s * J]:
                    ps = list(m.parents[s]) + [t]
                    delta[t + s * J] = d + score(s, m.node_type[s], ps) - local[s]
                    cells[0] += 1
            elif s < n and m.has_arc(t, s) and m.can_have_arc(s, t):
                ps = list(m.parents[s])
                Model.swap_remove(ps, t)
                parents.append(s)
                d = score(s, m.node_type[s], ps) + score(t, m.node_type[t], list(parents)) - local[s] - local[t]
                parents.pop()
                delta[s + t * J] = d
                cells[0] += 1
            elif m.can_have_arc(s, t):
                parents.append(s)
                d = score(t, m.node_type[t], list(parents)) - local[t]
                parents.pop()
                delta[s + t * J] = d
                cells[0] += 1

    def arcs_find_max(tabu):
        oracle.lib().oracle_sort_idx_by_delta_desc(sorted_idx.ctypes.data_as(C.c_void_p), C.c_int64(sorted_idx.size),
                                                   delta.ctypes.data_as(C.c_void_p))
        for idx in sorted_idx:
            s, t = int(idx) % J, int(idx) // J
            d = float(delta[idx])
            if m.has_arc(s, t):
                op = (1, s, t, d)
            elif s >= n:   # interface source: one direction, no cycle (operators.hpp:547-556)
                if max_indegree > 0 and len(m.parents[t]) >= max_indegree:
                    continue
                if not m.can_have_arc(s, t):
                    continue
                op = (0, s, t, d)
            elif m.has_arc(t, s) and m.can_flip_arc(t, s):
                if max_indegree > 0 and len(m.parents[t]) >= max_indegree:
                    continue
                op = (2, t, s, d)
            elif m.can_add_arc(s, t):
                if max_indegree > 0 and len(m.parents[t]) >= max_indegree:
                    continue
                op = (0, s, t, d)
            else:
                continue
            if tabu and any(_same(op, x) for x in tabu):
                continue
            return op
        return None

    def types_find_max(tabu):
        best, node = LOWEST, -1
        for i in range(n):
            if i in type_wl or not thas[i]:
                continue
            if tdelta[i] > best:
                if tabu and any(_same((3, i, m.alt_type(i), 0.0), x) for x in tabu):
                    continue
                best, node = tdelta[i], i
        return (3, node, m.alt_type(node), best) if best > LOWEST else None

    def find_max(tabu):
        best, bd = None, LOWEST
        order = [arcs_find_max, types_find_max] if arcs_first else [types_find_max, arcs_find_max]
        for fn in order:
            if (fn is arcs_find_max and not op_arcs) or (fn is types_find_max and not op_types):
                continue
            op = fn(tabu)
            if op is not None and op[3] > bd:
                best, bd = op, op[3]
        return best

    # ---- estimate_hc ------------------------------------------------------------------------------------------
    prev = m.clone()
    best_model, best_is_current = None, True
    vlocal = [vscore(v, m.node_type[v], list(m.parents[v])) for v in range(n)] if validated else None
    if arcs_first:
        if op_arcs:
            cache_arcs()
        if op_types:
            update_types(range(n))
    else:
        if op_types:
            update_types(range(n))
        if op_arcs:
            cache_arcs()
    p, offset, tabu, trace, it = 0, 0.0, [], [], 0
    flips = []
    end_gain = [None]

    def delta_of(kind, a, b):
        if kind == 3:
            return tdelta[a]
        if kind == 2:                      # FlipArc(a -> b) lives in the cell of the NEW arc b -> a
            return float(delta[b + a * J])
        return float(delta[a + b * J])

    while it < max_iters:
        it += 1
        op = find_max(None if (zero_patience or not tabu) else tabu)
        if follow is not None:
            stop = op is None or (op[3] - epsilon) < MACHINE_TOL
            if it - 1 >= len(follow):
                end_gain[0] = 0.0 if stop else float(op[3])   # what the restatement would still gain where the followed trace stops
                if not stop and op[3] > tie_tol * max(1.0, abs(op[3])) + MACHINE_TOL:
                    raise AssertionError(f"followed trace stops at iteration {it} but the restatement still improves by {op[3]} with {op[:3]}")
                break
            f = tuple(follow[it - 1][:3])
            fd = delta_of(*f)
            if stop or tuple(op[:3]) != f:
                own = None if op is None else op[3]
                gap = abs((own if own is not None else 0.0) - fd)
                flips.append({"iteration": it, "own": None if op is None else list(op[:3]), "followed": list(f), "own_delta": own,
                              "followed_delta": fd, "gap": gap})
                if gap > tie_tol * max(1.0, abs(fd)):
                    raise AssertionError(f"iteration {it}: followed op {f} (delta {fd}) is not a tie of the restatement's {op}")
            op = (f[0], f[1], f[2], fd)          # replay: always applied
        elif op is None or (op[3] - epsilon) < MACHINE_TOL:
            break
        m.apply(op)
        changed = [op[1], op[2]] if op[0] == 2 else ([op[1]] if op[0] == 3 else [op[2]])
        vdelta = op[3]
        if validated:
            pv = nv = 0.0
            for v in changed:
                pv += vlocal[v]
                vlocal[v] = vscore(v, m.node_type[v], list(m.parents[v]))
                nv += vlocal[v]
            vdelta = nv - pv
        if vdelta + offset > MACHINE_TOL:
            if not zero_patience:
                if p > 0:
                    best_is_current, p, offset = True, 0, 0.0
                tabu = []
        else:
            if zero_patience:
                best_model, best_is_current = prev, False
                break
            if p == 0:
                best_model, best_is_current = prev.clone(), False
            p += 1
            if p > patience:
                break
            offset += vdelta
            tabu.append(_opposite(op, m))
        prev.apply(op)
        trace.append(op)
        for v in changed:
            local[v] = local_of(m, v)
        if op_arcs:
            for v in changed:
                update_arcs(v)
        if op_types:
            update_types(changed)
    res = m if best_is_current else best_model
    arcs_out = [(s, t) for t in range(n) for s in res.parents[t]]
    return arcs_out, list(res.node_type), trace, {"iterations": it, "cells_scored": cells[0], "flips": flips, "end_gain": end_gain[0]}
