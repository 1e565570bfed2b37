"""ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

Pure-Python restatement of the reference's MMPC over all variables (learning/algorithms/mmpc.cpp) and of the
LinearCorrelation test (learning/independences/continuous/linearcorrelation.{hpp,cpp}), function by function.  Sets
are real libstdc++ std::unordered_set<int> objects (oracle_uset_*), because the reference's results depend on their
iteration order whenever p-values tie (exact zeros at large N).  P-values come from numpy's eigh and scipy's Student t
(Boost underneath, like the reference), i.e. from code that shares nothing with the product's Jacobi / continued
fraction.

PARITY UNPINNED: the reference has no test for LinearCorrelation or for the CPC sets of MMPC
(tests/learning/algorithms/constraint_test.py only checks that PC / MMPC run and respect white- / blacklists), so nothing
outside this restatement anchors these numbers."""
import ctypes as C
import itertools

import numpy as np

from . import oracle

STOP, RECOMPUTE = -1, -2
MACHINE_TOL = 1.4901161193847656e-08


class USet:
    def __init__(self, items=()):
        lib = oracle.lib()
        lib.oracle_uset_new.restype = C.c_void_p
        self._lib = lib
        self._h = C.c_void_p(lib.oracle_uset_new())
        for v in items:
            self.insert(v)

    def insert(self, v):
        self._lib.oracle_uset_insert(self._h, C.c_int(int(v)))

    def erase(self, v):
        self._lib.oracle_uset_erase(self._h, C.c_int(int(v)))

    def __contains__(self, v):
        return self._lib.oracle_uset_count(self._h, C.c_int(int(v))) > 0

    def __len__(self):
        return self._lib.oracle_uset_size(self._h)

    def items(self):
        n = len(self)
        buf = (C.c_int * max(n, 1))()
        self._lib.oracle_uset_items(self._h, buf)
        return [buf[i] for i in range(n)]

    def __iter__(self):
        return iter(self.items())   # snapshot: erasing while walking keeps the order of the survivors (node list unlink)

    def __del__(self):
        try:
            self._lib.oracle_uset_free(self._h)
        except Exception:
            pass


# ---- LinearCorrelation (linearcorrelation.hpp:19-60, .cpp:9-100) ---------------------------------------------------------
def cor_pvalue(cor, df):
    from scipy.stats import t

    with np.errstate(divide="ignore", invalid="ignore"):
        statistic = cor * np.sqrt(df) / np.sqrt(1 - cor * cor)
    return float(2 * t.sf(abs(statistic), df))


def cor_svd(d, u):
    tol = len(d) * d[-1] * np.finfo(float).eps
    p11 = p12 = p22 = 0.0
    for i in range(len(d)):
        if d[i] > tol:
            inv = 1.0 / d[i]
            p11 += u[0, i] * u[0, i] * inv
            p12 += u[0, i] * u[1, i] * inv
            p22 += u[1, i] * u[1, i] * inv
    if p11 < MACHINE_TOL or p22 < MACHINE_TOL:
        return 0.0
    return float(np.clip(-p12 / np.sqrt(p11 * p22), -1.0, 1.0))


def lincor_pvalue(cov, nrows, v1, v2, cond=()):
    cond = list(cond)
    if not cond:
        if cov[v1, v1] < MACHINE_TOL or cov[v2, v2] < MACHINE_TOL:
            cor = 0.0
        else:
            cor = float(np.clip(cov[v1, v2] / np.sqrt(cov[v1, v1] * cov[v2, v2]), -1.0, 1.0))
        return cor_pvalue(cor, nrows - 2)
    idx = [v1, v2] + cond
    d, u = np.linalg.eigh(cov[np.ix_(idx, idx)])
    cor = cor_svd(d, u)
    k = len(idx)
    return cor_pvalue(cor, nrows - 3 if len(cond) == 1 else nrows - 2 - k)


# ---- mmpc.cpp -------------------------------------------------------------------------------------------------------------
class Assoc:
    """BNCPCAssoc<PartiallyDirectedGraph> (mmpc.cpp:127-183)."""

    def __init__(self, n, alpha):
        self.min_assoc = np.zeros((n, n))
        self.maxmin_assoc = np.full(n, alpha)
        self.maxmin_index = np.full(n, STOP, dtype=int)
        self.alpha = alpha

    def reset_maxmin(self, col):
        self.maxmin_assoc[col] = self.alpha
        self.maxmin_index[col] = STOP

    def initialize_assoc(self, row, col, p):
        self.min_assoc[row, col] = p
        if p < self.maxmin_assoc[col]:
            self.maxmin_assoc[col] = p
            self.maxmin_index[col] = row

    def update_assoc(self, row, col, p):
        new_max = self.min_assoc[row, col] = max(self.min_assoc[row, col], p)
        if new_max < self.maxmin_assoc[col]:
            self.maxmin_assoc[col] = new_max
            self.maxmin_index[col] = row


class Counter:
    def __init__(self, fn):
        self.fn, self.calls = fn, 0

    def __call__(self, a, b, cond=()):
        self.calls += 1
        return self.fn(a, b, list(cond))


def update_min_assoc(test, variable, to_be_checked, cpc, assoc, last_added):   # mmpc.cpp:384-497
    assoc.reset_maxmin(variable)
    if len(cpc) == 0:
        for v in to_be_checked:
            assoc.initialize_assoc(v, variable, test(variable, v))
    elif len(cpc) == 1:
        for v in to_be_checked:
            assoc.update_assoc(v, variable, test(variable, v, [last_added]))
    elif len(cpc) == 2:
        cond = cpc.items()
        for v in to_be_checked:
            assoc.update_assoc(v, variable, test(variable, v, [last_added]))
            assoc.update_assoc(v, variable, test(variable, v, cond))
    else:
        old_cpc = [pc for pc in cpc if pc != last_added]
        for v in to_be_checked:
            assoc.update_assoc(v, variable, test(variable, v, [last_added]))
            for pc in old_cpc:
                assoc.update_assoc(v, variable, test(variable, v, [pc, last_added]))
            if len(cpc) > 3:
                for k in range(3, len(cpc)):   # AllSubsets(old_cpc, fixed = {last}, 3, |cpc| - 1)
                    for sub in itertools.combinations(old_cpc, k - 1):
                        assoc.update_assoc(v, variable, test(variable, v, list(sub) + [last_added]))
            assoc.update_assoc(v, variable, test(variable, v, old_cpc + [last_added]))


def update_to_be_checked(assoc, variable, to_be_checked, alpha):   # mmpc.cpp:499-508
    for v in to_be_checked:
        if assoc.min_assoc[v, variable] > alpha:
            to_be_checked.erase(v)


def mmpc_forward_phase(test, variable, alpha, cpc, to_be_checked, assoc, last_added):   # mmpc.cpp:510-554
    changed_cpc = True
    if len(cpc) == 0:
        assoc.min_assoc[:, variable] = 0
    elif last_added == RECOMPUTE:
        cpc_vec = cpc.items()
        assoc.reset_maxmin(variable)
        for v in to_be_checked:   # (the reference never advances this iterator: it does not terminate here)
            assoc.initialize_assoc(v, variable, test(variable, v, cpc_vec))
        to_add = assoc.maxmin_index[variable]
        if to_add != STOP:
            cpc.insert(to_add)
            to_be_checked.erase(to_add)
            last_added = to_add
            update_to_be_checked(assoc, variable, to_be_checked, alpha)
        else:
            changed_cpc = False
    while changed_cpc and len(to_be_checked) > 0:
        update_min_assoc(test, variable, to_be_checked, cpc, assoc, last_added)
        to_add = assoc.maxmin_index[variable]
        if to_add != STOP:
            cpc.insert(to_add)
            to_be_checked.erase(to_add)
            last_added = to_add
            update_to_be_checked(assoc, variable, to_be_checked, alpha)
        else:
            changed_cpc = False


def mmpc_backward_phase(test, variable, alpha, cpc, whitelisted):   # mmpc.cpp:561-644
    if len(cpc) <= 1:
        return
    subset_variables = cpc.items()
    for x in cpc.items():
        if whitelisted(variable, x):
            continue
        pos = subset_variables.index(x)   # swap_remove_v
        subset_variables[pos] = subset_variables[-1]
        subset_variables.pop()
        if test(variable, x) > alpha:
            cpc.erase(x)
            continue
        found = False
        for other in subset_variables:
            if test(variable, x, [other]) > alpha:
                cpc.erase(x)
                found = True
                break
        if not found and len(subset_variables) > 2:
            for k in range(2, len(subset_variables)):
                for sub in itertools.combinations(subset_variables, k):
                    if test(variable, x, list(sub)) > alpha:
                        cpc.erase(x)
                        found = True
                        break
                if found:
                    break
        if not found and len(subset_variables) > 1 and test(variable, x, subset_variables) > alpha:
            cpc.erase(x)
            found = True
        if not found:
            subset_variables.append(x)


def mmpc_all_variables(pvalue, n, alpha, arc_whitelist=(), edge_blacklist=(), edge_whitelist=(), symmetric=True, n_interface=0):
    """mmpc.cpp:833-966 (+ mmhc.cpp:12-22 remove_asymmetries).  Returns (list of CPC lists in set iteration order, #tests).
    n_interface > 0: the conditional-graph overloads (generate_cpcs :875-908, marginal pass :740-784) - the last
    n_interface of the n variables are interface nodes."""
    nn = n - n_interface
    test = Counter(pvalue)
    ebl = {(min(a, b), max(a, b)) for a, b in edge_blacklist}
    ewl = {(min(a, b), max(a, b)) for a, b in edge_whitelist}
    awl = set(map(tuple, arc_whitelist))
    whitelisted = lambda v, c: (min(v, c), max(v, c)) in ewl or (v, c) in awl or (c, v) in awl
    cpcs = [USet() for _ in range(n)]
    tbc = [USet() for _ in range(n)]
    for a, b in edge_whitelist:
        cpcs[a].insert(b)
        cpcs[b].insert(a)
    for a, b in arc_whitelist:
        cpcs[a].insert(b)
        cpcs[b].insert(a)
    if n_interface == 0:
        for i in range(n - 1):
            for j in range(i + 1, n):
                if (i, j) not in ebl:
                    if j not in cpcs[i]:
                        tbc[i].insert(j)
                    if i not in cpcs[j]:
                        tbc[j].insert(i)
    else:
        for i in range(nn):
            for j in range(n):
                if i != j and (min(i, j), max(i, j)) not in ebl:
                    if j not in cpcs[i]:
                        tbc[i].insert(j)
                    if i not in cpcs[j]:
                        tbc[j].insert(i)
    assoc = Assoc(n, alpha)

    def marginal_pair(i, j):
        if (len(cpcs[i]) == 0 or len(cpcs[j]) == 0) and (min(i, j), max(i, j)) not in ebl:
            p = test(i, j)
            if p < alpha:
                if len(cpcs[i]) == 0:
                    assoc.initialize_assoc(j, i, p)
                if len(cpcs[j]) == 0:
                    assoc.initialize_assoc(i, j, p)
            else:
                tbc[i].erase(j)
                tbc[j].erase(i)

    for i in range(nn - 1):   # marginal_cpcs_all_variables, nodes against nodes
        for j in range(i + 1, nn):
            marginal_pair(i, j)
    for i in range(nn):       # nodes against interface nodes
        for j in range(nn, n):
            marginal_pair(i, j)
    all_finished = True
    for i in range(n):
        if assoc.maxmin_index[i] != STOP:
            all_finished = False
            cpcs[i].insert(assoc.maxmin_index[i])
            tbc[i].erase(assoc.maxmin_index[i])
        if len(cpcs[i]) == 1:
            assoc.reset_maxmin(i)
    if not all_finished:
        for i in range(n):   # univariate_cpcs_all_variables
            if len(cpcs[i]) != 1:
                continue
            c = cpcs[i].items()[0]
            for p in tbc[i].items():
                repeated = len(cpcs[p]) == 1 and c == cpcs[p].items()[0] and i in tbc[p]
                if not repeated or i < p:
                    pv = test(i, p, [c])
                    assoc.update_assoc(p, i, pv)
                    if assoc.min_assoc[p, i] > alpha:
                        tbc[i].erase(p)
                    if repeated:
                        assoc.update_assoc(i, p, pv)
                        if assoc.min_assoc[i, p] > alpha:
                            tbc[p].erase(i)
        for i in range(n):
            if len(cpcs[i]) > 1:
                mmpc_forward_phase(test, i, alpha, cpcs[i], tbc[i], assoc, RECOMPUTE)
            elif assoc.maxmin_index[i] != STOP:
                add = int(assoc.maxmin_index[i])
                cpcs[i].insert(add)
                tbc[i].erase(add)
                mmpc_forward_phase(test, i, alpha, cpcs[i], tbc[i], assoc, add)
            mmpc_backward_phase(test, i, alpha, cpcs[i], whitelisted)
    if symmetric:
        for i in range(n):
            for v in cpcs[i].items():
                if i not in cpcs[v]:
                    cpcs[i].erase(v)
    return [c.items() for c in cpcs], test.calls
