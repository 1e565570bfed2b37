"""ORACLE — TEST INFRASTRUCTURE ONLY.  Serial restatement of the dynamic layer of the reference (paths under
/root/reference/pybnesian/):

  util/temporal.cpp:5-45                      temporal_name / temporal_names / temporal_slice_names
  dataset/dynamic_dataset.cpp:16-88           create_temporal_slice, create_static_df, create_transition_df
  learning/scores/scores.hpp:74-101           DynamicScoreAdaptator<Base>: one Base over the static table, one over the transition
  learning/independences/independence.hpp     table (DynamicAdaptator, dataset/dynamic_dataset.hpp:150-230); the same for tests
  learning/algorithms/dmmhc.cpp:12-118        static_blacklist; DMMHC::estimate = MMHC on the static table + conditional MMHC on the
                                              transition table
  learning/algorithms/mmhc.cpp:12-74          remove_asymmetries (inside mmpc_oracle), create_hc_blacklist,
                                              create_conditional_hc_blacklist

Everything numeric comes from the other restatements: mmpc_oracle (CPC search, LinearCorrelation p-values), hc_oracle (greedy search),
oracle.py (local scores).  PARITY UNPINNED: the reference ships no test that fixes a DMMHC structure, a DynamicScoreAdaptator value or
a static / transition table beyond their shapes (tests/dataset/dynamic_dataset_test.py checks column names and lengths: restated in
tests/test_dynamic_cpu.py); this file pins the product's decisions against an independent serial reading of the same source."""
import numpy as np

from . import hc_oracle, mmpc_oracle, oracle


def temporal_name(name, slice_index):
    return f"{name}_t_{slice_index}"


def temporal_names(variables, offset_slice, markovian_order):
    return [temporal_name(v, i) for v in variables for i in range(offset_slice, markovian_order + 1)]


def temporal_slice_names(variables, start_slice, markovian_order):
    return [[temporal_name(v, i) for v in variables] for i in range(start_slice, markovian_order + 1)]


def _temporal_slice(data, names, slice_index, slice_offset, markovian_order):
    """create_temporal_slice: rows [order - index, order - index + N - order) renamed to slice index + offset."""
    new_length = data.shape[0] - markovian_order
    offset = markovian_order - slice_index
    return [temporal_name(n, slice_index + slice_offset) for n in names], data[offset: offset + new_length]


def static_table(data, names, markovian_order):
    """create_static_df -> (column names, array): order 1 keeps every row under the names v_t_1; otherwise slices 1..order of the
    table shortened by order - 1 rows, slice-major."""
    data = np.asarray(data)
    if markovian_order == 1:
        return [temporal_name(n, 1) for n in names], data
    cols, parts = [], []
    for i in range(markovian_order):
        nm, part = _temporal_slice(data, names, i, 1, markovian_order - 1)
        cols += nm
        parts.append(part)
    return cols, np.column_stack(parts)


def transition_table(data, names, markovian_order):
    """create_temporal_slices + create_transition_df: slices 0..order of the table shortened by `order` rows, slice-major."""
    data = np.asarray(data)
    cols, parts = [], []
    for i in range(markovian_order + 1):
        nm, part = _temporal_slice(data, names, i, 0, markovian_order)
        cols += nm
        parts.append(part)
    return cols, np.column_stack(parts)


def static_blacklist(variables, markovian_order):
    """dmmhc.cpp:12-32: no arc from a more recent slice into an older one."""
    if markovian_order == 1:
        return []
    sl = temporal_slice_names(variables, 1, markovian_order)
    return [(s, d) for i in range(markovian_order - 1) for s in sl[i] for j in range(i + 1, markovian_order) for d in sl[j]]


def hc_blacklist(n, cpcs):
    """create_hc_blacklist (mmhc.cpp:24-42), node indices."""
    bl = []
    for i in range(n - 1):
        for j in range(i + 1, n):
            if j not in cpcs[i]:
                bl += [(i, j), (j, i)]
    return bl


def conditional_hc_blacklist(n, ni, cpcs):
    """create_conditional_hc_blacklist (mmhc.cpp:44-74): + interface -> node arcs outside the node's CPC."""
    bl = hc_blacklist(n, cpcs)
    for v in range(n):
        bl += [(n + k, v) for k in range(ni) if (n + k) not in cpcs[v]]
    return bl


def _table_view(cols, arr, order):
    """columns of `arr` in the order of the node list `order` (names)."""
    idx = {c: i for i, c in enumerate(cols)}
    return arr[:, [idx[c] for c in order]]


def dmmhc(data, variables, markovian_order, alpha, make_score, bn_type=0, node_types=None, op_types=False, **hc_kw):
    """DMMHC::estimate with LinearCorrelation as the independence test.

    make_score(table) -> (score(v, node_type, parents), vscore or None) over column indices of `table` (an N x m array whose
    columns are the nodes - and, for the transition part, the interface nodes behind them).
    Returns a dict: static / transition node names, CPCs, numbers of tests, arcs, node types, operator traces, cells scored."""
    data = np.asarray(data, dtype=np.float64)
    out = {}
    # ---- static part: MMHC over the slices 1..order ---------------------------------------------------------------------
    s_nodes = temporal_names(variables, 1, markovian_order)
    s_cols, s_arr = static_table(data, list(variables), markovian_order)
    S = _table_view(s_cols, s_arr, s_nodes)
    cov, _ = oracle.cov(S)
    pv = lambda a, b, c: mmpc_oracle.lincor_pvalue(cov, S.shape[0], a, b, tuple(c))
    cpcs, calls = mmpc_oracle.mmpc_all_variables(pv, len(s_nodes), alpha)
    pos = {n: i for i, n in enumerate(s_nodes)}
    bl = hc_blacklist(len(s_nodes), cpcs) + [(pos[s], pos[d]) for s, d in static_blacklist(list(variables), markovian_order)]
    score, vscore = make_score(S)
    types0 = None if node_types is None else [node_types] * len(s_nodes)
    arcs, types, trace, info = hc_oracle.estimate(len(s_nodes), bn_type, score, vscore=vscore, node_types=types0, arc_blacklist=bl,
                                                  op_types=op_types, **hc_kw)
    out["static"] = {"nodes": s_nodes, "cpcs": cpcs, "tests": calls, "arcs": sorted(arcs), "types": list(types),
                     "trace": [t[:3] for t in trace], "deltas": [t[3] for t in trace], "cells": info["cells_scored"]}
    # ---- transition part: conditional MMHC, nodes = slice 0, interface = the static nodes -----------------------------------
    t_nodes = temporal_names(variables, 0, 0)
    t_cols, t_arr = transition_table(data, list(variables), markovian_order)
    joint = t_nodes + s_nodes
    T = _table_view(t_cols, t_arr, joint)
    n, ni = len(t_nodes), len(s_nodes)
    cov, _ = oracle.cov(T)
    pv = lambda a, b, c: mmpc_oracle.lincor_pvalue(cov, T.shape[0], a, b, tuple(c))
    cpcs, calls = mmpc_oracle.mmpc_all_variables(pv, n + ni, alpha, n_interface=ni)   # the last ni of the n + ni variables are interface nodes
    bl = conditional_hc_blacklist(n, ni, cpcs)
    score, vscore = make_score(T)
    types0 = None if node_types is None else [node_types] * (n + ni)
    arcs, types, trace, info = hc_oracle.estimate(n, bn_type, score, vscore=vscore, node_types=types0, arc_blacklist=bl, op_types=op_types,
                                                  n_interface=ni, **hc_kw)
    out["transition"] = {"nodes": t_nodes, "interface": s_nodes, "cpcs": cpcs, "tests": calls, "arcs": sorted(arcs), "types": list(types),
                         "trace": [t[:3] for t in trace], "deltas": [t[3] for t in trace], "cells": info["cells_scored"]}
    return out
