"""GPU tier: BIC / BGe hill-climbing WITHOUT the orientation blacklist - tie-flip accounting (SURVEY.md §7 hard part a).

Under BIC and BGe the two orientations of an arc between nodes with equal parent sets are score-equivalent: delta(a -> b) and
delta(b -> a) are equal mathematically and differ in the last ulps numerically, and `find_max` (operators.hpp:489-525) breaks
the tie by an unstable std::sort on a persistent index vector.  The reference's own test admits the winner is arbitrary
(hillclimbing_test.py:55-57).  The product computes its deltas from one-pass device moments, the oracle from the reference's
two-pass / QR arithmetic, so the last ulps differ and a tie CAN go the other way.  What must hold, and is asserted here:

 1. replaying the product's operator sequence in the oracle (`hc_oracle.estimate(follow=...)`: the restatement still takes
    its own greedy decision from its own deltas at every step), every step where the two differ is a TIE of the oracle's
    own deltas (|own best - delta of the product's operator| <= 1e-9 relative): the product's trace is a greedy trace under
    the oracle's scores up to ties.  The flips are returned with their gaps (tools/tie_flips.py commits them under
    profiles/);
 2. the replay ends where the oracle itself would stop: no operator improves the product's final graph under the oracle's
    scores (the replay raises otherwise) - the product's result is a local optimum of the reference's score, reached by a
    greedy path of the reference's score;
 3. on the reference's 4-variable table the free oracle run and the product end in the same Markov equivalence class (same
    skeleton, same v-structures).  On larger tables this is NOT guaranteed and not asserted: greedy search is path
    dependent, and after a flipped tie (measured: the very first arc of the 16-node table, gap 2e-12 on a delta of 13 891)
    the two runs walk through different DAGs and may stop in different local optima of the same score; the report carries
    both final scores.
"""
import itertools

import numpy as np
import pandas as pd
import pytest

from helpers import COLS, frame

pytestmark = pytest.mark.gpu

TIE_TOL = 1e-9


@pytest.fixture(scope="module")
def pbn():
    import pybnesian_amd

    pybnesian_amd.load_library()
    return pybnesian_amd


def dag_table(n_rows, n_cols, seed):
    rng = np.random.default_rng(seed)
    cols = []
    for j in range(n_cols):
        k = int(rng.integers(0, min(3, j) + 1))
        parents = rng.choice(j, size=k, replace=False) if k else []
        col = rng.normal(scale=rng.uniform(0.5, 1.5), size=n_rows)
        for p in parents:
            col = col + rng.uniform(-1.5, 1.5) * cols[int(p)]
        cols.append(col)
    return pd.DataFrame(np.column_stack(cols), columns=[f"x{i}" for i in range(n_cols)])


def oracle_score(data, kind):
    from oracle import oracle

    n_total = data.shape[1]

    def score(v, t, ps):
        cols = data[:, [v] + list(ps)]
        return oracle.bic_lg(cols) if kind == "bic" else oracle.bge(cols, n_total)

    return score


def equivalence_class(arcs, n):
    """(skeleton, v-structures) of a DAG given as (source, target) index pairs."""
    parents = [set() for _ in range(n)]
    for s, t in arcs:
        parents[t].add(s)
    skel = {frozenset(a) for a in arcs}
    vs = set()
    for t in range(n):
        for a, b in itertools.combinations(sorted(parents[t]), 2):
            if frozenset((a, b)) not in skel:
                vs.add((a, t, b))
    return skel, vs


def run_case(pbn, df, kind, max_indegree=0):
    """Product run, oracle replay of its trace, free oracle run.  Returns a dict with the flips and both structures."""
    from oracle import hc_oracle

    names = list(df.columns)
    idx = {c: i for i, c in enumerate(names)}
    data = df.to_numpy()
    score = pbn.BIC(df) if kind == "bic" else pbn.BGe(df)
    hc = pbn.GreedyHillClimbing()
    res = hc.estimate(pbn.ArcOperatorSet(), score, pbn.GaussianNetwork(names), max_indegree=max_indegree)
    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    trace = [(kinds[type(op)], idx[op.source()], idx[op.target()]) for op in hc.last.trace]
    deltas = [op.delta() for op in hc.last.trace]
    sc = oracle_score(data, kind)
    r_arcs, _, r_trace, r_info = hc_oracle.estimate(len(names), 0, sc, follow=trace, tie_tol=TIE_TOL, max_indegree=max_indegree)
    f_arcs, _, f_trace, f_info = hc_oracle.estimate(len(names), 0, sc, max_indegree=max_indegree)
    got_arcs = sorted((idx[s], idx[t]) for s, t in res.arcs())

    def total(arcs):
        par = [[] for _ in names]
        for s_, t_ in arcs:
            par[t_].append(s_)
        return float(sum(sc(v, 0, par[v]) for v in range(len(names))))

    return {"oracle_score_of_product_graph": total(got_arcs), "oracle_score_of_oracle_graph": total(f_arcs),"kind": kind, "nodes": len(names), "rows": len(df), "iterations": len(trace), "flips": r_info["flips"],
            "product_arcs": got_arcs, "replay_arcs": sorted(r_arcs), "oracle_arcs": sorted(f_arcs),
            "identical_trace": trace == [t[:3] for t in f_trace],
            "max_delta_rel_diff": float(max((abs(a - b[3]) / max(1.0, abs(b[3])) for a, b in zip(deltas, r_trace)), default=0.0)),
            "same_equivalence_class": equivalence_class(got_arcs, len(names)) == equivalence_class(f_arcs, len(names))}


@pytest.mark.parametrize("kind", ["bic", "bge"])
def test_reference_table_no_blacklist(pbn, golden, kind):
    out = run_case(pbn, frame(golden["train10k"][:2000]), kind)
    assert out["product_arcs"] == out["replay_arcs"]          # the replay applied exactly the product's operators
    assert all(f["gap"] <= TIE_TOL * max(1.0, abs(f["followed_delta"])) for f in out["flips"])
    assert out["max_delta_rel_diff"] < 1e-8                   # device deltas vs oracle deltas along the trace
    assert out["same_equivalence_class"], out


@pytest.mark.parametrize("kind", ["bic", "bge"])
def test_16_node_table_no_blacklist(pbn, kind):
    out = run_case(pbn, dag_table(5000, 16, 21), kind)
    assert out["product_arcs"] == out["replay_arcs"]
    assert all(f["gap"] <= TIE_TOL * max(1.0, abs(f["followed_delta"])) for f in out["flips"])
    assert out["max_delta_rel_diff"] < 1e-8
    assert out["iterations"] >= 15
    # different local optima are possible after a flipped tie (see the module docstring); both are optima of the same score
    assert abs(out["oracle_score_of_product_graph"] - out["oracle_score_of_oracle_graph"]) <= 5e-3 * abs(out["oracle_score_of_oracle_graph"])
