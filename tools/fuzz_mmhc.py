"""Randomised end-to-end MMHC on hybrid tables (config 5's pipeline in small, fp64): hybrid MutualInformation -> MMPC CPCs -> restricted
ValidatedLikelihood hill-climb over arcs and node types, against mmpc_oracle over mi_oracle p-values and hc_oracle over the oracle's
DiscreteAdaptator scores: same CPCs and test counts, same operator trace, arcs, node types, cells.  Random mixing weights, cardinalities,
rows, folds, hold-out ratio.   python3 tools/fuzz_mmhc.py [cases, default 10] [seed]"""
import os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn
from oracle import oracle, hc_oracle, mmpc_oracle
from oracle.mi_oracle import MIOracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 10
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 5)
bad, t0 = 0, time.time()
LG, CKDE, DISC = 0, 1, 2
for case in range(cases):
    n = int(rng.integers(1200, 3500))
    ca, cb = int(rng.integers(2, 4)), int(rng.integers(2, 4))
    A = rng.integers(0, ca, size=n)
    B = np.minimum((rng.random(n) < np.where(A == 0, 0.3, 0.6)).astype(int) + (rng.random(n) < 0.25), cb - 1)
    sh = rng.uniform(1.0, 3.0, size=4)
    x = rng.normal(loc=sh[0] * (A - 0.5), scale=1.0)
    y = rng.uniform(0.4, 0.9) * x + sh[1] * (B - 1.0) + rng.normal(scale=rng.uniform(0.4, 0.8), size=n)
    z = np.tanh(x) * sh[2] - rng.uniform(0.2, 0.6) * y + rng.normal(scale=rng.uniform(0.3, 0.6), size=n)
    w = rng.uniform(0.3, 0.8) * z + (0.8 * A if rng.random() < 0.5 else 0.0) + rng.normal(scale=0.7, size=n)
    df = pd.DataFrame({"x": x, "y": y, "z": z, "w": w})
    df["A"] = pd.Categorical.from_codes(A, [f"a{i}" for i in range(ca)])
    df["B"] = pd.Categorical.from_codes(B, [f"b{i}" for i in range(cb)])
    codes, cards = {"A": A.astype(np.int32), "B": B.astype(np.int32)}, {"A": ca, "B": cb}
    names = list(df.columns)
    col = {c: i for i, c in enumerate(names)}
    disc = {"A", "B"}
    cols = {c: ((codes[c].astype(np.int64), cards[c]) if c in disc else df[c].to_numpy()) for c in names}
    mi = MIOracle(cols)
    test = pbn.MutualInformation(df)
    ratio, k, seed = float(rng.uniform(0.15, 0.3)), int(rng.integers(2, 5)), int(rng.integers(0, 30))
    want_cpcs, calls = mmpc_oracle.mmpc_all_variables(lambda a, b, c: mi.pvalue(names[a], names[b], [names[i] for i in c]), len(names), 0.05)
    tr, te = oracle.holdout_split(n, ratio, seed)
    folds = oracle.cv_folds(tr.size, k, seed)

    def unit(v, t, ps, train, test_rows):
        var, par = names[v], [names[p] for p in ps]
        if var in disc:
            return oracle.discrete_fit_slogl(codes[var], cards[var], [codes[p] for p in par], [cards[p] for p in par], train, test_rows)
        dpar, cpar = [p for p in par if p in disc], [p for p in par if p not in disc]
        return oracle.adaptator_fit_slogl(df[[var] + cpar].to_numpy(), [codes[d] for d in dpar], [cards[d] for d in dpar], train, test_rows,
                                          "ckde" if t == CKDE else "lg")

    score = lambda v, t, ps: sum(unit(v, t, ps, tr[a], tr[b]) for a, b in folds)
    vscore = lambda v, t, ps: unit(v, t, ps, tr, te)
    types = [DISC if c in disc else LG for c in names]
    bl = [(i, j) for i in range(len(names)) for j in range(len(names)) if i != j and j not in want_cpcs[i]]
    o_arcs, o_types, o_trace, info = hc_oracle.estimate(len(names), 1, score, vscore=vscore, node_types=types, arc_blacklist=bl, op_types=True,
                                                        max_indegree=3, patience=0)
    vl = pbn.ValidatedLikelihood(df, ratio, k, seed)
    mm = pbn.MMHC()
    res = mm.estimate(test, pbn.OperatorPool([pbn.ArcOperatorSet(), pbn.ChangeNodeTypeSet()]), vl, bn_type=pbn.SemiparametricBNType(), alpha=0.05,
                      max_indegree=3)
    kinds = {pbn.AddArc: 0, pbn.RemoveArc: 1, pbn.FlipArc: 2}
    tcode = {pbn.LinearGaussianCPDType(): LG, pbn.CKDEType(): CKDE, pbn.DiscreteFactorType(): DISC}
    got_trace = [(3, col[op.node()], tcode[op.node_type()]) if isinstance(op, pbn.ChangeNodeType) else (kinds[type(op)], col[op.source()], col[op.target()])
                 for op in mm.hc.last.trace]
    # a CPC is a set (mmpc.cpp keeps std::unordered_set<int>): the order in which two members with p-values below 1e-300 came in is not compared
    got_cpcs = [[col[v] for v in c] for c in mm.last_cpcs]
    order_only = got_cpcs != want_cpcs and [sorted(c) for c in got_cpcs] == [sorted(c) for c in want_cpcs]
    if order_only:
        print(f"   (case {case}: CPC members in another order: {got_cpcs} vs {want_cpcs})")
    ok = [sorted(c) for c in got_cpcs] == [sorted(c) for c in want_cpcs] and got_trace == [t[:3] for t in o_trace] and \
        sorted((col[s], col[t]) for s, t in res.arcs()) == sorted(o_arcs) and [tcode[res.node_type(c)] for c in names] == list(o_types) and \
        mm.hc.last.cells_scored == info["cells_scored"]
    if not ok:
        bad += 1
        print(f"MISMATCH case {case}: n={n} cards=({ca},{cb}) ratio={ratio:.3f} k={k} seed={seed}\n   cpcs {mm.last_cpcs} vs {want_cpcs}\n   trace {got_trace}\n   oracle {[t[:3] for t in o_trace]}")
    print(f"case {case:3d} n={n:5d} cards=({ca},{cb}) k={k}: {len(got_trace):2d} operators, {res.num_arcs()} arcs {'ok' if ok else 'MISMATCH'}", flush=True)
print(f"{cases} MMHC runs in {time.time() - t0:.0f} s: {'all CPCs, traces, arcs, node types and cell counts equal the oracles' if not bad else str(bad) + ' MISMATCHES'}")
sys.exit(1 if bad else 0)
