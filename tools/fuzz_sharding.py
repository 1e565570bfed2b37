"""Shape fuzz of the multi-GPU building blocks on one GPU: (1) pbn_score_terms - A(joint) - A(marginal) of random CKDE candidates is the local
score bit for bit, and totals installed in a fresh handle reproduce it without a sweep; (2) pbn_score_batch_parts - the per-part sums of random
hybrid CKDE candidates, split over a random number of ranks, added over the ranks and then in part order, are the local score bit for bit; (3) a local score is a function of (variable, parent set): another order of evaluation / of the parents, same bits.
(4) pbn_score_term_regions - a term's regions, evaluated in any order, add up to its total in region order; (5) hybrid candidates as one batch = one by one.
Random rows, dimensions, dtypes, fold counts / hold-out ratios, cardinalities.     python3 tools/fuzz_sharding.py [cases, default 30] [seed]"""
import os, sys, time
import numpy as np
import pandas as pd
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import pybnesian_amd as pbn
from pybnesian_amd import _lib

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 30
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad, t0 = 0, time.time()
for case in range(cases):
    n = int(rng.choice([int(rng.integers(400, 3000)), int(rng.integers(3000, 40000)), int(rng.integers(40000, 120000))]))
    nc = int(rng.integers(3, 7))
    dtype = "float64" if rng.random() < 0.6 else "float32"
    mix = np.eye(nc) + 0.4 * np.tril(rng.normal(size=(nc, nc)), -1)
    x = rng.normal(size=(n, nc)) @ mix.T
    x[:, 1] += np.sin(x[:, 0])
    cards = [int(rng.integers(2, 5)) for _ in range(2)]
    codes = [rng.choice(c, size=n, p=rng.dirichlet(np.full(c, 3.0))) for c in cards]
    x[:, 0] += 0.6 * codes[0]
    cont = [f"x{i}" for i in range(nc)]
    df = pd.DataFrame(x.astype(dtype), columns=cont)
    cv = rng.random() < 0.6
    k, seed, ratio = int(rng.integers(2, 8)), int(rng.integers(0, 100)), float(rng.uniform(0.1, 0.4))
    # ---- (1) continuous candidates: terms
    make = (lambda d: pbn.CVLikelihood(d, k=k, seed=seed)) if cv else (lambda d: pbn.HoldoutLikelihood(d, test_ratio=ratio, seed=seed))
    code = _lib.PBN_SCORE_CVLIK if cv else _lib.PBN_SCORE_HOLDOUT
    net = pbn.SemiparametricBN(cont, [], [(v, pbn.CKDEType()) for v in cont])
    cands = []
    for _ in range(5):
        v = int(rng.integers(nc))
        p = int(rng.integers(0, 4))
        cands.append((v, [int(q) for q in rng.choice([c for c in range(nc) if c != v], size=min(p, nc - 1), replace=False)]))
    ref = make(df)
    want = [ref.local_score(net, cont[v], [cont[q] for q in ps]) for v, ps in cands]
    terms = []
    for v, ps in cands:
        terms.append((len(ps) + 1, v) + tuple(ps))
        if ps:
            terms.append((len(ps) + 1,) + tuple(ps))
    src, dst = make(df), make(df)
    vals = src._terms("eval", code, terms)
    it = iter(vals)
    for (v, ps), w in zip(cands, want):
        j = next(it)
        m = next(it) if ps else 0.0
        if j - m != w:
            bad += 1; print("MISMATCH terms", case, n, nc, dtype, "cv" if cv else "holdout", v, ps, j - m, w)
    dst._terms("put", code, terms, vals)
    before = dst.kde_cache_stats()[1]
    got = [dst.local_score(net, cont[v], [cont[q] for q in ps]) for v, ps in cands]
    if got != want or dst.kde_cache_stats()[1] != before:
        bad += 1; print("MISMATCH installed totals", case, n, nc, dtype, got, want)
    # ---- (1c) a term's total = its regions (pbn_score_term_regions) added in region order, evaluated in a random order on a fresh handle
    regions = src._term_regions(code)
    items = [(i, f) for i in range(len(terms)) for f in range(regions)]
    rng.shuffle(items)
    reg = make(df)
    rv = reg._terms("eval_regions", code, [terms[i] for i, _ in items], regions=[f for _, f in items])
    per = np.zeros((len(terms), regions))
    for (i, f), v in zip(items, rv):
        per[i, f] = v
    for i in range(len(terms)):
        acc = 0.0
        for x in per[i].tolist():
            acc += x
        if acc != vals[i]:
            bad += 1; print("MISMATCH regions", case, n, nc, dtype, "cv" if cv else "holdout", terms[i], acc, vals[i])
    # ---- (1b) a score is a function of (variable, parent SET): another order of evaluation and of the parents gives the same bits
    other = make(df)
    for (v, ps), w in reversed(list(zip(cands, want))):
        q = list(ps)
        rng.shuffle(q)
        if other.local_score(net, cont[v], [cont[x] for x in q]) != w:
            bad += 1; print("MISMATCH order (continuous)", case, n, nc, dtype, v, ps, q)
    # ---- (2) hybrid candidates: parts
    hdf = df.copy()
    for j, (c, card) in enumerate(zip(codes, cards)):
        hdf[f"d{j}"] = pd.Categorical.from_codes(c, [f"k{i}" for i in range(card)])
    hnet = pbn.SemiparametricBN(list(hdf.columns), [], [(v, pbn.CKDEType()) for v in cont])
    hc = []
    for _ in range(3):
        v = int(rng.integers(nc))
        ps = [cont[int(q)] for q in rng.choice([c for c in range(nc) if c != v], size=int(rng.integers(0, 3)), replace=False)]
        ps += [f"d{j}" for j in range(2) if rng.random() < 0.7] or ["d0"]
        rng.shuffle(ps)
        hc.append((cont[v], list(ps)))
    href = make(hdf)
    hwant = [href.local_score(hnet, v, ps) for v, ps in hc]
    var, ntype, off, par = href._encode([(v, pbn.CKDEType(), ps) for v, ps in hc])
    world = int(rng.choice([2, 3, 5, 8, 16, 64]))
    fresh = make(hdf)
    total = np.zeros((len(hc), 64))
    for r in range(world):
        share = fresh._batch_parts(hnet, var, ntype, off, par, code, r, world)
        if np.any((share != 0) & (total != 0)):
            bad += 1; print("MISMATCH a part on two ranks", case)
        total += share
    for row, w, (v, ps) in zip(total, hwant, hc):
        acc = 0.0
        for q in row.tolist():
            acc += q
        if acc != w:
            bad += 1; print("MISMATCH parts", case, n, nc, dtype, "cv" if cv else "holdout", world, v, ps, acc, w)
            one = make(hdf)._batch_parts(hnet, var, ntype, off, par, code, 0, 1)[hc.index((v, ps))]
            d = np.nonzero(one != row)[0]
            alone = make(hdf).local_score(hnet, v, ps)
            print("   the candidate alone on a fresh handle:", alone, "| in the list after the others:", w, "| candidates:", hc)
            print("   parts differing between the one-rank and the", world, "-rank evaluation:", d.tolist(), [(one[q], row[q]) for q in d[:4]],
                  "| one-rank parts in order:", float(np.add.reduce(one)), "sequential", sum(one.tolist(), 0.0))
    # ---- (2b) the hybrid candidates as ONE batch of a fresh handle (one HybridBatch: shared chain, terms shared inside the batch)
    hbatch = make(hdf)
    if hbatch._batch(hnet, [(v, pbn.CKDEType(), ps) for v, ps in hc] * 2, code).tolist() != hwant * 2:
        bad += 1; print("MISMATCH hybrid batch", case, n, nc, dtype, hc)
    hother = make(hdf)
    for (v, ps), w in reversed(list(zip(hc, hwant))):
        q = list(ps)
        rng.shuffle(q)
        if hother.local_score(hnet, v, q) != w:
            bad += 1; print("MISMATCH order (hybrid)", case, n, nc, dtype, v, ps, q)
print(f"{cases} shapes (5 continuous + 3 hybrid candidates each) in {time.time() - t0:.0f} s: {'all ok - terms and parts add up to the local scores bit for bit, and a score does not depend on the order of evaluation or of the parents' if not bad else str(bad) + ' MISMATCHES'}")
sys.exit(1 if bad else 0)
