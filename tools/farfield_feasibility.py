"""Feasibility of a far-field (tile centroid expansion) for d = 2 CKDE terms of C3 / cv64: what share of the (tile, group) pairs of the
shell between a hand-over level 2^-h and the pruning margin 2^-43 of the sum bound satisfies u_max * rho <= x0 (expansion of order 6
good to ~1e-6 relative), and how much of a query's sum sits in that shell."""
import sys, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
n_rows, n_cols, which = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3:]
t = bench.make_dag_table(torch, torch.device('cpu'), n_rows, n_cols, 2, torch.float64, nonlinear=True).numpy()
rng = np.random.default_rng(0)
import os
SCALE = float(os.environ.get("SCALE", "16"))
def morton(u, bits=12):
    c = np.clip(np.floor(u * SCALE) + (1 << (bits - 1)), 0, (1 << bits) - 1).astype(np.uint64)
    key = np.zeros(len(u), dtype=np.uint64)
    for b in range(bits):
        for i in range(u.shape[1]):
            key |= ((c[:, i] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * u.shape[1] + i)
    return key
for pair in which:
    i, j = map(int, pair.split(','))
    x = np.column_stack([t[i], t[j]])
    perm = rng.permutation(n_rows)
    fold = n_rows // 10
    te, tr = x[perm[:fold]], x[perm[fold:]]
    N, d = tr.shape
    cov = np.cov(tr.T)
    H = cov * (4.0 / (N * (d + 2.0))) ** (2.0 / (d + 4))
    L = np.linalg.cholesky(H)
    Li = np.linalg.inv(L) * np.sqrt(np.log2(np.e))        # base-2 units: term = 2^(-|dz|^2 / 2)
    mu = tr.mean(0)
    ztr, zte = (tr - mu) @ Li.T, (te - mu) @ Li.T
    Lg = np.linalg.inv(np.linalg.cholesky(cov))
    ktr, kte = morton((tr - mu) @ Lg.T), morton((te - mu) @ Lg.T)
    ztr, zte = ztr[np.argsort(ktr, kind='stable')], zte[np.argsort(kte, kind='stable')]
    nt = N // 16
    tiles = ztr[: nt * 16].reshape(nt, 16, d)
    lo, hi, cen = tiles.min(1), tiles.max(1), tiles.mean(1)
    rho = np.sqrt(((tiles - cen[:, None, :]) ** 2).sum(2).max(1))
    ng = len(zte) // 16
    gsel = rng.choice(ng, size=200, replace=False)
    marg = 43 + np.log2(N / 1e6)
    tot = {h: [0, 0, 0.0, 0.0] for h in (26, 22, 18, 14)}   # shell pairs, expandable pairs, shell mass, total mass
    near_pairs = {h: 0 for h in tot}
    for g in gsel:
        q = zte[g * 16: g * 16 + 16]
        qlo, qhi = q.min(0), q.max(0)
        gap = np.maximum(np.maximum(lo - qhi, qlo - hi), 0.0)
        d2min = (gap ** 2).sum(1)
        far = np.maximum(hi - qlo, qhi - lo)
        d2max = (far ** 2).sum(1)
        # true sums of the 16 queries (exact, all tiles within the margin of a generous bound)
        cand = np.where(-0.5 * d2min > -80)[0]
        rows = tiles[cand].reshape(-1, d)
        ex = -0.5 * ((q[:, None, :] - rows[None, :, :]) ** 2).sum(2)
        S = np.exp2(ex).sum(1)
        thr = np.log2(S.min())                                  # the group's sum bound (ideal)
        E = -0.5 * d2min - thr
        tile_mass = np.exp2(ex).reshape(16, len(cand), 16).sum(2)   # [query, tile]
        Ec = E[cand]
        x = np.sqrt(d2max[cand]) * rho[cand]
        for h in tot:
            shell = (Ec <= -h) & (Ec > -marg)
            tot[h][0] += shell.sum()
            tot[h][1] += (shell & (x <= 0.62)).sum()
            tot[h][2] += (tile_mass[:, shell].sum(1) / S).mean()
            tot[h][3] += 1
            near_pairs[h] += (Ec > -h).sum()
    print(f"columns {i},{j}: N {N}, tiles {nt}, median tile rho {np.median(rho):.3f} (base-2 units; 1 bandwidth = 1.2), margin {marg:.1f}")
    for h, (sp, ep, mass, cnt) in tot.items():
        print(f"  hand-over 2^-{h}: near pairs {near_pairs[h] / len(gsel):.0f} / group, shell pairs {sp / len(gsel):.0f} / group, expandable {ep / max(sp, 1):.3f} of them; "
              f"mean shell mass / sum {mass / cnt:.2e}")
