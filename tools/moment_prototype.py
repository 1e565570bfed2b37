"""CPU prototype (numpy) of the tile-moment evaluation of csrc/kde_kernels.hip kde_moment_group_kernel (d = 2): exact sums vs (exact for non-qualifying tiles + order-P expansions about tile centroids for
qualifying ones) under the graded criterion; reports the realised relative error of the sums and the share of pairs expanded."""
import sys, os, math, numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
def hilbert(u, bits=12, scale=256.0):
    c = np.clip(np.floor(u * scale) + (1 << (bits - 1)), 0, (1 << bits) - 1).astype(np.int64)
    x, y = c[:, 0].copy(), c[:, 1].copy()
    d = np.zeros(len(u), dtype=np.int64)
    n1 = (1 << bits) - 1
    s_ = 1 << (bits - 1)
    while s_ > 0:
        rx = ((x & s_) > 0).astype(np.int64); ry = ((y & s_) > 0).astype(np.int64)
        d += s_ * s_ * ((3 * rx) ^ ry)
        m = ry == 0
        fl = m & (rx == 1)
        x = np.where(fl, n1 - x, x); y = np.where(fl, n1 - y, y)
        x, y = np.where(m, y, x), np.where(m, x, y)
        s_ >>= 1
    return d.astype(np.uint64)
P = int(os.environ.get("ORDER", "8"))
n_rows, n_cols = int(sys.argv[1]), int(sys.argv[2])
t = bench.make_dag_table(torch, torch.device('cpu'), n_rows, n_cols, 2, torch.float64, nonlinear=True).numpy()
rng = np.random.default_rng(0)
A = math.log(2.0)
alphas = [(i, j) for j in range(P + 1) for i in range(P + 1 - j)]
for pair in sys.argv[3:]:
    i, j = map(int, pair.split(','))
    x = np.column_stack([t[i], t[j]])
    perm = rng.permutation(n_rows); fold = n_rows // 10
    te, tr = x[perm[:fold]], x[perm[fold:]]
    N, d = tr.shape
    cov = np.cov(tr.T); H = cov * (4.0 / (N * (d + 2.0))) ** (2.0 / (d + 4))
    Li = np.linalg.inv(np.linalg.cholesky(H)) * np.sqrt(np.log2(np.e)); mu = tr.mean(0)
    ztr, zte = (tr - mu) @ Li.T, (te - mu) @ Li.T
    Lg = np.linalg.inv(np.linalg.cholesky(cov))
    ktr, kte = hilbert((tr - mu) @ Lg.T), hilbert((te - mu) @ Lg.T)
    ztr, zte = ztr[np.argsort(ktr, kind='stable')], zte[np.argsort(kte, kind='stable')]
    nt = N // 16
    tiles = ztr[: nt * 16].reshape(nt, 16, d)
    lo, hi, cen = tiles.min(1), tiles.max(1), tiles.mean(1)
    dl = tiles - cen[:, None, :]
    rho = np.sqrt((dl ** 2).sum(2).max(1))
    wt = np.exp2(-0.5 * (dl ** 2).sum(2))                               # [tile, row]
    # coefficients C_alpha = a^|alpha| / alpha! * sum_t w_t dx^i dy^j
    coef = np.stack([(A ** (a + b)) / (math.factorial(a) * math.factorial(b)) * (wt * dl[:, :, 0] ** a * dl[:, :, 1] ** b).sum(1) for a, b in alphas], 1)
    marg = 43 + np.log2(N / 1e6)
    ng = len(zte) // 16
    worst, qual, allp = 0.0, 0, 0
    for g in rng.choice(ng, size=150, replace=False):
        q = zte[g * 16: g * 16 + 16]
        qlo, qhi = q.min(0), q.max(0)
        gap = np.maximum(np.maximum(lo - qhi, qlo - hi), 0.0); d2min = (gap ** 2).sum(1)
        far = np.maximum(hi - qlo, qhi - lo); d2max = (far ** 2).sum(1)
        cand = np.where(-0.5 * d2min > -70)[0]
        ex = -0.5 * ((q[:, None, None, :] - tiles[cand][None]) ** 2).sum(3)      # [query, tile, row]
        tile_exact = np.exp2(ex).sum(2)                                           # [query, tile]
        S = tile_exact.sum(1)
        thr = np.log2(S.min())
        E = -0.5 * d2min[cand] - thr
        inside = E > -marg
        w = A * np.sqrt(d2max[cand]) * rho[cand]
        logR = (P + 1) * np.log2(np.maximum(w, 1e-300)) - math.log2(math.factorial(P + 1))   # (Lagrange remainder; the e^s factor is inside the term bound 2^E)
        ok = inside & (E + logR <= -(marg + 2))
        u = q[:, None, :] - cen[cand][None]                                        # [query, tile, 2]
        poly = sum(coef[cand][None, :, k] * u[:, :, 0] ** a * u[:, :, 1] ** b for k, (a, b) in enumerate(alphas))
        approx_tile = np.exp2(-0.5 * (u ** 2).sum(2)) * poly
        S2 = np.where(ok[None], approx_tile, np.where(inside[None], tile_exact, 0.0)).sum(1)
        Sx = np.where(inside[None], tile_exact, 0.0).sum(1)
        worst = max(worst, np.max(np.abs(S2 - Sx) / Sx))
        qual += ok.sum(); allp += inside.sum()
    print(f"columns {i},{j} N {N} order {P}: {qual / allp:.3f} of the visited (tile, group) pairs expanded; worst relative error of a sum {worst:.2e} (budget 1.1e-7)")
