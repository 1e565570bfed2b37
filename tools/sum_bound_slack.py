"""How tight is the prepass sum bound?  true min_q log2 S_q of a query group vs (a) the +-256-tile window bound with maxdist boxes,
(b) the same over ALL tiles, (c) per-query exact sums restricted to the window."""
import sys, os, numpy as np, torch
sys.path.insert(0, '/root/repo')
import bench
n_rows, n_cols = int(sys.argv[1]), int(sys.argv[2])
SCALE = float(os.environ.get("SCALE", "256"))
t = bench.make_dag_table(torch, torch.device('cpu'), n_rows, n_cols, 2, torch.float64, nonlinear=True).numpy()
rng = np.random.default_rng(0)
def morton(u, bits=12):
    c = np.clip(np.floor(u * SCALE) + (1 << (bits - 1)), 0, (1 << bits) - 1).astype(np.uint64)
    key = np.zeros(len(u), dtype=np.uint64)
    for b in range(bits):
        for i in range(u.shape[1]):
            key |= ((c[:, i] >> np.uint64(b)) & np.uint64(1)) << np.uint64(b * u.shape[1] + i)
    return key
for pair in sys.argv[3:]:
    i, j = map(int, pair.split(','))
    x = np.column_stack([t[i], t[j]])
    perm = rng.permutation(n_rows)
    fold = n_rows // 10
    te, tr = x[perm[:fold]], x[perm[fold:]]
    N, d = tr.shape
    cov = np.cov(tr.T)
    H = cov * (4.0 / (N * (d + 2.0))) ** (2.0 / (d + 4))
    Li = np.linalg.inv(np.linalg.cholesky(H)) * np.sqrt(np.log2(np.e))
    mu = tr.mean(0)
    ztr, zte = (tr - mu) @ Li.T, (te - mu) @ Li.T
    Lg = np.linalg.inv(np.linalg.cholesky(cov))
    ktr, kte = morton((tr - mu) @ Lg.T), morton((te - mu) @ Lg.T)
    otr = np.argsort(ktr, kind='stable'); ote = np.argsort(kte, kind='stable')
    ztr, zte, ktr_s, kte_s = ztr[otr], zte[ote], ktr[otr], kte[ote]
    nt = N // 16
    tiles = ztr[: nt * 16].reshape(nt, 16, d)
    lo, hi = tiles.min(1), tiles.max(1)
    ng = len(zte) // 16
    gsel = rng.choice(ng, size=300, replace=False)
    gaps_w, gaps_all = [], []
    for g in gsel:
        q = zte[g * 16: g * 16 + 16]
        qlo, qhi = q.min(0), q.max(0)
        far = np.maximum(hi - qlo, qhi - lo)
        d2max = (far ** 2).sum(1)
        gap = np.maximum(np.maximum(lo - qhi, qlo - hi), 0.0)
        d2min = (gap ** 2).sum(1)
        cand = np.where(-0.5 * d2min > -60)[0]
        rows = tiles[cand].reshape(-1, d)
        ex = -0.5 * ((q[:, None, :] - rows[None, :, :]) ** 2).sum(2)
        true = np.log2(np.exp2(ex).sum(1).min())
        pos = np.searchsorted(ktr_s, kte_s[g * 16]) // 16
        w0, w1 = max(0, pos - 256), min(nt, pos + 256)
        bw = np.log2(np.exp2(-0.5 * d2max[w0:w1]).sum()) + 4.0
        ba = np.log2(np.exp2(-0.5 * d2max).sum()) + 4.0
        gaps_w.append(true - bw); gaps_all.append(true - ba)
    gw, ga = np.array(gaps_w), np.array(gaps_all)
    print(f"columns {i},{j} N {N}: true - window bound: mean {gw.mean():.2f} bits (median {np.median(gw):.2f}, p90 {np.percentile(gw, 90):.2f}); true - all-tiles box bound: mean {ga.mean():.2f} (median {np.median(ga):.2f}, p90 {np.percentile(ga, 90):.2f})")
