"""ORACLE - TEST INFRASTRUCTURE ONLY (see oracle/oracle.py).

numpy restatement of the reference's hybrid MutualInformation (learning/independences/hybrid/mutual_information.cpp),
overload by overload, with the reference's own arithmetic: per-configuration means first, then centred sums, division
by (count - 1), determinants, entropies, chi-square tail from scipy (Boost underneath, like the reference).  It shares
no code path with the product (one-pass pilot-shifted moments, pooled by addition).

PARITY UNPINNED: the reference ships no test for its independence tests (there is no tests/learning/independences/ in
/root/reference), so nothing outside this restatement anchors these numbers; the restatement has to be read against
mutual_information.cpp line by line (the cited line ranges)."""
import numpy as np


def entropy_mvn(d, det):   # mutual_information.cpp:921-924
    with np.errstate(divide="ignore", invalid="ignore"):
        return 0.5 * d + 0.5 * d * np.log(2 * np.pi) + 0.5 * np.log(det)


def _cov_det(rows):
    """determinant of the unbiased covariance of the given rows (n x k); k = 0 -> 1 (empty determinant)."""
    n, k = rows.shape
    if k == 0:
        return 1.0
    with np.errstate(divide="ignore", invalid="ignore"):
        mean = rows.sum(axis=0) / n
        c = rows - mean
        cov = (c.T @ c) / (n - 1)
    return float(np.linalg.det(cov)) if np.all(np.isfinite(cov)) else float("nan")


class MIOracle:
    def __init__(self, columns, asymptotic_df=True):
        """columns: dict name -> float array (continuous) or (int codes, cardinality) tuple (discrete)."""
        self.cols = columns
        self.asymptotic = asymptotic_df
        self.N = len(next(iter(columns.values()))[0] if isinstance(next(iter(columns.values())), tuple) else next(iter(columns.values())))

    def is_disc(self, v):
        return isinstance(self.cols[v], tuple)

    def card(self, v):
        return self.cols[v][1]

    def codes(self, v):
        return self.cols[v][0]

    # ---- MI -------------------------------------------------------------------------------------------------------
    def mi(self, x, y, z=()):
        z = list(z)
        zD = [v for v in z if self.is_disc(v)]
        zC = [v for v in z if not self.is_disc(v)]
        if self.is_disc(x):
            if self.is_disc(y):
                return self.both_discrete(x, y, zD, zC)
            return self.mixed(x, y, zD, zC)
        if self.is_disc(y):
            return self.mixed(y, x, zD, zC)
        return self.both_continuous(x, y, zD, zC)

    def _zindex(self, zD):
        idx = np.zeros(self.N, dtype=np.int64)
        stride = 1
        for v in zD:
            idx += self.codes(v).astype(np.int64) * stride
            stride *= self.card(v)
        return idx, stride

    def _mat(self, names):
        return np.column_stack([np.asarray(self.cols[v], dtype=np.float64) for v in names]) if names else np.zeros((self.N, 0))

    def both_discrete(self, x, y, zD, zC):   # :926-955, :1391-1446, :1448-1531
        N = float(self.N)
        cx, cy = self.card(x), self.card(y)
        xi, yi = self.codes(x), self.codes(y)
        zi, zcat = self._zindex(zD)
        Z = self._mat(zC)
        zc = len(zC)
        mi = 0.0
        for k in range(zcat):
            inz = zi == k
            Nz = int(inz.sum())
            if Nz == 0:
                continue
            pz = Nz / N
            for i in range(cx):
                inxz = inz & (xi == i)
                pxz = inxz.sum() / N
                for j in range(cy):
                    sel = inxz & (yi == j)
                    Nxyz = int(sel.sum())
                    if Nxyz == 0:
                        continue
                    pyz = (inz & (yi == j)).sum() / N
                    pxyz = Nxyz / N
                    term = np.log((pz * pxyz) / (pxz * pyz))
                    if zc:
                        term -= entropy_mvn(zc, _cov_det(Z[sel]))
                    mi += pxyz * term
            if not zc:
                continue
            for i in range(cx):
                sel = inz & (xi == i)
                if sel.sum() == 0:
                    continue
                mi += (sel.sum() / N) * entropy_mvn(zc, _cov_det(Z[sel]))
            for j in range(cy):
                sel = inz & (yi == j)
                if sel.sum() == 0:
                    continue
                mi += (sel.sum() / N) * entropy_mvn(zc, _cov_det(Z[sel]))
            mi -= pz * entropy_mvn(zc, _cov_det(Z[inz]))
        return mi if not zc else _clamp(mi)

    def mixed(self, xd, yc, zD, zC):   # :957-1033, :1533-1588
        N = float(self.N)
        cx = self.card(xd)
        xi = self.codes(xd)
        zi, zcat = self._zindex(zD)
        YZ = self._mat([yc] + zC)
        zc = len(zC)
        mi = 0.0
        for k in range(zcat):
            inz = zi == k
            Nz = int(inz.sum())
            if Nz == 0:
                continue
            pz = Nz / N
            for i in range(cx):
                sel = inz & (xi == i)
                n = int(sel.sum())
                if n == 0:
                    continue
                pxz = n / N
                mi -= pxz * entropy_mvn(zc + 1, _cov_det(YZ[sel]))
                if zc:
                    mi += pxz * entropy_mvn(zc, _cov_det(YZ[sel][:, 1:]))
            mi += pz * entropy_mvn(zc + 1, _cov_det(YZ[inz]))
            if zc:
                mi -= pz * entropy_mvn(zc, _cov_det(YZ[inz][:, 1:]))
        return _clamp(mi)

    def both_continuous(self, x, y, zD, zC):   # :1055-1075, :1590-1625
        if not zD and not zC:
            cov = np.cov(np.asarray(self.cols[x], dtype=np.float64), np.asarray(self.cols[y], dtype=np.float64))
            cor = cov[0, 1] / np.sqrt(cov[0, 0] * cov[1, 1])
            return float(-0.5 * np.log(1 - cor * cor))
        N = float(self.N)
        zi, zcat = self._zindex(zD)
        XYZ = self._mat([x, y] + zC)
        zc = len(zC)
        mi = 0.0
        for k in range(zcat):
            inz = zi == k
            Nz = int(inz.sum())
            if Nz == 0:
                continue
            pz = Nz / N
            R = XYZ[inz]
            h_xyz = entropy_mvn(zc + 2, _cov_det(R))
            h_xz = entropy_mvn(zc + 1, _cov_det(R[:, [0] + list(range(2, zc + 2))]))
            h_yz = entropy_mvn(zc + 1, _cov_det(R[:, [1] + list(range(2, zc + 2))]))
            mi += pz * (h_xz + h_yz - h_xyz)
            if zc:
                mi -= pz * entropy_mvn(zc, _cov_det(R[:, 2:]))
        return _clamp(mi)

    # ---- degrees of freedom (:1093-1123, :1314-1375, :1660-1731) ------------------------------------------------------
    def df(self, x, y, z=()):
        zD = [v for v in z if self.is_disc(v)]
        zc = float(len([v for v in z if not self.is_disc(v)]))
        llz = 1
        for v in zD:
            llz *= self.card(v)
        xd, yd = self.is_disc(x), self.is_disc(y)
        if xd and yd:
            base = (self.card(x) - 1) * (self.card(y) - 1) * llz
            return base * (1 + 0.5 * (zc * (zc + 3))) if self.asymptotic else base * (1 + 0.5 * (zc * (zc + 1)))
        if xd != yd:
            llx = self.card(x if xd else y)
            return (llx - 1) * llz * (zc + 2) if self.asymptotic else (llx - 1) * llz * (zc + 1)
        return float(llz)

    def pvalue(self, x, y, z=()):
        from scipy.stats import chi2

        return float(chi2.sf(2 * self.N * self.mi(x, y, z), self.df(x, y, z)))


def _clamp(mi):   # std::max(mi, 0.) keeps a NaN first argument
    return mi if np.isnan(mi) else max(mi, 0.0)
