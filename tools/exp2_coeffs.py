"""Coefficients of the 2^f polynomials used by kde_kernels.hip: Remez exchange (relative error) in 60-digit arithmetic.
  python tools/exp2_coeffs.py 7 0 1      -> degree 7 on [0, 1)   (v_fract_f64 range reduction)
  python tools/exp2_coeffs.py 7 -0.5 0.5 -> degree 7 on [-1/2, 1/2] (v_rndne_f64 range reduction)
  python tools/exp2_coeffs.py 6 pinned   -> degree 6 on [0, 1) with p(0) = 1 and p(1) = 2 exactly (the sweep's default)
Prints C hex-float literals (highest degree first) and the max relative error of the double-rounded polynomial."""
import sys

import mpmath as mp

mp.mp.dps = 60


def remez(deg, a, b, iters=30):
    n = deg + 2
    xs = [(a + b) / 2 + (b - a) / 2 * mp.cos(mp.pi * (n - 1 - i) / (n - 1)) for i in range(n)]
    f = lambda x: mp.mpf(2) ** x
    for _ in range(iters):
        # solve sum c_k x^k + (-1)^i E f(x_i) = f(x_i)   (relative error equi-oscillation)
        A = mp.matrix(n, n)
        rhs = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for k in range(deg + 1):
                A[i, k] = x ** k
            A[i, deg + 1] = (-1) ** i * f(x)
            rhs[i] = f(x)
        sol = mp.lu_solve(A, rhs)
        c = [sol[k] for k in range(deg + 1)]
        err = lambda x: (sum(c[k] * x ** k for k in range(deg + 1)) - f(x)) / f(x)
        # new extrema: dense scan between sign changes
        grid = [a + (b - a) * mp.mpf(i) / 4000 for i in range(4001)]
        vals = [err(x) for x in grid]
        ext = []
        for i in range(len(grid)):
            lo = vals[i - 1] if i > 0 else None
            hi = vals[i + 1] if i + 1 < len(grid) else None
            v = vals[i]
            if (lo is None or abs(v) >= abs(lo)) and (hi is None or abs(v) >= abs(hi)):
                if not ext or mp.sign(vals[ext[-1]]) != mp.sign(v):
                    ext.append(i)
                elif abs(v) > abs(vals[ext[-1]]):
                    ext[-1] = i
        if len(ext) != n:
            break
        xs = [grid[i] for i in ext]
    return c, max(abs(v) for v in vals)


def remez_pinned(deg, iters=40):
    """Minimax (relative) of 2^x on [0, 1) among p(x) = 1 + x + x (x - 1) r(x), deg r = deg - 2: exact at both ends, so
    2^x stays continuous across the integers where the v_fract range reduction wraps (an unconstrained minimax of even
    degree has errors of opposite sign at 0 and 1: a jump of twice its bound exactly where the largest term of a KDE sum
    sits, x = bias + 0)."""
    nr = deg - 1
    f2 = lambda x: mp.mpf(2) ** x
    basis = lambda k, x: x * (x - 1) * x ** k / f2(x)
    target = lambda x: (f2(x) - 1 - x) / f2(x)
    n = nr + 1
    xs = [mp.mpf(1) / 2 + mp.cos(mp.pi * (2 * (n - i) - 1) / (2 * n)) / 2 for i in range(n)]
    for _ in range(iters):
        A = mp.matrix(n, n)
        rhs = mp.matrix(n, 1)
        for i, x in enumerate(xs):
            for k in range(nr):
                A[i, k] = basis(k, x)
            A[i, nr] = (-1) ** i
            rhs[i] = target(x)
        sol = mp.lu_solve(A, rhs)
        c = [sol[k] for k in range(nr)]
        err = lambda x: sum(c[k] * basis(k, x) for k in range(nr)) - target(x)
        grid = [mp.mpf(i) / 4000 for i in range(4001)]
        vals = [err(x) for x in grid]
        ext = []
        for i in range(1, len(grid) - 1):
            v = vals[i]
            if abs(v) >= abs(vals[i - 1]) and abs(v) >= abs(vals[i + 1]):
                if not ext or mp.sign(vals[ext[-1]]) != mp.sign(v):
                    ext.append(i)
                elif abs(v) > abs(vals[ext[-1]]):
                    ext[-1] = i
        if len(ext) != n:
            break
        xs = [grid[i] for i in ext]
    co = [mp.mpf(0)] * (deg + 1)
    co[0] = co[1] = mp.mpf(1)
    for k in range(nr):
        co[k + 2] += c[k]
        co[k + 1] -= c[k]
    return co, max(abs(v) for v in vals)


def main():
    if len(sys.argv) > 2 and sys.argv[2] == "pinned":
        deg = int(sys.argv[1])
        c, e = remez_pinned(deg)
        dbl = [float(x) for x in c]
        print(f"degree {deg} on [0, 1), p(0) = 1, p(1) = 2: max relative error (exact coefficients) {mp.nstr(e, 4)}")
        for k in range(deg, -1, -1):
            print(f"  C{k} = {dbl[k].hex()}")
        worst = 0
        for i in range(20001):
            x = mp.mpf(i) / 20000
            p = mp.mpf(0)
            for k in range(deg, -1, -1):
                p = p * x + mp.mpf(dbl[k])
            worst = max(worst, abs(p / mp.mpf(2) ** x - 1))
        print(f"  double-rounded coefficients, exact Horner: {mp.nstr(worst, 4)}")
        return
    deg = int(sys.argv[1])
    a, b = mp.mpf(sys.argv[2]), mp.mpf(sys.argv[3])
    c, e = remez(deg, a, b)
    dbl = [float(x) for x in c]
    print(f"degree {deg} on [{a}, {b}]: max relative error (exact coefficients) {mp.nstr(e, 4)}")
    for k in range(deg, -1, -1):
        print(f"  C{k} = {dbl[k].hex()}")
    worst = 0
    for i in range(20001):
        x = a + (b - a) * mp.mpf(i) / 20000
        p = mp.mpf(0)
        for k in range(deg, -1, -1):
            p = p * x + mp.mpf(dbl[k])
        worst = max(worst, abs(p / mp.mpf(2) ** x - 1))
    print(f"  double-rounded coefficients, exact Horner: {mp.nstr(worst, 4)}")


if __name__ == "__main__":
    main()
