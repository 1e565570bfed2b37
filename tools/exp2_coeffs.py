"""Coefficients of the 2^f polynomials used by kde_kernels.hip: Remez exchange (relative error) in 60-digit arithmetic.
  python tools/exp2_coeffs.py 7 0 1      -> degree 7 on [0, 1)   (v_fract_f64 range reduction)
  python tools/exp2_coeffs.py 7 -0.5 0.5 -> degree 7 on [-1/2, 1/2] (v_rndne_f64 range reduction)
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


def main():
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
