// Small dense linear algebra on the host (d <= a few dozen): the O(d^3) glue the reference does with
// Eigen (LLT, SelfAdjointEigenSolver, inverse, determinant) between its device calls.  Written from
// the textbook algorithms; nothing here touches O(N) data.
#pragma once
#include <cmath>
#include <cstdint>
#include <limits>
#include <vector>

namespace pbn {
namespace hm {

// All matrices column-major, n x n, a[i + j*n].

// Cholesky A = L L^T (lower).  Returns false if a pivot is <= 0 or not finite.
inline bool cholesky(const double* a, int n, double* L) {
    for (int i = 0; i < n * n; ++i) L[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        double s = a[j + j * n];
        for (int k = 0; k < j; ++k) s -= L[j + k * n] * L[j + k * n];
        if (!(s > 0.0) || !std::isfinite(s)) return false;
        const double ljj = std::sqrt(s);
        L[j + j * n] = ljj;
        for (int i = j + 1; i < n; ++i) {
            double t = a[i + j * n];
            for (int k = 0; k < j; ++k) t -= L[i + k * n] * L[j + k * n];
            L[i + j * n] = t / ljj;
        }
    }
    return true;
}

// Inverse of a lower-triangular matrix (column-major) -> lower-triangular.
inline void lower_inverse(const double* L, int n, double* Li) {
    for (int i = 0; i < n * n; ++i) Li[i] = 0.0;
    for (int j = 0; j < n; ++j) {
        Li[j + j * n] = 1.0 / L[j + j * n];
        for (int i = j + 1; i < n; ++i) {
            double s = 0.0;
            for (int k = j; k < i; ++k) s += L[i + k * n] * Li[k + j * n];
            Li[i + j * n] = -s / L[i + i * n];
        }
    }
}

// LU with partial pivoting in place; returns determinant sign (0 if singular), perm in piv.
inline int lu(double* a, int n, int* piv) {
    int sign = 1;
    for (int k = 0; k < n; ++k) {
        int p = k;
        double best = std::fabs(a[k + k * n]);
        for (int i = k + 1; i < n; ++i) {
            double v = std::fabs(a[i + k * n]);
            if (v > best) { best = v; p = i; }
        }
        piv[k] = p;
        if (best == 0.0) return 0;
        if (p != k) {
            sign = -sign;
            for (int j = 0; j < n; ++j) std::swap(a[k + j * n], a[p + j * n]);
        }
        for (int i = k + 1; i < n; ++i) {
            a[i + k * n] /= a[k + k * n];
            const double f = a[i + k * n];
            for (int j = k + 1; j < n; ++j) a[i + j * n] -= f * a[k + j * n];
        }
    }
    return sign;
}

inline double determinant(const double* a, int n) {
    std::vector<double> t(a, a + (size_t)n * n);
    std::vector<int> piv(n);
    int s = lu(t.data(), n, piv.data());
    if (s == 0) return 0.0;
    double d = s;
    for (int i = 0; i < n; ++i) d *= t[i + i * n];
    return d;
}

// General inverse via LU; returns false when singular.
inline bool inverse(const double* a, int n, double* inv) {
    std::vector<double> t(a, a + (size_t)n * n);
    std::vector<int> piv(n);
    if (lu(t.data(), n, piv.data()) == 0) return false;
    for (int c = 0; c < n; ++c) {
        std::vector<double> b(n, 0.0);
        b[c] = 1.0;
        for (int k = 0; k < n; ++k) std::swap(b[k], b[piv[k]]);
        for (int i = 0; i < n; ++i)
            for (int k = 0; k < i; ++k) b[i] -= t[i + k * n] * b[k];
        for (int i = n - 1; i >= 0; --i) {
            for (int k = i + 1; k < n; ++k) b[i] -= t[i + k * n] * b[k];
            b[i] /= t[i + i * n];
        }
        for (int i = 0; i < n; ++i) inv[i + c * n] = b[i];
    }
    return true;
}

// Eigenvalues of a symmetric matrix by cyclic Jacobi rotations (ascending order not guaranteed).
inline void sym_eigenvalues(const double* a, int n, double* ev) {
    std::vector<double> m(a, a + (size_t)n * n);
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int j = 0; j < n; ++j)
            for (int i = 0; i < n; ++i) (i == j ? diag : off) += m[i + j * n] * m[i + j * n];
        if (off <= 1e-30 * (diag > 0 ? diag : 1.0)) break;
        for (int p = 0; p < n - 1; ++p)
            for (int q = p + 1; q < n; ++q) {
                const double apq = m[p + q * n];
                if (apq == 0.0) continue;
                const double app = m[p + p * n], aqq = m[q + q * n];
                const double theta = (aqq - app) / (2.0 * apq);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < n; ++k) {
                    const double akp = m[k + p * n], akq = m[k + q * n];
                    m[k + p * n] = c * akp - s * akq;
                    m[k + q * n] = s * akp + c * akq;
                }
                for (int k = 0; k < n; ++k) {
                    const double apk = m[p + k * n], aqk = m[q + k * n];
                    m[p + k * n] = c * apk - s * aqk;
                    m[q + k * n] = s * apk + c * aqk;
                }
            }
    }
    for (int i = 0; i < n; ++i) ev[i] = m[i + i * n];
}

// util::is_psd (/root/reference/pybnesian/util/basic_eigen_ops.hpp:136-148):
// min eigenvalue >= max eigenvalue * n * eps(T).
inline bool is_psd(const double* a, int n, bool f32) {
    std::vector<double> ev(n);
    sym_eigenvalues(a, n, ev.data());
    double mx = ev[0], mn = ev[0];
    for (int i = 1; i < n; ++i) { mx = std::max(mx, ev[i]); mn = std::min(mn, ev[i]); }
    const double eps = f32 ? (double)std::numeric_limits<float>::epsilon() : std::numeric_limits<double>::epsilon();
    return !(mn < mx * n * eps) && std::isfinite(mn);
}

}  // namespace hm
}  // namespace pbn
