// CPU BASELINE — TEST / BENCH INFRASTRUCTURE ONLY.  Never linked into, imported by or called from the product path
// (pybnesian_amd/).  Only bench.py's cpu_baseline legs (and the test that pins it to pbn_oracle.cpp) use it.
//
// The same quantities as pbn_oracle.cpp's KDE / ProductKDE / CKDE log-likelihoods (kde/KDE.hpp:592-640,
// kde/ProductKDE.hpp:240-293, factors/continuous/CKDE.hpp:256-287 of /root/reference/pybnesian), written the way a CPU
// implementation that wanted to be FAST would be ("vectorised, single thread and all cores"), so that the
// GPU/CPU ratio in the bench line is not inflated by a naive port:
//   * the training set is whitened once (z = L^-1 x, or x / sqrt(h) for the diagonal bandwidth): no forward substitution
//     and no division per pair;
//   * a tile of TB = 1024 training rows (L2-resident) is reused by a block of QB = 64 test rows per thread;
//   * the exponentials are vectorised (glibc libmvec through -Ofast + omp simd), logsumexp is kept online per tile.
// pbn_oracle.cpp stays the CHECKER (it follows the reference's arithmetic operation by operation); this file is only ever
// TIMED, and tests/test_oracle_golden.py holds it to the checker at 1e-10.  Built on the box that times it:
// `make -C oracle baseline` = g++ -Ofast -march=native -fopenmp (a library built with -march=native on another CPU would
// not be portable, so it is never shipped prebuilt).
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#ifdef _OPENMP
#include <omp.h>
#endif

namespace {

constexpr double PI = 3.14159265358979323846264338327950288;
constexpr int TB = 1024;  // training rows per tile: 1024 x 8 doubles = 64 KB, L2-resident while a block of queries goes over it
constexpr int QB = 64;    // test rows per block: the training set is streamed once per 64 queries (16 made 128 threads DRAM-bound)

bool chol(const double* a, int n, double* L) {
    std::fill(L, L + (size_t)n * n, 0.0);
    for (int j = 0; j < n; ++j) {
        double s = a[j + (size_t)j * n];
        for (int k = 0; k < j; ++k) s -= L[j + (size_t)k * n] * L[j + (size_t)k * n];
        if (!(s > 0)) return false;
        L[j + (size_t)j * n] = std::sqrt(s);
        for (int i = j + 1; i < n; ++i) {
            double t = a[i + (size_t)j * n];
            for (int k = 0; k < j; ++k) t -= L[i + (size_t)k * n] * L[j + (size_t)k * n];
            L[i + (size_t)j * n] = t / L[j + (size_t)j * n];
        }
    }
    return true;
}

// rows (n x d column-major) -> whitened copy, tile-major: tile b holds its rows column by column (d x TB contiguous)
void whiten_tiles(const double* x, int64_t n, int d, const double* L /* lower, or nullptr */, const double* inv_sd, std::vector<double>& out) {
    const int64_t nt = (n + TB - 1) / TB;
    out.assign((size_t)nt * d * TB, 0.0);
#pragma omp parallel for schedule(static)
    for (int64_t b = 0; b < nt; ++b) {
        const int64_t r0 = b * TB, cnt = std::min<int64_t>(TB, n - r0);
        double* o = out.data() + (size_t)b * d * TB;
        double z[64];
        for (int64_t r = 0; r < cnt; ++r) {
            for (int c = 0; c < d; ++c) z[c] = x[r0 + r + (size_t)c * n];
            if (L) {
                for (int c = 0; c < d; ++c) {
                    double s = z[c];
                    for (int k = 0; k < c; ++k) s -= L[c + (size_t)k * d] * z[k];
                    z[c] = s / L[c + (size_t)c * d];
                }
            } else {
                for (int c = 0; c < d; ++c) z[c] *= inv_sd[c];
            }
            for (int c = 0; c < d; ++c) o[(size_t)c * TB + r] = z[c];
        }
        for (int64_t r = cnt; r < TB; ++r)  // padding rows: infinitely far away
            for (int c = 0; c < d; ++c) o[(size_t)c * TB + r] = 1e150;
    }
}

// log sum_t exp(-1/2 |zt - zq|^2) for every test row; zt tile-major (whiten_tiles), zq row-major m x d
void lse_whitened(const std::vector<double>& zt, int64_t n, int d, const double* zq, int64_t m, double* out) {
    const int64_t nt = (n + TB - 1) / TB;
    const int64_t nqb = (m + QB - 1) / QB;
#pragma omp parallel
    {
        alignas(64) double v[TB];
#pragma omp for schedule(dynamic, 1)
        for (int64_t qb = 0; qb < nqb; ++qb) {
            const int64_t q0 = qb * QB, qc = std::min<int64_t>(QB, m - q0);
            double mx[QB], sm[QB];
            for (int i = 0; i < QB; ++i) mx[i] = -1e300, sm[i] = 0.0;  // finite: built with -Ofast (no infinities assumed)
            for (int64_t b = 0; b < nt; ++b) {
                const double* tile = zt.data() + (size_t)b * d * TB;
                for (int64_t qi = 0; qi < qc; ++qi) {
                    const double* q = zq + (size_t)(q0 + qi) * d;
                    const double q0v = q[0];
                    const double* c0 = tile;
#pragma omp simd
                    for (int t = 0; t < TB; ++t) {
                        const double e = c0[t] - q0v;
                        v[t] = e * e;
                    }
                    for (int c = 1; c < d; ++c) {
                        const double qv = q[c];
                        const double* cc = tile + (size_t)c * TB;
#pragma omp simd
                        for (int t = 0; t < TB; ++t) {
                            const double e = cc[t] - qv;
                            v[t] += e * e;
                        }
                    }
                    double lo = v[0];
#pragma omp simd reduction(min : lo)
                    for (int t = 0; t < TB; ++t) lo = std::min(lo, v[t]);
                    const double tmx = -0.5 * lo;
                    if (!(tmx > -1e290)) continue;  // a tile of padding only
                    double s = 0.0;
#pragma omp simd reduction(+ : s)
                    for (int t = 0; t < TB; ++t) s += std::exp(-0.5 * v[t] - tmx);
                    if (tmx > mx[qi]) {
                        sm[qi] = sm[qi] * std::exp(mx[qi] - tmx) + s;
                        mx[qi] = tmx;
                    } else {
                        sm[qi] += s * std::exp(tmx - mx[qi]);
                    }
                }
            }
            for (int64_t qi = 0; qi < qc; ++qi) out[q0 + qi] = std::log(sm[qi]) + mx[qi];
        }
    }
}

void whiten_rows(const double* x, int64_t n, int d, const double* L, const double* inv_sd, std::vector<double>& out) {
    out.resize((size_t)n * d);
#pragma omp parallel for schedule(static)
    for (int64_t r = 0; r < n; ++r) {
        double* z = out.data() + (size_t)r * d;
        for (int c = 0; c < d; ++c) z[c] = x[r + (size_t)c * n];
        if (L) {
            for (int c = 0; c < d; ++c) {
                double s = z[c];
                for (int k = 0; k < c; ++k) s -= L[c + (size_t)k * d] * z[k];
                z[c] = s / L[c + (size_t)c * d];
            }
        } else {
            for (int c = 0; c < d; ++c) z[c] *= inv_sd[c];
        }
    }
}

int full_logl(const double* train, int64_t N, int d, const double* H, const double* test, int64_t m, double* out) {
    if (d > 64) return 1;
    std::vector<double> L((size_t)d * d);
    if (!chol(H, d, L.data())) return 2;
    double logdet = 0;
    for (int i = 0; i < d; ++i) logdet += std::log(L[i + (size_t)i * d]);
    const double lognorm = -logdet - 0.5 * d * std::log(2 * PI) - std::log((double)N);
    std::vector<double> zt, zq;
    whiten_tiles(train, N, d, L.data(), nullptr, zt);
    whiten_rows(test, m, d, L.data(), nullptr, zq);
    lse_whitened(zt, N, d, zq.data(), m, out);
    for (int64_t i = 0; i < m; ++i) out[i] += lognorm;
    return 0;
}

}  // namespace

extern "C" {

int baseline_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}
void baseline_set_num_threads(int n) {
#ifdef _OPENMP
    omp_set_num_threads(n);
#else
    (void)n;
#endif
}

// KDE.logl, full bandwidth matrix H (d x d, column-major); train N x d, test m x d, column-major
int baseline_kde_logl_f64(const double* train, int64_t N, int d, const double* H, const double* test, int64_t m, double* out) {
    return full_logl(train, N, d, H, test, m, out);
}

// ProductKDE.logl, diagonal bandwidth h (variances)
int baseline_product_kde_logl_f64(const double* train, int64_t N, int d, const double* h, const double* test, int64_t m, double* out) {
    if (d > 64) return 1;
    double slog = 0, inv_sd[64];
    for (int i = 0; i < d; ++i) {
        if (!(h[i] > 0)) return 2;
        slog += std::log(h[i]);
        inv_sd[i] = 1.0 / std::sqrt(h[i]);
    }
    const double lognorm = -0.5 * d * std::log(2 * PI) - 0.5 * slog - std::log((double)N);
    std::vector<double> zt, zq;
    whiten_tiles(train, N, d, nullptr, inv_sd, zt);
    whiten_rows(test, m, d, nullptr, inv_sd, zq);
    lse_whitened(zt, N, d, zq.data(), m, out);
    for (int64_t i = 0; i < m; ++i) out[i] += lognorm;
    return 0;
}

// CKDE.logl = joint - marginal; column 0 the variable, columns 1.. the evidence, H the joint bandwidth
int baseline_ckde_logl_f64(const double* train, int64_t N, int d, const double* H, const double* test, int64_t m, double* out) {
    int rc = full_logl(train, N, d, H, test, m, out);
    if (rc || d == 1) return rc;
    std::vector<double> Hm((size_t)(d - 1) * (d - 1)), lm(m);
    for (int j = 1; j < d; ++j)
        for (int i = 1; i < d; ++i) Hm[(i - 1) + (size_t)(j - 1) * (d - 1)] = H[i + (size_t)j * d];
    rc = full_logl(train + N, N, d - 1, Hm.data(), test + m, m, lm.data());
    if (rc) return rc;
    for (int64_t i = 0; i < m; ++i) out[i] -= lm[i];
    return 0;
}

}  // extern "C"
