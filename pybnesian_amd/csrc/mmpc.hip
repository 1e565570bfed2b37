// The restriction phase in front of the hill-climb: Max-Min Parents and Children over all variables
// (learning/algorithms/mmpc.cpp:356-966) and the LinearCorrelation independence test it is usually run with
// (learning/independences/continuous/linearcorrelation.{hpp,cpp}).  Host logic; the only O(N) work - the covariance of
// all continuous columns - is one pass of the Gram kernel on the device.
//
// The CPC / to-be-checked sets are libstdc++ std::unordered_set<int>, as in the reference, because their iteration
// order decides ties: with 10^6 rows the p-values of strongly dependent pairs underflow to exactly 0, several
// candidates share the minimum, and the first one met in iteration order is the one added to the CPC.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <limits>
#include <memory>
#include <unordered_set>

#include "common.hpp"
#include "stats_kernels.hpp"

using namespace pbn;

// ---- LinearCorrelation -------------------------------------------------------------------------------------------
struct pbn_lincor {
    int n = 0;
    int64_t rows = 0;
    std::vector<double> cov;  // n x n
};

namespace {

constexpr double MACHINE_TOL = 1.4901161193847656e-08;  // util/math_constants.hpp:30
constexpr int STOP = -1, RECOMPUTE = -2;                // mmpc.cpp:16

// log(Gamma(a + 1/2) / Gamma(a)) without the cancellation of two lgamma() of ~a log a each
double lgamma_ratio_half(double a) {
    if (a < 16.0) return std::lgamma(a + 0.5) - std::lgamma(a);
    const double r = 1.0 / a;
    // Gamma(a+1/2)/Gamma(a) = sqrt(a) (1 - 1/(8a) + 1/(128a^2) + 5/(1024a^3) - 21/(32768a^4) - 399/(262144a^5) ...)
    const double s = 1.0 + r * (-1.0 / 8 + r * (1.0 / 128 + r * (5.0 / 1024 + r * (-21.0 / 32768 + r * (-399.0 / 262144 + r * (869.0 / 4194304))))));
    return 0.5 * std::log(a) + std::log(s);
}

// Regularised incomplete beta I_x(a, b) by the modified Lentz continued fraction; log_pref = log of x^a (1-x)^b / B(a,b).
double ibeta_cf(double a, double b, double x) {
    const double tiny = 1e-300, eps = 1e-16;
    double c = 1.0, d = 1.0 - (a + b) * x / (a + 1.0);
    if (std::fabs(d) < tiny) d = tiny;
    d = 1.0 / d;
    double h = d;
    for (int m = 1; m < 100000; ++m) {
        const double m2 = 2.0 * m;
        double aa = m * (b - m) * x / ((a + m2 - 1.0) * (a + m2));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        h *= d * c;
        aa = -(a + m) * (a + b + m) * x / ((a + m2) * (a + m2 + 1.0));
        d = 1.0 + aa * d; if (std::fabs(d) < tiny) d = tiny;
        c = 1.0 + aa / c; if (std::fabs(c) < tiny) c = tiny;
        d = 1.0 / d;
        const double del = d * c;
        h *= del;
        if (std::fabs(del - 1.0) < eps) break;
    }
    return h;
}

// 2 * P(T_df > |t|) = I_{df/(df+t^2)}(df/2, 1/2)   (linearcorrelation.cpp:9-13 with boost's students_t)
double two_sided_t_pvalue(double t, double df) {
    // Boost's students_t rejects df <= 0 (domain_error): fewer rows than variables + 2
    if (std::isnan(t) || !(df > 0)) return std::numeric_limits<double>::quiet_NaN();
    if (std::isinf(t)) return 0.0;
    const double t2 = t * t;
    if (t2 == 0.0) return 1.0;
    const double a = 0.5 * df, b = 0.5;
    const double x = df / (df + t2), y = t2 / (df + t2);  // y = 1 - x without cancellation
    // log B(a, 1/2) = lgamma(1/2) - log(Gamma(a + 1/2) / Gamma(a))
    const double lbeta = 0.5 * std::log(3.14159265358979323846264338327950288) - lgamma_ratio_half(a);
    const double lx = (t2 < df) ? std::log1p(-y) : std::log(x);
    const double ly = (t2 < df) ? std::log(y) : std::log1p(-x);
    const double log_pref = a * lx + b * ly - lbeta;
    if (x < (a + 1.0) / (a + b + 2.0)) {
        // tails below the smallest normal double are reported as 0: the power terms of Boost's / cephes' incomplete
        // beta underflow there, and exact zeros are what the tie-breaking of MMPC sees for such pairs
        const double p = std::exp(log_pref) * ibeta_cf(a, b, x) / a;
        return p < std::numeric_limits<double>::min() ? 0.0 : p;
    }
    return 1.0 - std::exp(log_pref) * ibeta_cf(b, a, y) / b;
}

double cor_pvalue(double cor, int64_t df) {
    const double statistic = cor * std::sqrt((double)df) / std::sqrt(1 - cor * cor);
    return two_sided_t_pvalue(std::fabs(statistic), (double)df);
}

// Symmetric eigen-decomposition by cyclic Jacobi rotations (k is the conditioning-set size + 2: a handful).
// Eigenvalues ascending in d, eigenvectors in the columns of u.
void jacobi_eigh(std::vector<double>& a, int k, std::vector<double>& d, std::vector<double>& u) {
    u.assign((size_t)k * k, 0.0);
    for (int i = 0; i < k; ++i) u[i + (size_t)i * k] = 1.0;
    auto A = [&](int i, int j) -> double& { return a[i + (size_t)j * k]; };
    auto U = [&](int i, int j) -> double& { return u[i + (size_t)j * k]; };
    for (int sweep = 0; sweep < 64; ++sweep) {
        double off = 0.0, diag = 0.0;
        for (int i = 0; i < k; ++i) {
            diag += A(i, i) * A(i, i);
            for (int j = i + 1; j < k; ++j) off += A(i, j) * A(i, j);
        }
        if (off <= 1e-34 * diag || off == 0.0) break;
        for (int p = 0; p < k - 1; ++p)
            for (int q = p + 1; q < k; ++q) {
                if (A(p, q) == 0.0) continue;
                const double theta = (A(q, q) - A(p, p)) / (2.0 * A(p, q));
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int r = 0; r < k; ++r) {
                    const double arp = A(r, p), arq = A(r, q);
                    A(r, p) = c * arp - s * arq;
                    A(r, q) = s * arp + c * arq;
                }
                for (int r = 0; r < k; ++r) {
                    const double apr = A(p, r), aqr = A(q, r);
                    A(p, r) = c * apr - s * aqr;
                    A(q, r) = s * apr + c * aqr;
                }
                for (int r = 0; r < k; ++r) {
                    const double urp = U(r, p), urq = U(r, q);
                    U(r, p) = c * urp - s * urq;
                    U(r, q) = s * urp + c * urq;
                }
            }
    }
    std::vector<int> order(k);
    for (int i = 0; i < k; ++i) order[i] = i;
    std::sort(order.begin(), order.end(), [&](int x, int y) { return A(x, x) < A(y, y); });
    d.resize(k);
    std::vector<double> us((size_t)k * k);
    for (int j = 0; j < k; ++j) {
        d[j] = A(order[j], order[j]);
        for (int i = 0; i < k; ++i) us[i + (size_t)j * k] = U(i, order[j]);
    }
    u.swap(us);
}

// cor_svd (linearcorrelation.hpp:29-47): partial correlation of the first two variables from the pseudo-inverse
double cor_from_eigen(const std::vector<double>& d, const std::vector<double>& u, int k) {
    double p11 = 0, p12 = 0, p22 = 0;
    const double tol = k * d[k - 1] * std::numeric_limits<double>::epsilon();
    for (int i = 0; i < k; ++i)
        if (d[i] > tol) {
            const double inv = 1.0 / d[i], u0 = u[0 + (size_t)i * k], u1 = u[1 + (size_t)i * k];
            p11 += u0 * u0 * inv;
            p12 += u0 * u1 * inv;
            p22 += u1 * u1 * inv;
        }
    if (p11 < MACHINE_TOL || p22 < MACHINE_TOL) return 0;
    return std::min(1.0, std::max(-1.0, -p12 / std::sqrt(p11 * p22)));
}

double lincor_pvalue(const pbn_lincor* h, int v1, int v2, int k, const int* cond) {
    const int n = h->n;
    auto C = [&](int i, int j) { return h->cov[i + (size_t)j * n]; };
    if (k == 0) {  // cor_0cond, df = N - 2
        double cor = 0;
        if (!(C(v1, v1) < MACHINE_TOL || C(v2, v2) < MACHINE_TOL))
            cor = std::min(1.0, std::max(-1.0, C(v1, v2) / std::sqrt(C(v1, v1) * C(v2, v2))));
        return cor_pvalue(cor, h->rows - 2);
    }
    const int m = k + 2;
    std::vector<int> idx(m);
    idx[0] = v1; idx[1] = v2;
    for (int i = 0; i < k; ++i) idx[i + 2] = cond[i];
    std::vector<double> a((size_t)m * m), d, u;
    for (int j = 0; j < m; ++j)
        for (int i = 0; i < m; ++i) a[i + (size_t)j * m] = C(idx[i], idx[j]);
    jacobi_eigh(a, m, d, u);
    const double cor = cor_from_eigen(d, u, m);
    // linearcorrelation.cpp:46,93: df = N - 3 for one conditioning variable; the general overload builds a (k+2)
    // matrix and uses N - 2 - (k + 2)
    return cor_pvalue(cor, k == 1 ? h->rows - 3 : h->rows - 2 - m);
}

// ---- MMPC ----------------------------------------------------------------------------------------------------------
using IntSet = std::unordered_set<int>;

struct PairSet {  // ArcSet / EdgeSet over node indices
    std::unordered_set<int64_t> s;
    int n = 0;
    bool symmetric = false;
    int64_t key(int a, int b) const { if (symmetric && a > b) std::swap(a, b); return (int64_t)a * n + b; }
    void add(int a, int b) { s.insert(key(a, b)); }
    bool has(int a, int b) const { return s.count(key(a, b)) > 0; }
};

// k-subsets of {0..m-1} in lexicographic order of their index vectors (util/combinations.hpp:87-104)
struct Lex {
    int m, k;
    std::vector<int> idx;
    bool live;
    Lex(int m_, int k_) : m(m_), k(k_), idx(k_), live(k_ <= m_) { for (int i = 0; i < k; ++i) idx[i] = i; }
    void next() {
        for (int i = k - 1; i >= 0; --i)
            if (idx[i] < m - k + i) {
                ++idx[i];
                for (int j = i + 1; j < k; ++j) idx[j] = idx[j - 1] + 1;
                return;
            }
        live = false;
    }
};

struct Mmpc {
    int n;        // all variables: nodes first, then ni interface nodes (ConditionalPartiallyDirectedGraph)
    double alpha;
    pbn_ci_pvalue_fn fn;
    void* user;
    pbn_ci_pvalue_batch_fn batch_fn = nullptr;   // optional: many independent tests per call (one device launch)
    int64_t tests = 0;
    std::vector<double> min_assoc;   // n x n, column = variable whose CPC is built
    std::vector<double> maxmin;
    std::vector<int> maxmin_idx;
    PairSet arc_wl, edge_bl, edge_wl;
    std::vector<std::pair<int, int>> arc_wl_list, edge_wl_list;
    int ni = 0;   // interface nodes: candidates of the nodes only, never of each other (mmpc.cpp:875-908)

    double pvalue(int a, int b, const std::vector<int>& cond) {
        ++tests;
        const double p = fn(user, a, b, (int)cond.size(), cond.data());
        if (std::isnan(p)) throw invalid_error("MMPC: the independence test failed");
        return p;
    }
    struct Req { int a, b; std::vector<int> cond; };
    // p-values of independent requests, in request order; one call of the batched callback when there is one
    std::vector<double> pvalues(const std::vector<Req>& reqs, bool count = true) {
        std::vector<double> out(reqs.size());
        if (!batch_fn || reqs.size() < 2) {
            for (size_t i = 0; i < reqs.size(); ++i) out[i] = pvalue(reqs[i].a, reqs[i].b, reqs[i].cond);
            if (!count) tests -= (int64_t)reqs.size();
            return out;
        }
        std::vector<int> v1, v2, off{0}, cond;
        for (const Req& r : reqs) {
            v1.push_back(r.a); v2.push_back(r.b);
            cond.insert(cond.end(), r.cond.begin(), r.cond.end());
            off.push_back((int)cond.size());
        }
        if (cond.empty()) cond.push_back(0);
        batch_fn(user, (int)reqs.size(), v1.data(), v2.data(), off.data(), cond.data(), out.data());
        if (count) tests += (int64_t)reqs.size();
        for (double p : out)
            if (std::isnan(p)) throw invalid_error("MMPC: the independence test failed");
        return out;
    }
    double pvalue(int a, int b) { static const std::vector<int> none; return pvalue(a, b, none); }
    double pvalue(int a, int b, int c) { return pvalue(a, b, std::vector<int>{c}); }

    double& cell(int row, int col) { return min_assoc[row + (size_t)col * n]; }
    void reset(int col) { maxmin[col] = alpha; maxmin_idx[col] = STOP; }
    void init_assoc(int row, int col, double p) {
        cell(row, col) = p;
        if (p < maxmin[col]) { maxmin[col] = p; maxmin_idx[col] = row; }
    }
    void update_assoc(int row, int col, double p) {
        const double nm = cell(row, col) = std::max(cell(row, col), p);
        if (nm < maxmin[col]) { maxmin[col] = nm; maxmin_idx[col] = row; }
    }

    void drop_independent(int var, IntSet& tbc) {  // update_to_be_checked
        for (auto it = tbc.begin(); it != tbc.end();)
            if (cell(*it, var) > alpha) it = tbc.erase(it); else ++it;
    }

    // association of every remaining candidate given every subset of the CPC that contains the variable added last
    void extend_assoc(int var, const IntSet& tbc, const IntSet& cpc, int last) {
        reset(var);
        // the tests of one extension step are independent of each other: requested together, applied in the
        // reference's order
        std::vector<Req> reqs;
        if (cpc.empty()) {
            for (int v : tbc) reqs.push_back({var, v, {}});
            const std::vector<double> p = pvalues(reqs);
            for (size_t i = 0; i < reqs.size(); ++i) init_assoc(reqs[i].b, var, p[i]);
            return;
        }
        std::vector<int> old;
        for (int pc : cpc) if (pc != last) old.push_back(pc);
        std::vector<int> cond;
        for (int v : tbc)
            for (int sz = 0; sz <= (int)old.size(); ++sz)   // {last}, {pc, last}, ..., the whole CPC
                for (Lex c((int)old.size(), sz); c.live; c.next()) {
                    cond.clear();
                    for (int i : c.idx) cond.push_back(old[i]);
                    cond.push_back(last);
                    reqs.push_back({var, v, cond});
                }
        const std::vector<double> p = pvalues(reqs);
        for (size_t i = 0; i < reqs.size(); ++i) update_assoc(reqs[i].b, var, p[i]);
    }

    void forward(int var, IntSet& cpc, IntSet& tbc, int last) {
        bool changed = true;
        if (cpc.empty()) {
            std::fill(min_assoc.begin() + (size_t)var * n, min_assoc.begin() + (size_t)(var + 1) * n, 0.0);
        } else if (last == RECOMPUTE) {
            // whitelisted CPC: association given the whole CPC (mmpc.cpp:357-382; the reference's loop never advances
            // its iterator there, so with a non-empty whitelist it does not terminate - here it does)
            std::vector<int> cond(cpc.begin(), cpc.end());
            reset(var);
            for (int v : tbc) init_assoc(v, var, pvalue(var, v, cond));
            const int add = maxmin_idx[var];
            if (add != STOP) { cpc.insert(add); tbc.erase(add); last = add; drop_independent(var, tbc); }
            else changed = false;
        }
        while (changed && !tbc.empty()) {
            extend_assoc(var, tbc, cpc, last);
            const int add = maxmin_idx[var];
            if (add != STOP) { cpc.insert(add); tbc.erase(add); last = add; drop_independent(var, tbc); }
            else changed = false;
        }
    }

    bool whitelisted(int var, int c) const { return edge_wl.has(var, c) || arc_wl.has(var, c) || arc_wl.has(c, var); }

    void backward(int var, IntSet& cpc) {
        if (cpc.size() <= 1) return;
        std::vector<int> rest(cpc.begin(), cpc.end());
        for (auto it = cpc.begin(); it != cpc.end();) {
            const int x = *it;
            if (whitelisted(var, x)) { ++it; continue; }
            auto pos = std::find(rest.begin(), rest.end(), x);   // swap_remove_v
            *pos = rest.back();
            rest.pop_back();
            bool separated = pvalue(var, x) > alpha;
            if (!separated && !batch_fn) {
                std::vector<int> cond;
                // subsets by increasing size, lexicographic inside a size, as the reference visits them
                for (int sz = 1; sz <= (int)rest.size() && !separated; ++sz)
                    for (Lex c((int)rest.size(), sz); c.live && !separated; c.next()) {
                        cond.clear();
                        for (int i : c.idx) cond.push_back(rest[i]);
                        separated = pvalue(var, x, cond) > alpha;
                    }
            } else if (!separated) {
                // batched callback: the subsets of one size are requested together and read in the reference's order up
                // to the first separating one - same decisions, same number of tests counted; what lies behind the
                // separating subset was evaluated for nothing, which costs less than one launch per subset
                std::vector<Req> reqs;
                std::vector<int> cond;
                for (int sz = 1; sz <= (int)rest.size() && !separated; ++sz) {
                    reqs.clear();
                    for (Lex c((int)rest.size(), sz); c.live; c.next()) {
                        cond.clear();
                        for (int i : c.idx) cond.push_back(rest[i]);
                        reqs.push_back({var, x, cond});
                    }
                    const std::vector<double> p = pvalues(reqs, false);
                    size_t used = reqs.size();
                    for (size_t i = 0; i < p.size(); ++i)
                        if (p[i] > alpha) { separated = true; used = i + 1; break; }
                    tests += (int64_t)used;
                }
            }
            if (separated) it = cpc.erase(it);
            else { rest.push_back(x); ++it; }
        }
    }

    std::vector<IntSet> run() {
        std::vector<IntSet> cpcs(n), tbc(n);
        for (auto& e : edge_wl_list) { cpcs[e.first].insert(e.second); cpcs[e.second].insert(e.first); }
        for (auto& a : arc_wl_list) { cpcs[a.first].insert(a.second); cpcs[a.second].insert(a.first); }
        const int nn = n - ni;   // nodes
        if (ni == 0) {
            for (int i = 0; i + 1 < n; ++i)
                for (int j = i + 1; j < n; ++j)
                    if (!edge_bl.has(i, j)) {
                        if (!cpcs[i].count(j)) tbc[i].insert(j);
                        if (!cpcs[j].count(i)) tbc[j].insert(i);
                    }
        } else {   // generate_cpcs of the conditional graph: every node against every other joint node
            for (int i = 0; i < nn; ++i)
                for (int j = 0; j < n; ++j)
                    if (i != j && !edge_bl.has(i, j)) {
                        if (!cpcs[i].count(j)) tbc[i].insert(j);
                        if (!cpcs[j].count(i)) tbc[j].insert(i);
                    }
        }
        min_assoc.assign((size_t)n * n, 0.0);
        maxmin.assign(n, alpha);
        maxmin_idx.assign(n, STOP);
        // marginal associations of all pairs at once (mmpc.cpp:698-738; node x interface pairs :740-784)
        // (the CPCs do not change during this pass, so the pairs can be requested together)
        std::vector<Req> mreq;
        auto marginal_pair = [&](int i, int j) {
            if ((cpcs[i].empty() || cpcs[j].empty()) && !edge_bl.has(i, j)) mreq.push_back({i, j, {}});
        };
        for (int i = 0; i + 1 < nn; ++i)
            for (int j = i + 1; j < nn; ++j) marginal_pair(i, j);
        for (int i = 0; i < nn; ++i)
            for (int j = nn; j < n; ++j) marginal_pair(i, j);
        {
            const std::vector<double> mp = pvalues(mreq);
            for (size_t q = 0; q < mreq.size(); ++q) {
                const int i = mreq[q].a, j = mreq[q].b;
                const double p = mp[q];
                if (p < alpha) {
                    if (cpcs[i].empty()) init_assoc(j, i, p);
                    if (cpcs[j].empty()) init_assoc(i, j, p);
                } else {
                    tbc[i].erase(j);
                    tbc[j].erase(i);
                }
            }
        }
        bool all_finished = true;
        for (int i = 0; i < n; ++i) {
            if (maxmin_idx[i] != STOP) {
                all_finished = false;
                cpcs[i].insert(maxmin_idx[i]);
                tbc[i].erase(maxmin_idx[i]);
            }
            if (cpcs[i].size() == 1) reset(i);
        }
        if (all_finished) return cpcs;
        // order-1 associations for the variables whose CPC holds one node; a test shared by both ends is run once
        // (mmpc.cpp:786-831)
        for (int i = 0; i < n; ++i) {
            if (cpcs[i].size() != 1) continue;
            const int c = *cpcs[i].begin();
            for (auto it = tbc[i].begin(); it != tbc[i].end();) {
                const int p = *it;
                const bool repeated = cpcs[p].size() == 1 && c == *cpcs[p].begin() && tbc[p].count(i) > 0;
                if (!repeated || i < p) {
                    const double pv = pvalue(i, p, c);
                    update_assoc(p, i, pv);
                    if (cell(p, i) > alpha) it = tbc[i].erase(it); else ++it;
                    if (repeated) {
                        update_assoc(i, p, pv);
                        if (cell(i, p) > alpha) tbc[p].erase(i);
                    }
                } else {
                    ++it;
                }
            }
        }
        for (int i = 0; i < n; ++i) {
            if (cpcs[i].size() > 1) {
                forward(i, cpcs[i], tbc[i], RECOMPUTE);
            } else if (maxmin_idx[i] != STOP) {
                const int add = maxmin_idx[i];
                cpcs[i].insert(add);
                tbc[i].erase(add);
                forward(i, cpcs[i], tbc[i], add);
            }
            backward(i, cpcs[i]);
        }
        return cpcs;
    }
};

}  // namespace

extern "C" {

int pbn_lincor_create(pbn_ctx* ctx, const pbn_table* table, pbn_lincor** out) {
    return guarded(mu_of(ctx), [&] {
        if (!ctx || !table || !out) throw invalid_error("pbn_lincor_create: null argument");
        const int n = table->n_cols;
        if (n < 2) throw invalid_error("DataFrame does not contain enough continuous columns.");
        auto h = std::make_unique<pbn_lincor>();
        h->n = n;
        h->rows = table->n_rows;
        h->cov.assign((size_t)n * n, 0.0);
        std::vector<int> cols;
        std::vector<double> mu, sse;
        // 32-column blocks so that every pair of columns meets in one Gram launch (<= 64 columns per launch)
        const int nb = (n + 31) / 32;
        for (int bi = 0; bi < nb; ++bi)
            for (int bj = bi + (nb > 1 ? 1 : 0); bj < nb; ++bj) {
                cols.clear();
                for (int c = bi * 32; c < std::min(n, bi * 32 + 32); ++c) cols.push_back(c);
                if (bj != bi)
                    for (int c = bj * 32; c < std::min(n, bj * 32 + 32); ++c) cols.push_back(c);
                const int d = (int)cols.size();
                mu.assign(d, 0.0);
                sse.assign((size_t)d * d, 0.0);
                if (pbn_table_sse(table, cols.data(), d, 0, table->n_rows, mu.data(), sse.data()) != PBN_OK)
                    throw device_error(pbn_last_error());
                for (int j = 0; j < d; ++j)
                    for (int i = 0; i < d; ++i) h->cov[cols[i] + (size_t)cols[j] * n] = sse[i + (size_t)j * d] / (double)(table->n_rows - 1);
            }
        *out = h.release();
    });
}

// Host-only construction from a covariance matrix already at hand (n x n, col-major) and the number of rows.
int pbn_lincor_from_cov(int n, int64_t rows, const double* cov, pbn_lincor** out) {
    return guarded([&] {
        if (n < 2 || !cov || !out) throw invalid_error("pbn_lincor_from_cov: bad argument");
        auto h = std::make_unique<pbn_lincor>();
        h->n = n;
        h->rows = rows;
        h->cov.assign(cov, cov + (size_t)n * n);
        *out = h.release();
    });
}

void pbn_lincor_destroy(pbn_lincor* h) { PBN_API_LOCK; delete h; }

int pbn_lincor_cov(const pbn_lincor* h, double* cov) {
    return guarded([&] {
        if (!h || !cov) throw invalid_error("pbn_lincor_cov: null argument");
        std::memcpy(cov, h->cov.data(), h->cov.size() * sizeof(double));
    });
}

// pbn_ci_pvalue_fn over a pbn_lincor handle (user = the handle); NaN on bad indices.
double pbn_lincor_pvalue(void* user, int v1, int v2, int n_cond, const int* cond) {
    const pbn_lincor* h = (const pbn_lincor*)user;
    if (!h || v1 < 0 || v2 < 0 || v1 >= h->n || v2 >= h->n || n_cond < 0 || (n_cond > 0 && !cond)) return std::nan("");
    for (int i = 0; i < n_cond; ++i)
        if (cond[i] < 0 || cond[i] >= h->n) return std::nan("");
    return lincor_pvalue(h, v1, v2, n_cond, cond);
}

static int mmpc_cpcs_impl(int n, int n_interface, pbn_ci_pvalue_fn fn, pbn_ci_pvalue_batch_fn batch_fn, void* user, double alpha, int n_arc_whitelist,
                          const int* arc_whitelist, int n_edge_blacklist, const int* edge_blacklist, int n_edge_whitelist,
                          const int* edge_whitelist, int symmetric, int* cpc_off, int* cpc, int64_t* n_tests) {
    std::recursive_mutex own;   // host-only search; the independence-test callbacks lock the contexts they use
    return guarded(own, [&] {
        if (n <= 0 || n_interface < 0 || n_interface >= n || !fn || !cpc_off || !cpc) throw invalid_error("pbn_mmpc_cpcs: bad argument");
        if (!(alpha > 0 && alpha < 1)) throw invalid_error("alpha must be a number between 0 and 1.");
        Mmpc m{n, alpha, fn, user};
        m.batch_fn = batch_fn;
        m.ni = n_interface;
        m.arc_wl.n = m.edge_bl.n = m.edge_wl.n = n;
        m.edge_bl.symmetric = m.edge_wl.symmetric = true;
        auto chk = [&](int v) { if (v < 0 || v >= n) throw invalid_error("pbn_mmpc_cpcs: node index out of range"); };
        for (int i = 0; i < n_arc_whitelist; ++i) {
            chk(arc_whitelist[2 * i]); chk(arc_whitelist[2 * i + 1]);
            m.arc_wl.add(arc_whitelist[2 * i], arc_whitelist[2 * i + 1]);
            m.arc_wl_list.push_back({arc_whitelist[2 * i], arc_whitelist[2 * i + 1]});
        }
        for (int i = 0; i < n_edge_blacklist; ++i) {
            chk(edge_blacklist[2 * i]); chk(edge_blacklist[2 * i + 1]);
            m.edge_bl.add(edge_blacklist[2 * i], edge_blacklist[2 * i + 1]);
        }
        for (int i = 0; i < n_edge_whitelist; ++i) {
            chk(edge_whitelist[2 * i]); chk(edge_whitelist[2 * i + 1]);
            m.edge_wl.add(edge_whitelist[2 * i], edge_whitelist[2 * i + 1]);
            m.edge_wl_list.push_back({edge_whitelist[2 * i], edge_whitelist[2 * i + 1]});
        }
        std::vector<IntSet> cpcs = m.run();
        if (symmetric) {  // remove_asymmetries (mmhc.cpp:12-22)
            for (int i = 0; i < n; ++i) {
                for (auto it = cpcs[i].begin(); it != cpcs[i].end();) {
                    if (!cpcs[*it].count(i)) it = cpcs[i].erase(it);
                    else ++it;
                }
            }
        }
        int pos = 0;
        for (int i = 0; i < n; ++i) {
            cpc_off[i] = pos;
            for (int v : cpcs[i]) cpc[pos++] = v;   // iteration order of the set, as the reference would walk it
        }
        cpc_off[n] = pos;
        if (n_tests) *n_tests = m.tests;
    });
}

int pbn_mmpc_cpcs(int n, pbn_ci_pvalue_fn fn, void* user, double alpha, int n_arc_whitelist, const int* arc_whitelist,
                  int n_edge_blacklist, const int* edge_blacklist, int n_edge_whitelist, const int* edge_whitelist,
                  int symmetric, int* cpc_off, int* cpc, int64_t* n_tests) {
    return mmpc_cpcs_impl(n, 0, fn, nullptr, user, alpha, n_arc_whitelist, arc_whitelist, n_edge_blacklist, edge_blacklist, n_edge_whitelist,
                          edge_whitelist, symmetric, cpc_off, cpc, n_tests);
}

// mmpc_all_variables over a ConditionalPartiallyDirectedGraph (mmpc.cpp:875-908, 740-784, 984-993): the last
// n_interface of the n variables are interface nodes.
int pbn_mmpc_cpcs_conditional(int n, int n_interface, pbn_ci_pvalue_fn fn, void* user, double alpha, int n_arc_whitelist,
                              const int* arc_whitelist, int n_edge_blacklist, const int* edge_blacklist, int n_edge_whitelist,
                              const int* edge_whitelist, int symmetric, int* cpc_off, int* cpc, int64_t* n_tests) {
    return mmpc_cpcs_impl(n, n_interface, fn, nullptr, user, alpha, n_arc_whitelist, arc_whitelist, n_edge_blacklist, edge_blacklist,
                          n_edge_whitelist, edge_whitelist, symmetric, cpc_off, cpc, n_tests);
}

// The same with a batched test callback for the steps whose tests are independent of each other (the marginal pass and
// every extension of the forward phase); fn still serves the sequential steps.  batch_fn may be NULL.
int pbn_mmpc_cpcs_batched(int n, int n_interface, pbn_ci_pvalue_fn fn, pbn_ci_pvalue_batch_fn batch_fn, void* user, double alpha,
                          int n_arc_whitelist, const int* arc_whitelist, int n_edge_blacklist, const int* edge_blacklist,
                          int n_edge_whitelist, const int* edge_whitelist, int symmetric, int* cpc_off, int* cpc, int64_t* n_tests) {
    return mmpc_cpcs_impl(n, n_interface, fn, batch_fn, user, alpha, n_arc_whitelist, arc_whitelist, n_edge_blacklist, edge_blacklist,
                          n_edge_whitelist, edge_whitelist, symmetric, cpc_off, cpc, n_tests);
}

}  // extern "C"
