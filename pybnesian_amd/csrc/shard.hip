// One process per GPU: planner and driver of the sharded delta-score cache (include/pbn_hip.h "one process per GPU").
// Host code only.  Shards the serial candidate loops of learning/operators/operators.cpp:100-132,296-347 and the fold loop of
// learning/scores/cv_likelihood.cpp:5-25 of the reference, which is a single process.
//
// A batch of Score::local_score requests is cut into four kinds of work:
//   * continuous CKDE candidates of a likelihood score -> their TERMS A(S, m) (local(v | P) = A({v} u P) - A(P): A({s}) serves every child
//     of s, A({s, t}) both directions of the arc).  The unknown terms are dealt by cost, each evaluated by one rank; a batch with fewer
//     unknown terms than SPLIT_TERMS_BELOW per rank (the update batches of a search) is dealt (term, fold) by (term, fold) instead, the folds
//     added in fold order by every rank - the one-process double;
//   * CKDE candidates with discrete parents -> their slices: every rank evaluates the parts dealt to it of the engine's 64 fixed parts
//     (pbn_score_batch_parts), the per-part sums are added over the ranks (a part is non-zero on one rank only), then over the parts in order;
//   * other device-heavy candidates -> dealt whole, by variable set (candidates over one set share their sums in the engine's cache);
//   * LinearGaussian / discrete candidates and the assembly of the term-sharded ones -> every rank (host arithmetic on replicated moments).
// The plan is a pure function of the batch and of the engine's replicated state (which totals are installed), so every rank computes the
// same one; this rank evaluates its share, ONE all-gather per batch (terms | parts | whole candidates | error flag) hands every rank
// everything, and every rank assembles the same doubles.  A rank that fails while evaluating its share still enters the collective (flag set,
// NaNs) and every rank fails afterwards - the failing one with its own error, the others naming it.  What that does NOT cover: the plan
// before the collective and the assembly after it.  Both work on replicated inputs only (the batch, the installed totals, host allocations
// of a few KB), so a failure there is the same failure on every rank; a rank-LOCAL one (a device error in the light candidates' Gram
// solves, host out of memory) leaves the other ranks in the next batch's collective - bind a communicator whose collective times out
// (torch: init_process_group(timeout=...); RCCL: NCCL_ASYNC_ERROR_HANDLING), as INTEGRATION.md says.
// A batch in which nothing is dealt (LinearGaussian / discrete candidates only, every term already installed) makes NO collective call:
// the plan is the same on every rank, so every rank knows, evaluates locally, and a local error stays local.
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <string>
#include <vector>

#include "common.hpp"
#include "scoring_internal.hpp"

namespace pbn {
namespace shard {

constexpr int SPLIT_TERMS_BELOW = 4;   // batches with fewer unknown terms than this many per rank are dealt by (term, fold)
constexpr int PARTS = pbn::score::PBN_HYBRID_PARTS;

// Price of sweeping one region of a term of `dims` columns: test rows x training rows a query meets x cost per pair.  Pruned sweeps
// (d <= 4, training sets of at least 32 768 rows - kde_group_applies) meet ~ c_d N^(4/(d+4)) rows per query under the normal-reference
// bandwidth (constants: profiles/r4/prune_visits.txt, the same law hybrid.hip prices its slice parts with) at a higher cost per pair
// than the unpruned sweep (visit masks, partial tiles); per-pair weights in ps from the sweep timings at 1e6 x 1e5 rows
// (profiles/r2/sweep_dims.txt, d = 8 from the round-4 headline).  Only ratios matter.
double term_cost(int dims, int64_t ntr, int64_t nte) {
    if (dims < 1 || ntr < 1 || nte < 1) return 0.0;
    static const double cd[5] = {0, 4.6, 18.0, 67.0, 243.0};
    static const double w_pruned[5] = {0, 0.53, 0.62, 0.71, 0.81};
    const double n = (double)ntr, q = (double)nte;
    if (dims <= 4 && ntr >= 32768) return q * std::min(n, cd[dims] * std::pow(n, 4.0 / (dims + 4.0))) * w_pruned[dims];
    const double w = dims <= 4 ? 0.33 : dims == 5 ? 0.36 : dims == 6 ? 0.45 : dims == 7 ? 0.51 : 0.46 + 0.05 * std::max(0, (dims - 1) / 4 - 1);
    return q * n * w;
}

// longest processing time first: decreasing cost, equal costs by decreasing tie, then by index; each to the least loaded rank
void deal(const std::vector<double>& cost, const uint32_t* tie, int world, double* load_io, std::vector<int>& owner) {
    const int n = (int)cost.size();
    std::vector<int> order(n);
    for (int i = 0; i < n; ++i) order[i] = i;
    std::stable_sort(order.begin(), order.end(), [&](int a, int b) {
        if (cost[a] != cost[b]) return cost[a] > cost[b];
        if (tie && tie[a] != tie[b]) return tie[a] > tie[b];
        return false;
    });
    std::vector<double> own_load;
    double* load = load_io;
    if (!load) { own_load.assign((size_t)world, 0.0); load = own_load.data(); }
    owner.assign((size_t)n, 0);
    for (int i : order) {
        int best = 0;
        for (int r = 1; r < world; ++r)
            if (load[r] < load[best]) best = r;
        owner[i] = best;
        load[best] += cost[i];
    }
}

// equal costs (the initial cache: every set a pair) are ordered by a hash of the set, not by appearance: in order of appearance rank r
// gets the sets i = r mod world, i.e. the pairs of the SAME few variables - and a variable whose sweeps prune badly made its rank 25 %
// slower than the mean of eight; scattered, the ranks' sums differ by a few percent
uint32_t mix(const int* v, int n) {
    uint32_t h = 0x9E3779B9u;
    for (int i = 0; i < n; ++i) {
        h = (h ^ ((uint32_t)v[i] + 0x7F4A7C15u)) * 0x85EBCA6Bu;
        h ^= h >> 13;
    }
    return h;
}

struct Failure { int rc = PBN_OK; std::string msg; };

[[noreturn]] void rethrow(const Failure& f) {
    if (f.rc == PBN_ERR_INVALID) throw invalid_error(f.msg);
    if (f.rc == PBN_ERR_SINGULAR) throw singular_error(f.msg);
    throw device_error(f.msg);
}

// engine / collective calls: a non-zero status becomes the matching exception with the message the callee left
void must(int rc, const char* what) {
    if (rc == PBN_OK) return;
    Failure f{rc, pbn_last_error()};
    if (f.msg.empty()) f.msg = std::string(what) + " failed";
    rethrow(f);
}

struct Sub {   // a sub-batch in the layout of pbn_score_batch
    std::vector<int> var, nt, off{0}, par;
    void add(int v, int t, const int* p, int np) { var.push_back(v); nt.push_back(t); par.insert(par.end(), p, p + np); off.push_back((int)par.size()); }
    const int* parp() const { static const int zero = 0; return par.empty() ? &zero : par.data(); }
    int n() const { return (int)var.size(); }
};

struct TermList {   // terms in the layout of pbn_score_terms
    std::vector<int> off{0}, vars, m;
    void add(const std::vector<int>& key) { m.push_back(key[0]); vars.insert(vars.end(), key.begin() + 1, key.end()); off.push_back((int)vars.size()); }
    int n() const { return (int)m.size(); }
};

void run(const pbn_shard_engine* eng, const pbn_comm* comm, int kind, int n, const int* var, const int* ntype, const int* off, const int* par,
         bool shard_all, double* out) {
    if (!eng || !eng->batch) throw invalid_error("pbn_shard_batch: the engine needs at least `batch`");
    if (n < 0 || (n > 0 && (!var || !off || !out))) throw invalid_error("pbn_shard_batch: null argument");
    if (n == 0) return;
    if (!comm) {   // one process
        must(eng->batch(eng->user, kind, n, var, ntype, off, par, out), "engine batch");
        return;
    }
    if (!comm->all_gather || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world) throw invalid_error("pbn_shard_batch: bad communicator");
    const int rank = comm->rank, world = comm->world, n_cont = eng->n_cont;
    const bool likelihood = kind == PBN_SCORE_CVLIK || kind == PBN_SCORE_HOLDOUT;
    auto type_of = [&](int i) { return ntype ? ntype[i] : (int)PBN_NODE_LG; };
    auto np_of = [&](int i) { return off[i + 1] - off[i]; };
    int regions = 1;
    int64_t ntr = 0, nte = 0;
    if (eng->shape) must(eng->shape(eng->user, kind, &regions, &ntr, &nte), "engine shape");
    if (regions < 1) regions = 1;
    if (ntr < 1 || nte < 1) { ntr = 1000000; nte = 100000; }   // an engine that does not say: the shape the per-pair weights were measured on
    auto price = [&](int dims) { return eng->term_price ? eng->term_price(eng->user, kind, dims) : term_cost(dims, ntr, nte); };

    // ---- the plan ------------------------------------------------------------------------------------------------------------------
    std::vector<char> heavy((size_t)n, 0), taken((size_t)n, 0);   // taken: evaluated through the collective (terms assembled later count as light)
    for (int i = 0; i < n; ++i) heavy[i] = shard_all || type_of(i) == PBN_NODE_CKDE;
    // (a) terms of the continuous CKDE candidates
    std::vector<int> by_term;
    if (likelihood && eng->terms && eng->terms_missing && eng->terms_put)
        for (int i = 0; i < n; ++i) {
            if (!heavy[i] || type_of(i) != PBN_NODE_CKDE || var[i] >= n_cont) continue;
            bool cont = true;
            for (int j = off[i]; j < off[i + 1]; ++j) cont = cont && par[j] < n_cont;
            if (cont) by_term.push_back(i);
        }
    std::vector<std::vector<int>> todo;                 // unknown terms: [m, sorted columns...]
    std::vector<std::vector<int>> term_lists((size_t)world);
    std::vector<std::pair<int, int>> items;             // (term, fold) dealing: item -> (index in todo, region)
    bool by_fold = false;
    if (!by_term.empty()) {   // (one candidate too: its one or two terms are dealt fold by fold instead of being swept by every rank)
        std::vector<std::vector<int>> terms;
        std::map<std::vector<int>, int> seen;
        for (int i : by_term) {
            const int p = np_of(i), d = p + 1;
            std::vector<int> joint{d, var[i]}, marg{d};
            joint.insert(joint.end(), par + off[i], par + off[i + 1]);
            marg.insert(marg.end(), par + off[i], par + off[i + 1]);
            std::sort(joint.begin() + 1, joint.end());
            std::sort(marg.begin() + 1, marg.end());
            if (seen.emplace(joint, (int)terms.size()).second) terms.push_back(joint);
            if (p > 0 && seen.emplace(marg, (int)terms.size()).second) terms.push_back(marg);
        }
        TermList tl;
        for (const auto& t : terms) tl.add(t);
        std::vector<int> missing((size_t)tl.n(), 0);
        must(eng->terms_missing(eng->user, kind, tl.n(), tl.off.data(), tl.vars.data(), tl.m.data(), missing.data()), "engine terms_missing");
        for (int j = 0; j < tl.n(); ++j)
            if (missing[j]) todo.push_back(terms[j]);
        by_fold = !todo.empty() && regions > 1 && eng->term_regions && (int)todo.size() < SPLIT_TERMS_BELOW * world;
        std::vector<double> cost;
        std::vector<uint32_t> tie;
        std::vector<int> owner;
        if (by_fold) {
            for (int j = 0; j < (int)todo.size(); ++j)
                for (int f = 0; f < regions; ++f) { items.emplace_back(j, f); cost.push_back(price((int)todo[j].size() - 1)); }
            deal(cost, nullptr, world, nullptr, owner);
        } else if (!todo.empty()) {
            for (const auto& t : todo) { cost.push_back(regions * price((int)t.size() - 1)); tie.push_back(mix(t.data() + 1, (int)t.size() - 1)); }
            deal(cost, tie.data(), world, nullptr, owner);
        }
        for (size_t i = 0; i < owner.size(); ++i) term_lists[(size_t)owner[i]].push_back((int)i);
        for (int i : by_term) heavy[i] = 0;   // assembled by every rank from the shared terms
    }
    // (b) the slices of CKDE candidates with discrete parents
    std::vector<int> sliced;
    if (likelihood && eng->batch_parts && world <= PARTS)
        for (int i = 0; i < n; ++i) {
            if (!heavy[i] || type_of(i) != PBN_NODE_CKDE || var[i] >= n_cont) continue;
            bool disc = false;
            for (int j = off[i]; j < off[i + 1]; ++j) disc = disc || par[j] >= n_cont;
            if (disc) { sliced.push_back(i); heavy[i] = 0; taken[i] = 1; }
        }
    // (c) whole candidates, by variable set
    std::vector<int> whole;
    for (int i = 0; i < n; ++i)
        if (heavy[i]) whole.push_back(i);
    std::vector<std::vector<int>> whole_lists((size_t)world);
    if (!whole.empty()) {   // (a lone one goes to one rank: evaluated once in the job)
        std::map<std::vector<int>, int> set_of;
        std::vector<std::vector<int>> sets;
        std::vector<int> count, cand_set;
        for (int i : whole) {
            std::vector<int> key{var[i]};
            key.insert(key.end(), par + off[i], par + off[i + 1]);
            std::sort(key.begin(), key.end());
            auto it = set_of.emplace(key, (int)sets.size());
            if (it.second) { sets.push_back(key); count.push_back(0); }
            ++count[(size_t)it.first->second];
            cand_set.push_back(it.first->second);
        }
        std::vector<double> cost;
        std::vector<uint32_t> tie;
        for (size_t s = 0; s < sets.size(); ++s) {   // one joint sweep per set + one marginal sweep per candidate
            const int d = (int)sets[s].size();
            cost.push_back(regions * (price(d) + count[s] * price(d - 1)));
            tie.push_back(mix(sets[s].data(), d));
        }
        std::vector<int> owner;
        deal(cost, tie.data(), world, nullptr, owner);
        for (size_t j = 0; j < whole.size(); ++j) { whole_lists[(size_t)owner[(size_t)cand_set[j]]].push_back(whole[j]); taken[whole[j]] = 1; }
    }

    // ---- this rank's share -----------------------------------------------------------------------------------------------------------
    size_t per_terms = 0, per_whole = 0;
    for (int r = 0; r < world; ++r) { per_terms = std::max(per_terms, term_lists[(size_t)r].size()); per_whole = std::max(per_whole, whole_lists[(size_t)r].size()); }
    const size_t n_parts = sliced.size() * (size_t)PARTS;
    if (per_terms + n_parts + per_whole == 0) {   // nothing dealt: no collective (every rank takes this branch: the plan is replicated)
        must(eng->batch(eng->user, kind, n, var, ntype, off, par, out), "engine batch");
        return;
    }
    const size_t count = per_terms + n_parts + per_whole + 1;
    const size_t o_parts = per_terms, o_whole = per_terms + n_parts, o_flag = count - 1;
    std::vector<double> send(count, 0.0), recv(count * (size_t)world, 0.0);
    Failure failure;
    Sub sl;
    for (int i : sliced) sl.add(var[i], type_of(i), par + off[i], np_of(i));
    try {
        const std::vector<int>& mine = term_lists[(size_t)rank];
        if (!mine.empty()) {
            TermList tl;
            std::vector<int> reg;
            for (int i : mine) {
                tl.add(todo[(size_t)(by_fold ? items[(size_t)i].first : i)]);
                if (by_fold) reg.push_back(items[(size_t)i].second);
            }
            if (by_fold) must(eng->term_regions(eng->user, kind, tl.n(), tl.off.data(), tl.vars.data(), tl.m.data(), reg.data(), send.data()), "engine term_regions");
            else must(eng->terms(eng->user, kind, tl.n(), tl.off.data(), tl.vars.data(), tl.m.data(), send.data()), "engine terms");
        }
        if (!sliced.empty())
            must(eng->batch_parts(eng->user, kind, sl.n(), sl.var.data(), sl.nt.data(), sl.off.data(), sl.parp(), rank, world, send.data() + o_parts), "engine batch_parts");
        const std::vector<int>& wm = whole_lists[(size_t)rank];
        if (!wm.empty()) {
            Sub sb;
            for (int i : wm) sb.add(var[i], type_of(i), par + off[i], np_of(i));
            must(eng->batch(eng->user, kind, sb.n(), sb.var.data(), sb.nt.data(), sb.off.data(), sb.parp(), send.data() + o_whole), "engine batch");
        }
    } catch (const invalid_error& e) { failure = {PBN_ERR_INVALID, e.what()};
    } catch (const singular_error& e) { failure = {PBN_ERR_SINGULAR, e.what()};
    } catch (const std::exception& e) { failure = {PBN_ERR_DEVICE, e.what()}; }
    if (failure.rc != PBN_OK) {   // never skip the collective: the other ranks are already on their way to it
        std::fill(send.begin(), send.end(), std::nan(""));
        send[o_flag] = 1.0;
    }

    // ---- the batch's one collective ----------------------------------------------------------------------------------------------------
    if (comm->all_gather(comm->user, send.data(), (int64_t)count, recv.data()) != 0) {
        if (failure.rc != PBN_OK) rethrow(failure);
        throw device_error("pbn_shard_batch: the host's all_gather failed");
    }
    {
        std::string bad;
        for (int r = 0; r < world; ++r)
            if (recv[(size_t)r * count + o_flag] != 0.0) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
        if (failure.rc != PBN_OK) rethrow(failure);
        if (!bad.empty()) throw device_error("sharded_batch: rank(s) [" + bad + "] failed while computing their share of the batch");
    }

    // ---- install the terms, assemble ---------------------------------------------------------------------------------------------------
    if (!todo.empty()) {
        TermList tl;
        std::vector<double> values;
        if (by_fold) {
            std::vector<double> per(todo.size() * (size_t)regions, 0.0);
            for (int r = 0; r < world; ++r)
                for (size_t q = 0; q < term_lists[(size_t)r].size(); ++q) {
                    const auto& it = items[(size_t)term_lists[(size_t)r][q]];
                    per[(size_t)it.first * regions + it.second] = recv[(size_t)r * count + q];
                }
            for (size_t j = 0; j < todo.size(); ++j) {
                double acc = 0.0;
                for (int f = 0; f < regions; ++f) acc += per[j * regions + f];   // the folds in order, as the engine adds them
                tl.add(todo[j]);
                values.push_back(acc);
            }
        } else {
            for (int r = 0; r < world; ++r)
                for (size_t q = 0; q < term_lists[(size_t)r].size(); ++q) {
                    tl.add(todo[(size_t)term_lists[(size_t)r][q]]);
                    values.push_back(recv[(size_t)r * count + q]);
                }
        }
        must(eng->terms_put(eng->user, kind, tl.n(), tl.off.data(), tl.vars.data(), tl.m.data(), values.data()), "engine terms_put");
    }
    {   // light candidates and the term-sharded ones (from the shared terms): every rank
        Sub rest;
        std::vector<int> idx;
        for (int i = 0; i < n; ++i)
            if (!taken[i]) { rest.add(var[i], type_of(i), par + off[i], np_of(i)); idx.push_back(i); }
        if (rest.n() > 0) {
            std::vector<double> vals((size_t)rest.n());
            must(eng->batch(eng->user, kind, rest.n(), rest.var.data(), rest.nt.data(), rest.off.data(), rest.parp(), vals.data()), "engine batch");
            for (size_t q = 0; q < idx.size(); ++q) out[idx[q]] = vals[q];
        }
    }
    for (size_t j = 0; j < sliced.size(); ++j) {
        double total[PARTS];
        for (int q = 0; q < PARTS; ++q) total[q] = 0.0;
        for (int r = 0; r < world; ++r)                 // exact: every part is non-zero on one rank only
            for (int q = 0; q < PARTS; ++q) total[q] += recv[(size_t)r * count + o_parts + j * PARTS + q];
        double acc = 0.0;
        for (int q = 0; q < PARTS; ++q) acc += total[q];   // the parts in order, as the engine adds them
        out[sliced[j]] = acc;
    }
    for (int r = 0; r < world; ++r)
        for (size_t q = 0; q < whole_lists[(size_t)r].size(); ++q) out[whole_lists[(size_t)r][q]] = recv[(size_t)r * count + o_whole + q];
}

// ---- the engine of a pbn_scoredata -----------------------------------------------------------------------------------------------------
struct Bound { pbn_scoredata* sd; const double* params; int n_params; };

int b_shape(void* u, int kind, int* regions, int64_t* ntr, int64_t* nte) {
    const pbn_scoredata* sd = ((Bound*)u)->sd;
    if (kind == PBN_SCORE_CVLIK && sd->k > 0) {
        *regions = sd->k;
        const int64_t fold = sd->n_cv / sd->k;
        *ntr = sd->n_cv - fold;
        *nte = fold;
    } else {
        *regions = 1;
        *ntr = sd->n_cv;
        *nte = sd->n_hold;
    }
    return PBN_OK;
}
// the price list corrected by what THIS engine's kernels cost (tools/term_prices.py, profiles/r6/term_prices.txt: 500 000-row fp64 table,
// 10 folds - terms of 1 / 2 / 3 / 4 variables 6.0 / 13.0 / 14.1 / 27.7 ms against the list's 1 : 0.81 : 1.00 : 1.63): a one-variable
// term whose folds take the tile-moment pass costs 0.42 of its list price; the fp32 ratios sit within 25 % of the list
double b_price(void* u, int kind, int dims) {
    const pbn_scoredata* sd = ((Bound*)u)->sd;
    int regions = 1;
    int64_t ntr = 0, nte = 0;
    b_shape(u, kind, &regions, &ntr, &nte);
    double c = term_cost(dims, ntr, nte);
    if (sd->dtype == PBN_F64 && dims == 1 && ntr >= (int64_t)knob_int("PBN_MOMENT_MIN_ROWS", 250000) && knob_int("PBN_MOMENT_PASS", 1) != 0) c *= 0.42;
    return c;
}
int b_batch(void* u, int kind, int n, const int* var, const int* nt, const int* off, const int* par, double* out) {
    Bound* b = (Bound*)u;
    return pbn::score::score_batch_local(b->sd, kind, n, var, nt, off, par, b->params, b->n_params, out);
}
int b_missing(void* u, int kind, int n, const int* off, const int* vars, const int* m, int* missing) { return pbn_score_terms_missing(((Bound*)u)->sd, kind, n, off, vars, m, missing); }
int b_terms(void* u, int kind, int n, const int* off, const int* vars, const int* m, double* out) { return pbn_score_terms(((Bound*)u)->sd, kind, n, off, vars, m, out); }
int b_regions(void* u, int kind, int n, const int* off, const int* vars, const int* m, const int* region, double* out) {
    return pbn_score_term_regions(((Bound*)u)->sd, kind, n, off, vars, m, region, out);
}
int b_put(void* u, int kind, int n, const int* off, const int* vars, const int* m, const double* values) { return pbn_score_terms_put(((Bound*)u)->sd, kind, n, off, vars, m, values); }
int b_parts(void* u, int kind, int n, const int* var, const int* nt, const int* off, const int* par, int part, int n_parts, double* out) {
    return pbn_score_batch_parts(((Bound*)u)->sd, kind, n, var, nt, off, par, part, n_parts, out);
}

}  // namespace shard

namespace score {
// pbn_score_batch on a handle with a communicator (scoring.hip)
void score_batch_sharded(pbn_scoredata* sd, int kind, int n_cand, const int* var, const int* node_type, const int* par_off, const int* parents,
                         const double* params, int n_params, double* out) {
    shard::Bound b{sd, params, n_params};
    pbn_shard_engine eng{&b, sd->n, shard::b_shape, shard::b_batch, shard::b_missing, shard::b_terms, shard::b_regions, shard::b_put, shard::b_parts, shard::b_price};
    shard::run(&eng, &sd->comm, kind, n_cand, var, node_type, par_off, parents, false, out);
}
}  // namespace score
}  // namespace pbn

using namespace pbn;

extern "C" {

int pbn_shard_batch(const pbn_shard_engine* engine, const pbn_comm* comm, int kind, int n_cand, const int* var, const int* node_type,
                    const int* par_off, const int* parents, int shard_all, double* out) {
    // (no lock: the engine's own entry points take theirs, and the host's collective must not run under one)
    try {
        shard::run(engine, comm, kind, n_cand, var, node_type, par_off, parents, shard_all != 0, out);
        return PBN_OK;
    } catch (const invalid_error& e) { set_last_error(e.what()); return PBN_ERR_INVALID;
    } catch (const singular_error& e) { set_last_error(e.what()); return PBN_ERR_SINGULAR;
    } catch (const std::exception& e) { set_last_error(e.what()); return PBN_ERR_DEVICE; }
}

int pbn_shard_deal(int n_items, const double* cost, const uint32_t* tie, int world, double* load, int* owner) {
    return guarded([&] {
        if (n_items < 0 || world < 1 || (n_items > 0 && (!cost || !owner))) throw invalid_error("pbn_shard_deal: bad argument");
        std::vector<int> o;
        shard::deal(std::vector<double>(cost, cost + n_items), tie, world, load, o);
        std::copy(o.begin(), o.end(), owner);
    });
}

double pbn_shard_term_cost(int dims, int64_t train_rows, int64_t test_rows) { return shard::term_cost(dims, train_rows, test_rows); }

int pbn_scoredata_set_comm(pbn_scoredata* sd, const pbn_comm* comm) {
    return guarded(mu_of(sd), [&] {
        if (!sd) throw invalid_error("pbn_scoredata_set_comm: null handle");
        if (comm && (!comm->all_gather || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world)) throw invalid_error("pbn_scoredata_set_comm: bad communicator");
        sd->has_comm = comm != nullptr;
        if (comm) sd->comm = *comm;
    });
}

int pbn_scoredata_reduce_moments(pbn_scoredata* sd, const pbn_comm* comm) {
    try {
        if (!sd || !comm || !comm->all_gather || comm->world < 1) throw invalid_error("pbn_scoredata_reduce_moments: null argument");
        int64_t len = 0;
        shard::must(pbn_scoredata_moments(sd, nullptr, &len, 0), "pbn_scoredata_moments");
        std::vector<double> mine((size_t)len), all((size_t)len * comm->world);
        shard::must(pbn_scoredata_moments(sd, mine.data(), &len, 0), "pbn_scoredata_moments");
        if (comm->all_gather(comm->user, mine.data(), len, all.data()) != 0) throw device_error("pbn_scoredata_reduce_moments: the host's all_gather failed");
        // added in rank order on every rank (exact besides: a segment is non-zero on one rank only)
        std::copy(all.begin(), all.begin() + len, mine.begin());
        for (int r = 1; r < comm->world; ++r)
            for (int64_t i = 0; i < len; ++i) mine[(size_t)i] += all[(size_t)r * len + i];
        shard::must(pbn_scoredata_moments(sd, mine.data(), &len, 1), "pbn_scoredata_moments");
        return PBN_OK;
    } catch (const invalid_error& e) { set_last_error(e.what()); return PBN_ERR_INVALID;
    } catch (const singular_error& e) { set_last_error(e.what()); return PBN_ERR_SINGULAR;
    } catch (const std::exception& e) { set_last_error(e.what()); return PBN_ERR_DEVICE; }
}

int pbn_kde_slogl_sharded(pbn_kde* k, const pbn_table* test, const int* cols, int64_t row0, int64_t n, const pbn_comm* comm, double* out) {
    try {
        if (!out) throw invalid_error("pbn_kde_slogl_sharded: null output");
        if (!comm) { shard::must(pbn_kde_slogl(k, test, cols, row0, n, out), "pbn_kde_slogl"); return PBN_OK; }
        if (!comm->all_gather || comm->world < 1 || comm->rank < 0 || comm->rank >= comm->world) throw invalid_error("pbn_kde_slogl_sharded: bad communicator");
        const int64_t lo = n * comm->rank / comm->world, hi = n * (comm->rank + 1) / comm->world;
        double send[2] = {0.0, 0.0};
        shard::Failure failure;
        if (hi > lo) {
            const int rc = pbn_kde_slogl(k, test, cols, row0 + lo, hi - lo, &send[0]);
            if (rc != PBN_OK) { failure = {rc, pbn_last_error()}; send[0] = std::nan(""); send[1] = 1.0; }
        }
        std::vector<double> recv((size_t)2 * comm->world);
        if (comm->all_gather(comm->user, send, 2, recv.data()) != 0) {
            if (failure.rc != PBN_OK) shard::rethrow(failure);
            throw device_error("pbn_kde_slogl_sharded: the host's all_gather failed");
        }
        if (failure.rc != PBN_OK) shard::rethrow(failure);
        std::string bad;
        double total = 0.0;
        for (int r = 0; r < comm->world; ++r) {
            if (recv[(size_t)2 * r + 1] != 0.0) bad += (bad.empty() ? "" : ", ") + std::to_string(r);
            total += recv[(size_t)2 * r];
        }
        if (!bad.empty()) throw device_error("sharded_slogl: rank(s) [" + bad + "] failed while computing their share of the batch");
        *out = total;
        return PBN_OK;
    } catch (const invalid_error& e) { set_last_error(e.what()); return PBN_ERR_INVALID;
    } catch (const singular_error& e) { set_last_error(e.what()); return PBN_ERR_SINGULAR;
    } catch (const std::exception& e) { set_last_error(e.what()); return PBN_ERR_DEVICE; }
}

}  // extern "C"
