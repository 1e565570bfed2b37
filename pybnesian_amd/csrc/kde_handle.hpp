// The public fitted-KDE handle (opaque `pbn_kde` of include/pbn_hip.h), shared by capi.hip and sampling.hip.
#pragma once
#include <memory>

#include "kde_kernels.hpp"
#include "kde_model.hpp"

using pbn::KdeModel;
using pbn::PackArgs;
using pbn::dev_buf;

struct pbn_kde {
    pbn::ctx_ptr ctx;
    KdeModel m;
    dev_buf<char> Apack, nxpack, Axpack;
    dev_buf<char> prune_store;          // tile boxes | Morton-ordered whitened rows | keys (when m.prune)
    // CKDE::cdf state (pbn_ckde_fit only): classic fragments of the evidence dimensions + u = (x - b.e)/(sigma_c sqrt 2)
    bool ckde = false;
    int cdf_KS = 0;
    bool cdf_wide = false;              // more than 16 evidence variables: fp64 fragments (whatever the table's type), runtime-sized kernels
    std::vector<int> cols_fit;          // caller's column order (variable first)
    const pbn_table* train = nullptr;   // borrowed: CKDE::sample reads the sampled training rows from it
    int64_t train_row0 = 0;
    std::vector<double> wu;
    dev_buf<char> cA, cN, cU;
    // Low-dimensional CKDE handles (where tile pruning pays, see kde_fit_impl): logl = logl_joint - logl_marginal from two
    // pruned PLAIN sweeps - the reference's own formulation (factors/continuous/CKDE.hpp:256-287) - instead of the fused
    // sweep, which cannot prune on the joint box.  `m` then only carries the host-side whitening / normalisation data.
    std::unique_ptr<pbn_kde> split_joint, split_marg;
};

// WidePackArgs of the cdf fragments of a handle with more than 16 evidence variables: whitening order (evidence first, variable last),
// contraction over the evidence only, u from all d columns (kde_model.hpp: kde_wide_pack_args)
inline void fill_cdf_pack_wide(pbn_ctx* ctx, pbn::WidePackArgs& wa, const pbn_kde& k, const pbn_table* t, const int* cols) {
    const KdeModel& m = k.m;
    std::vector<int> hc((size_t)m.d);
    for (int i = 0; i < m.d; ++i) hc[i] = cols[m.perm[i]];
    pbn::kde_wide_pack_args(ctx, wa, t, hc.data(), m.d, m.d - 1, m.W.data(), m.d, m.mu.data(), k.wu.data());
}

// PackArgs for the cdf fragments: whitening order (evidence first, variable last), contraction over the evidence only.
inline void fill_cdf_pack(PackArgs& pa, const pbn_kde& k, const pbn_table* t, const int* cols) {
    const KdeModel& m = k.m;
    if (m.d > PBN_W_INLINE_D) throw pbn::invalid_error("internal: a wide CKDE reached the fixed-size cdf pack");
    pa.Wdev = nullptr;
    pa.base = t->data; pa.ld = t->ld; pa.d = m.d; pa.dm = m.d - 1; pa.KS = k.cdf_KS;
    for (int i = 0; i < m.d; ++i) pa.cols[i] = cols[m.perm[i]];
    for (int i = 0; i < m.d * m.d; ++i) pa.W[i] = m.W[i];
    for (int i = 0; i < m.d; ++i) { pa.mu[i] = m.mu[i]; pa.wu[i] = k.wu[i]; }
}

