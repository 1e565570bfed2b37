/* Plain-C consumer of include/pbn_hip.h: what a non-Python host (the reference's C++ side, a cgo/JNI stub ...)
 * would do.  Builds a table from host column buffers, takes the covariance on device, selects the normal-reference
 * bandwidth, fits a KDE and a CKDE, evaluates slogl, and scores one BIC candidate.  Prints the results as
 * "name value" lines that tests/test_capi_c_gpu.py compares with the Python classes.
 *   gcc -O2 -Iinclude tests/c/capi_demo.c -Lpybnesian_amd -lpbn_hip -Wl,-rpath,$PWD/pybnesian_amd -lm -o capi_demo */
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include "pbn_hip.h"

#define CHECK(x)                                                           \
    do {                                                                   \
        int rc_ = (x);                                                     \
        if (rc_ != PBN_OK) {                                               \
            fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, pbn_last_error()); \
            return 1;                                                      \
        }                                                                  \
    } while (0)

static uint64_t lcg = 88172645463325252ULL;
static double unif(void) { /* xorshift64: deterministic, reproduced in the Python test */
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) / 9007199254740992.0;
}

int main(void) {
    enum { N = 5000, M = 700, D = 3 };
    static double train[D][N], test[D][M];
    for (int i = 0; i < N; ++i) {
        double a = unif() + unif() + unif() - 1.5, b = unif() + unif() - 1.0, c = unif() - 0.5;
        train[0][i] = a; train[1][i] = 0.6 * a + b; train[2][i] = a - 0.4 * b + c;
    }
    for (int i = 0; i < M; ++i) {
        double a = unif() + unif() + unif() - 1.5, b = unif() + unif() - 1.0, c = unif() - 0.5;
        test[0][i] = a; test[1][i] = 0.6 * a + b; test[2][i] = a - 0.4 * b + c;
    }
    pbn_ctx* ctx;
    CHECK(pbn_ctx_create(0, &ctx));
    const void* tr_cols[D] = {train[0], train[1], train[2]};
    const void* te_cols[D] = {test[0], test[1], test[2]};
    pbn_table *ttrain, *ttest;
    CHECK(pbn_table_create(ctx, tr_cols, D, N, PBN_F64, NULL, 0, &ttrain));
    CHECK(pbn_table_create(ctx, te_cols, D, M, PBN_F64, NULL, 0, &ttest));
    int cols[D] = {0, 1, 2};
    double means[D], sse[D * D], cov[D * D], H[D * D];
    CHECK(pbn_table_sse(ttrain, cols, D, 0, N, means, sse));
    for (int i = 0; i < D * D; ++i) cov[i] = sse[i] / (N - 1);
    CHECK(pbn_bandwidth(PBN_SEL_NORMAL_REFERENCE, PBN_BW_FULL, cov, D, N, PBN_F64, H));
    pbn_kde *kde, *ckde;
    double s;
    CHECK(pbn_kde_fit(ctx, ttrain, cols, D, 0, N, H, PBN_BW_FULL, means, &kde));
    CHECK(pbn_kde_slogl(kde, ttest, cols, 0, M, &s));
    printf("kde_slogl %.17g\n", s);
    CHECK(pbn_ckde_fit(ctx, ttrain, cols, D, 0, N, H, means, &ckde));
    CHECK(pbn_kde_slogl(ckde, ttest, cols, 0, M, &s));
    printf("ckde_slogl %.17g\n", s);
    pbn_scoredata* sd;
    CHECK(pbn_scoredata_create(ctx, ttrain, PBN_SPLIT_NONE, 0, 0, 0.0, &sd));
    int var = 2, parents[2] = {0, 1}, off[2] = {0, 2};
    CHECK(pbn_score_batch(sd, PBN_SCORE_BIC, 1, &var, NULL, off, parents, NULL, 0, &s));
    printf("bic_c_ab %.17g\n", s);
    /* error convention: a column index out of range is PBN_ERR_INVALID with a message */
    int bad[1] = {7};
    int rc = pbn_kde_slogl(kde, ttest, bad, 0, M, &s);
    printf("bad_rc %d\n", rc);
    pbn_scoredata_destroy(sd);
    pbn_kde_destroy(kde);
    pbn_kde_destroy(ckde);
    pbn_table_destroy(ttrain);
    pbn_table_destroy(ttest);
    pbn_ctx_destroy(ctx);
    return 0;
}
