/* A C host of the sharded delta cache: what the reference's C++ side would add to run one process per GPU (INTEGRATION.md "sharded
 * cache_scores").  The host owns the communicator - here RCCL, initialised with ONE rank on device 0 so that the program runs on a
 * one-GPU box (ncclCommInitRank with the job's rank / size in a real launch) - and gives the library ONE function: an all-gather of doubles.
 * Everything else (which rank sweeps which CKDE term or fold, the single collective per batch, the assembly) happens behind
 * pbn_score_batch once pbn_scoredata_set_comm was called.
 * Prints "name value" lines; tests/test_capi_c_gpu.py compares them with the plain one-process calls (bit for bit) and with the oracle.
 *   gcc -O2 -D__HIP_PLATFORM_AMD__ -Iinclude -I/opt/rocm/include tests/c/shard_rccl_demo.c -Lpybnesian_amd -lpbn_hip \
 *       -L/opt/rocm/lib -lrccl -lamdhip64 -Wl,-rpath,$PWD/pybnesian_amd -Wl,-rpath,/opt/rocm/lib -lm -o shard_rccl_demo */
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "pbn_hip.h"

#define CHECK(x)                                                                \
    do {                                                                        \
        int rc_ = (x);                                                          \
        if (rc_ != PBN_OK) {                                                    \
            fprintf(stderr, "%s failed (%d): %s\n", #x, rc_, pbn_last_error()); \
            return 1;                                                           \
        }                                                                       \
    } while (0)
#define HIPOK(x) do { if ((x) != hipSuccess) { fprintf(stderr, "%s failed\n", #x); return 1; } } while (0)

typedef struct {
    ncclComm_t comm;
    hipStream_t stream;
    int world;
    double *send_d, *recv_d;   /* device staging, grown on demand */
    int64_t cap;
    int64_t calls;
} host_comm;

/* pbn_allgather_fn: host buffers in, host buffers out; the exchange itself is ncclAllGather on device memory (xGMI between GPUs) */
static int all_gather(void* user, const double* send, int64_t count, double* recv) {
    host_comm* h = (host_comm*)user;
    if (count > h->cap) {
        if (h->send_d) { hipFree(h->send_d); hipFree(h->recv_d); }
        h->cap = count < 1024 ? 1024 : 2 * count;
        if (hipMalloc((void**)&h->send_d, (size_t)h->cap * sizeof(double)) != hipSuccess) return 1;
        if (hipMalloc((void**)&h->recv_d, (size_t)h->cap * h->world * sizeof(double)) != hipSuccess) return 1;
    }
    if (hipMemcpyAsync(h->send_d, send, (size_t)count * sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess) return 1;
    if (ncclAllGather(h->send_d, h->recv_d, (size_t)count, ncclDouble, h->comm, h->stream) != ncclSuccess) return 1;
    if (hipMemcpyAsync(recv, h->recv_d, (size_t)count * h->world * sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess) return 1;
    if (hipStreamSynchronize(h->stream) != hipSuccess) return 1;
    ++h->calls;
    return 0;
}

static uint64_t lcg = 88172645463325252ULL;
static double unif(void) { /* xorshift64: deterministic, reproduced in the Python test */
    lcg ^= lcg << 13; lcg ^= lcg >> 7; lcg ^= lcg << 17;
    return (double)(lcg >> 11) / 9007199254740992.0;
}

int main(void) {
    enum { N = 3000, D = 4, K = 3 };
    static double data[D][N];
    for (int i = 0; i < N; ++i) {
        double a = unif() + unif() + unif() - 1.5, b = unif() + unif() - 1.0, c = unif() - 0.5, e = unif() + unif() - 1.0;
        data[0][i] = a; data[1][i] = 0.6 * a + b; data[2][i] = a - 0.4 * b + c; data[3][i] = 0.3 * c + e;
    }
    HIPOK(hipSetDevice(0));
    host_comm hc;
    memset(&hc, 0, sizeof hc);
    hc.world = 1;
    int devs[1] = {0};
    if (ncclCommInitAll(&hc.comm, 1, devs) != ncclSuccess) { fprintf(stderr, "ncclCommInitAll failed\n"); return 1; }
    HIPOK(hipStreamCreate(&hc.stream));
    int nranks = 0;
    ncclCommCount(hc.comm, &nranks);
    printf("rccl_ranks %d\n", nranks);
    pbn_comm comm = {0, 1, all_gather, &hc};

    pbn_ctx* ctx;
    CHECK(pbn_ctx_create(0, &ctx));
    const void* cols[D] = {data[0], data[1], data[2], data[3]};
    pbn_table* table;
    CHECK(pbn_table_create(ctx, cols, D, N, PBN_F64, NULL, 0, &table));
    /* a delta-cache batch: x1 | x0 ; x0 | x1 ; x2 | x0, x1 ; x3 | {} ; x3 | x2 as CKDE, x2 | x0 as LinearGaussian */
    enum { NC = 6 };
    int var[NC] = {1, 0, 2, 3, 3, 2}, nt[NC] = {PBN_NODE_CKDE, PBN_NODE_CKDE, PBN_NODE_CKDE, PBN_NODE_CKDE, PBN_NODE_CKDE, PBN_NODE_LG};
    int off[NC + 1] = {0, 1, 2, 4, 4, 5, 6}, par[6] = {0, 1, 0, 1, 2, 0};
    double plain[NC], sharded[NC], again[NC];

    pbn_scoredata* one;   /* the one-process run */
    CHECK(pbn_scoredata_create(ctx, table, PBN_SPLIT_CV, K, 7, 0.0, &one));
    CHECK(pbn_score_batch(one, PBN_SCORE_CVLIK, NC, var, nt, off, par, NULL, 0, plain));

    pbn_scoredata* sd;    /* the job's rank: row-sharded moments, one exchange, then the communicator bound to the handle */
    CHECK(pbn_scoredata_create_sharded(ctx, table, PBN_SPLIT_CV, K, 7, 0.0, comm.rank, comm.world, &sd));
    CHECK(pbn_scoredata_reduce_moments(sd, &comm));
    CHECK(pbn_scoredata_set_comm(sd, &comm));
    CHECK(pbn_score_batch(sd, PBN_SCORE_CVLIK, NC, var, nt, off, par, NULL, 0, sharded));
    const int64_t after_first = hc.calls;
    CHECK(pbn_score_batch(sd, PBN_SCORE_CVLIK, NC, var, nt, off, par, NULL, 0, again));   /* every term known now */
    int same = 1;
    for (int i = 0; i < NC; ++i) {
        same = same && memcmp(&plain[i], &sharded[i], sizeof(double)) == 0 && memcmp(&plain[i], &again[i], sizeof(double)) == 0;
        printf("score_%d %.17g\n", i, sharded[i]);
    }
    printf("bit_identical %d\n", same);
    printf("collectives_first_batch %lld\n", (long long)(after_first - 1));   /* minus the moments' exchange */
    printf("collectives_total %lld\n", (long long)hc.calls);
    int64_t entries = 0, sweeps_one = 0, sweeps_sd = 0;
    CHECK(pbn_scoredata_cache_stats(one, &entries, &sweeps_one));
    CHECK(pbn_scoredata_cache_stats(sd, &entries, &sweeps_sd));
    printf("sweeps_one %lld\nsweeps_sharded %lld\n", (long long)sweeps_one, (long long)sweeps_sd);

    pbn_scoredata_destroy(sd);
    pbn_scoredata_destroy(one);
    pbn_table_destroy(table);
    pbn_ctx_destroy(ctx);
    if (hc.send_d) { hipFree(hc.send_d); hipFree(hc.recv_d); }
    hipStreamDestroy(hc.stream);
    ncclCommDestroy(hc.comm);
    return same ? 0 : 2;
}
