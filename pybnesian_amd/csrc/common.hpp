// Shared host-side plumbing for libpbn_hip (MI355X / gfx950 only).
//
// Error convention (mirrors the reference's exception classes, see
// SURVEY.md §8b "Error conventions"):
//   PBN_OK                 success
//   PBN_ERR_INVALID        std::invalid_argument  -> Python ValueError
//   PBN_ERR_SINGULAR       util::singular_covariance_data -> SingularCovarianceData
//   PBN_ERR_DEVICE         HIP failure            -> RuntimeError
#pragma once
#include <hip/hip_runtime.h>
#include <cstdlib>
#include <utility>

#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <atomic>
#include <mutex>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/pbn_hip.h"

#define PBN_MAX_D_HOST 33  // == PBN_MAX_D of kde_kernels.hpp (32 whitened dims + 1 CKDE coordinate)

namespace pbn {

struct invalid_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct singular_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};
struct device_error : std::runtime_error {
    using std::runtime_error::runtime_error;
};

void set_last_error(const std::string& s);

// Locking of the C ABI.  A context has ONE stream and grow-only scratch arenas that every call re-carves (scratch_q, scratch_part,
// ...), and the Python layer loads the library with ctypes.CDLL, which drops the GIL around calls - the reference holds the GIL
// through its pybind11 calls (SURVEY.md 8b "Threading"), so there two Python threads can never be inside the library at once.
// Here every entry point that works on a context - directly or through a handle that belongs to one (table, KDE, score data, MI /
// kMI engines) - takes THAT CONTEXT's recursive mutex (pbn_ctx::mu, mu_of below): two threads on the same context are serialised as
// before, threads driving different contexts (GPUs) are not.  A hill-climb handle and the stand-alone search entry points
// (pbn_hc_*, pbn_mmpc_*) hold only their own state's lock while they run - their score / independence-test callbacks take the lock
// of whatever context they touch - so a search no longer blocks the rest of the process for its whole duration (round 2: one
// process-wide mutex).  Host-only entry points (bandwidths, split layouts, samplers) and creation / destruction of contexts use the
// process-wide mutex.  All mutexes are recursive: entry points call each other, and Python callbacks re-enter on the same thread.
std::recursive_mutex& api_mutex();
#define PBN_API_LOCK std::lock_guard<std::recursive_mutex> pbn_api_lock_(::pbn::api_mutex())

inline void hip_check(hipError_t e, const char* what, const char* file, int line) {
    if (e != hipSuccess) {
        char buf[512];
        snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
        throw device_error(buf);
    }
}
#define HIP_CHECK(x) ::pbn::hip_check((x), #x, __FILE__, __LINE__)

// Translate C++ exceptions into status codes at the C-ABI boundary; `mu` is the lock of the state the call works on (see above).
template <typename F>
int guarded(std::recursive_mutex& mu, F&& f) noexcept {
    try {
        std::lock_guard<std::recursive_mutex> lock(mu);
        f();
        return PBN_OK;
    } catch (const invalid_error& e) {
        set_last_error(e.what());
        return PBN_ERR_INVALID;
    } catch (const singular_error& e) {
        set_last_error(e.what());
        return PBN_ERR_SINGULAR;
    } catch (const device_error& e) {
        set_last_error(e.what());
        return PBN_ERR_DEVICE;
    } catch (const std::bad_alloc&) {
        set_last_error("out of host memory");
        return PBN_ERR_DEVICE;
    } catch (const std::exception& e) {
        set_last_error(e.what());
        return PBN_ERR_INVALID;
    }
}
template <typename F>
int guarded(F&& f) noexcept {   // host-only entry points: the process-wide mutex
    return guarded(api_mutex(), std::forward<F>(f));
}

// RAII device buffer bound to a device; freed with hipFree.
template <typename T>
struct dev_buf {
    T* p = nullptr;
    size_t n = 0;
    dev_buf() = default;
    explicit dev_buf(size_t count) { alloc(count); }
    dev_buf(const dev_buf&) = delete;
    dev_buf& operator=(const dev_buf&) = delete;
    dev_buf(dev_buf&& o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    dev_buf& operator=(dev_buf&& o) noexcept {
        if (this != &o) {
            release();
            p = o.p; n = o.n; o.p = nullptr; o.n = 0;
        }
        return *this;
    }
    ~dev_buf() { release(); }
    void alloc(size_t count) {
        release();
        n = count;
        if (count) HIP_CHECK(hipMalloc((void**)&p, count * sizeof(T)));
    }
    // grow-only scratch
    void reserve(size_t count) {
        if (count > n) alloc(count);
    }
    void release() {
        if (p) (void)hipFree(p);
        p = nullptr; n = 0;
    }
};

inline size_t dtype_size(int dtype) { return dtype == PBN_F32 ? 4 : 8; }

inline int64_t ceil_div(int64_t a, int64_t b) { return (a + b - 1) / b; }

}  // namespace pbn

// ---- handle definitions (opaque in the C header) ---------------------------------------------

struct pbn_ctx {
    std::recursive_mutex mu;   // every entry point working on this context holds it (pbn::mu_of)
    std::atomic<int> refs{1};  // the creator's reference + one per live handle (pbn::ctx_ptr): see pbn_ctx_destroy
    int device = 0;
    hipStream_t stream = nullptr;
    int num_cus = 256;
    // scratch reused across calls (grow-only)
    pbn::dev_buf<char> scratch_part;
    pbn::dev_buf<char> scratch_q;
    pbn::dev_buf<char> scratch_misc;
    pbn::dev_buf<double> scratch_red;
    pbn::dev_buf<char> scratch_train;  // packed training fragments of the score engine
    pbn::dev_buf<char> scratch_prune;  // pruned sweeps: whitened rows, keys, permutation, tile boxes of the training side
    pbn::dev_buf<char> scratch_pruneq; // ... of the query side
    pbn::dev_buf<char> scratch_sort;   // radix sort temporaries
    pbn::dev_buf<double> scratch_split; // CKDE handles evaluated as two plain sweeps: joint / marginal logl or sums
    pbn::dev_buf<double> scratch_w;     // whitening matrices too large for the kernel arguments (more than 16 variables)
    pbn::dev_buf<char> scratch_group;   // grouped KDE evaluation (kde_group.hip): tables, sorted rows, every unit's packs and partials
    // the score engine's result slots (sums, max-norms) of the batch / candidate being evaluated: NOT part of a lane (every lane's
    // evaluations write into it), grow-only - a hipMalloc + hipFree per hybrid candidate (the free synchronises the device) was 0.2 s of C5
    pbn::dev_buf<double> scratch_sums;
    // host buffers of asynchronous uploads that must outlive the call that enqueued them; dropped by whoever synchronises next
    std::vector<std::vector<char>> staged;
    void drop_staged() { staged.clear(); }
    // Extra issue lanes (score engine): independent evaluations are enqueued round-robin on `stream` and on the parked lanes'
    // streams, each with its own scratch, so that the tail of one sweep - the last workgroups of a pruned sweep run 2-3 ms
    // with the slots emptying - overlaps the next evaluations' sorts, packs and sweeps.  swap_lane(k) exchanges the active
    // resources with parked lane k; every routine keeps using ctx->stream / ctx->scratch_*.
    struct Lane {
        hipStream_t stream = nullptr;
        pbn::dev_buf<char> part, q, misc, train, prune, pruneq, sort, group;
        pbn::dev_buf<double> red, split, w;
        hipEvent_t fence = nullptr;
    };
    static constexpr int MAX_PARKED = 3;
    Lane parked[MAX_PARKED];
    void ensure_lanes(int n_parked) {
        for (int k = 0; k < n_parked && k < MAX_PARKED; ++k)
            if (!parked[k].stream) {
                HIP_CHECK(hipStreamCreateWithFlags(&parked[k].stream, hipStreamNonBlocking));
                HIP_CHECK(hipEventCreateWithFlags(&parked[k].fence, hipEventDisableTiming));
            }
    }
    void swap_lane(int k) {
        Lane& alt = parked[k];
        std::swap(stream, alt.stream);
        std::swap(scratch_part, alt.part); std::swap(scratch_q, alt.q); std::swap(scratch_misc, alt.misc);
        std::swap(scratch_train, alt.train); std::swap(scratch_prune, alt.prune); std::swap(scratch_pruneq, alt.pruneq);
        std::swap(scratch_sort, alt.sort); std::swap(scratch_group, alt.group); std::swap(scratch_red, alt.red); std::swap(scratch_split, alt.split); std::swap(scratch_w, alt.w);
    }
    // the parked lanes wait for everything enqueued so far on the active one (e.g. the zeroing of a result buffer)
    void lanes_wait_for_stream(int n_parked) {
        for (int k = 0; k < n_parked && k < MAX_PARKED; ++k) {
            HIP_CHECK(hipEventRecord(parked[k].fence, stream));
            HIP_CHECK(hipStreamWaitEvent(parked[k].stream, parked[k].fence, 0));
        }
    }
    // the active stream waits for everything enqueued so far on parked lane k (device-side join, no host wait)
    void stream_waits_for_lane(int k) {
        HIP_CHECK(hipEventRecord(parked[k].fence, parked[k].stream));
        HIP_CHECK(hipStreamWaitEvent(stream, parked[k].fence, 0));
    }
    void sync_lanes(int n_parked) {
        for (int k = 0; k < n_parked && k < MAX_PARKED; ++k)
            if (parked[k].stream) HIP_CHECK(hipStreamSynchronize(parked[k].stream));
    }
    // optional per-kernel timing (pbn_ctx_set_profiling): HIP events recorded on `stream` around launches
    bool profiling = false;
    bool timing = false;   // events recorded (KernelTimer), behaviour unchanged: pbn_ctx_set_profiling(ctx, 2)
    struct Timed { hipEvent_t e0, e1; int which; };
    std::vector<Timed> pending;
    double kernel_ms[PBN_NUM_KERNEL_CLASSES] = {0};
    int64_t kernel_launches[PBN_NUM_KERNEL_CLASSES] = {0};
};

namespace pbn {
// Lifetime.  Every handle created on a context (tables, KDE models, score data, MI / kMI engines) holds a counted reference on it:
// pbn_ctx_destroy only drops the creator's reference, and the streams, arenas and the mutex go with the LAST one.  A garbage
// collector finalises the objects of a reference cycle in no particular order (CPython, PEP 442): a handle's destroy call that
// arrives after its context's must still find the mutex it locks - before this count such a call locked freed memory, and a test
// run hung in DeviceTable.__del__ once in a few hundred runs.
void ctx_release(pbn_ctx* c);   // capi.hip
inline void ctx_retain(pbn_ctx* c) { c->refs.fetch_add(1, std::memory_order_relaxed); }
struct ctx_ptr {   // a handle's `ctx` member: reads like the plain pointer, owns one reference
    pbn_ctx* p = nullptr;
    ctx_ptr() = default;
    ctx_ptr(const ctx_ptr&) = delete;
    ctx_ptr& operator=(const ctx_ptr&) = delete;
    ctx_ptr& operator=(pbn_ctx* c) {
        if (c) ctx_retain(c);
        if (p) ctx_release(p);
        p = c;
        return *this;
    }
    ~ctx_ptr() { if (p) ctx_release(p); }
    operator pbn_ctx*() const { return p; }
    pbn_ctx* operator->() const { return p; }
};
// A destroy entry point pins the context BEFORE it takes the context's lock: the handle's own reference goes with `delete h` inside
// the locked region, and the mutex must outlive the unlock.
struct ctx_pin {
    pbn_ctx* c;
    explicit ctx_pin(pbn_ctx* c_) : c(c_) { if (c) ctx_retain(c); }
    ~ctx_pin() { if (c) ctx_release(c); }
    ctx_pin(const ctx_pin&) = delete;
    ctx_pin& operator=(const ctx_pin&) = delete;
};
}  // namespace pbn

namespace pbn {
// the lock an entry point takes: the context's own, through any handle that carries a `ctx` member; a null handle falls back to
// the process-wide mutex (the call then fails on its own null check)
inline std::recursive_mutex& mu_of(const pbn_ctx* c) { return c ? const_cast<pbn_ctx*>(c)->mu : api_mutex(); }
template <typename H>
inline std::recursive_mutex& mu_of(const H* h) { return h ? mu_of(static_cast<const pbn_ctx*>(h->ctx)) : api_mutex(); }
}  // namespace pbn

// ---- run-time switches ---------------------------------------------------------------------------------------------------------------
// knob_*(): the documented environment switches of the product (DESIGN.md 6b holds the one table of them).  PBN_TUNE(NAME, default):
// tuning constants and alternative paths kept from past measurements - compiled to their defaults; only a library built with
// -DPBN_EXPERIMENTS (the probe scripts under tools/: `make EXPERIMENTS=1`) reads PBN_<NAME> from the environment.
namespace pbn {
inline long long knob_ll(const char* name, long long dflt) { const char* e = std::getenv(name); return (e && *e) ? std::atoll(e) : dflt; }
inline int knob_int(const char* name, int dflt) { return (int)knob_ll(name, dflt); }
inline double knob_double(const char* name, double dflt) { const char* e = std::getenv(name); return (e && *e) ? std::atof(e) : dflt; }
}  // namespace pbn
#ifdef PBN_EXPERIMENTS
#define PBN_TUNE(NAME, dflt) (::pbn::knob_int("PBN_" #NAME, (int)(dflt)))
#define PBN_TUNE_D(NAME, dflt) (::pbn::knob_double("PBN_" #NAME, (double)(dflt)))
#else
#define PBN_TUNE(NAME, dflt) ((int)(dflt))
#define PBN_TUNE_D(NAME, dflt) ((double)(dflt))
#endif

namespace pbn {
// PBN_SCORE_LANES (at most 4): issue lanes of the score engine's independent evaluations; 1 keeps everything on the
// context's own stream.  Default 2; 3 for tables of at most 250 000 rows, whose evaluations are chains of short launches
// (sorts, boxes, prepass, a 250 us sweep, finish: `--hc cv64` 7.2 -> 6.3 s with three lanes, 7.2 with four) - on the large
// tables a third lane only takes CUs from the other two's sweeps (C5 +5 %).
inline int score_lanes(int64_t table_rows = -1) {
    static const int v = [] {
        const int n = knob_int("PBN_SCORE_LANES", 0);
        return n < 1 ? 0 : (n > 1 + pbn_ctx::MAX_PARKED ? 1 + pbn_ctx::MAX_PARKED : n);
    }();
    if (v) return v;
    return (table_rows >= 0 && table_rows <= 250000) ? 3 : 2;
}
// RAII: the enclosed enqueues go to lane `lane` (0 = the context's own, k > 0 = parked lane k - 1); the active lane is restored on
// scope exit, also by a throw
struct LaneSwitch {
    pbn_ctx* ctx; int k;
    LaneSwitch(pbn_ctx* c, int lane) : ctx(c), k(lane - 1) { if (k >= 0) ctx->swap_lane(k); }
    ~LaneSwitch() { if (k >= 0) ctx->swap_lane(k); }
    LaneSwitch(const LaneSwitch&) = delete;
    LaneSwitch& operator=(const LaneSwitch&) = delete;
};
// RAII: brackets the launches issued during its lifetime with two events when profiling is on.
struct KernelTimer {
    pbn_ctx* ctx; int which; hipEvent_t e0 = nullptr;
    KernelTimer(pbn_ctx* c, int w) : ctx(c), which(w) {
        // timing-only mode brackets the dominant classes only (sweeps, Gram passes: long kernels) - events around the short pack / finish
        // launches of a chain cost more than they measure (cv64's weak leg 0.61 -> 0.91 s with every launch bracketed)
        if (ctx->profiling || (ctx->timing && (w == PBN_K_SWEEP || w == PBN_K_GRAM || w == PBN_K_MOMENT))) { HIP_CHECK(hipEventCreate(&e0)); HIP_CHECK(hipEventRecord(e0, ctx->stream)); }
    }
    ~KernelTimer() {
        if (e0) {
            hipEvent_t e1;
            if (hipEventCreate(&e1) == hipSuccess && hipEventRecord(e1, ctx->stream) == hipSuccess)
                ctx->pending.push_back({e0, e1, which});
        }
    }
};
}  // namespace pbn

// Column-major device table: column c lives at base + c*ld elements.
struct pbn_table {
    pbn::ctx_ptr ctx;
    int dtype = PBN_F64;
    int n_cols = 0;
    int64_t n_rows = 0;
    int64_t ld = 0;          // elements between consecutive columns
    void* data = nullptr;    // device
    bool owns = false;
    const void* col(int c) const { return (const char*)data + (size_t)c * ld * pbn::dtype_size(dtype); }
};
