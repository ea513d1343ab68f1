// csrc/pcu_hip.hip -- C ABI (include/pcu_hip.h) + host orchestration of the gfx950 kernels.
//
// One of the library's two translation units (the other, search_kernels.hip, holds the k > 1 search kernels' instantiations; recipe:
// __graft_entry__.py), each built with:  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fPIC -c
// (-ffp-contract=off is part of the numerical contract: see search.h). No PyTorch, no Python, no fallback:
// every entry point either runs the HIP path or returns an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <stdlib.h>
#include <stddef.h>
#include <algorithm>
#include <limits>
#include <string>
#include <vector>
#include <chrono>
#include <atomic>
#include <signal.h>
#include <time.h>

#include "../../include/pcu_hip.h"
#include "grid.h"
#include "grid2.h"
#include "reduce.h"
#include "search.h"
#include "search_brick.h"
namespace pcu {          // the k > 1 search kernels are compiled in search_kernels.hip (second translation unit, built in parallel)
#define PCU_SEARCH_INST extern template
#include "search_inst.h"
#undef PCU_SEARCH_INST
}
#include "kd_order.h"
#include "normals.h"
#include "morton.h"
#include "voxel.h"
#include "sinkhorn.h"

using namespace pcu;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static thread_local int g_fail_code = 0;       // code of this thread's last fail(): internal layers pass "failed" up as -1 (their positive codes mean other
                                               // things), the ABI boundary returns the code the failure was recorded with (abi_rc)
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    g_fail_code = code;
    return code;
}
static int abi_rc(int rc);
#define HIP_TRY(x)                                                                              \
    do { hipError_t e_ = (x); if (e_ != hipSuccess)                                             \
        return fail(PCU_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// ------------------------------------------------------------------------------------------------ cancellation
// The reference polls PyErr_CheckSignals() per query and per kd-tree node and turns Ctrl-C into KeyboardInterrupt
// (src/point_cloud_distance.cpp:60-75, 96-98; external/nanoflann/nanoflann.hpp:1004). Here a call is a sequence of host phases (enqueue, wait,
// decide, enqueue ...), and every host wait is a bounded poll that also looks at one process-wide request counter:
//   pcu_hip_cancel()            bumps it (any thread, async-signal-safe);
//   pcu_hip_watch_sigint(1)     chains a SIGINT handler in front of the installed one (Python's): it bumps the counter and then calls the previous
//                               handler, so the interpreter still records the signal for its own handler -- which decides (the Python side asks
//                               it through PyErr_CheckSignals, as the reference does: _lib.py: _after_call).
// A compute entry point notes the counter on entry (CallGuard); a request made while the call is in flight makes the two differ: the call stops
// enqueuing, drains ITS streams (kernels cannot be killed; what is queued is one phase: milliseconds), and returns PCU_HIP_ERR_CANCELLED. A
// request is for the calls in flight: one made between two calls is not remembered (after a SIGINT the interpreter raises between the calls
// anyway). Entry points that do no work (context / index create and destroy) neither note nor clear anything, so a lane created in the middle
// of a batch call, or a context destroyed by another thread or the garbage collector, cannot swallow a request. Contexts reset their cross-call
// device state (the "no memset" fill words, the speculative tree top) when they see that a call was abandoned since their last one
// (g_cancel_epoch).
static std::atomic<int> g_cancel_source{0};            // who made the last one: PCU_HIP_CANCEL_BY_REQUEST / PCU_HIP_CANCEL_BY_SIGINT
static std::atomic<unsigned> g_cancel_gen{0};          // number of requests so far
// ... and its copy in pinned, device-visible host memory (allocated with the first context): the kernels that can run for tens or hundreds of
// milliseconds -- the tie-order / big-k traversals, the wave-per-query pass -- look at it every ~1000 steps and return when it has moved past
// their call's value (pcu_types.h: cancel_seen), so that ONE long call is abandoned within milliseconds, not at its next host phase. The copy is
// written after the counter, and a kernel only reacts to "newer than my call": whenever a kernel has bailed out, the host's own test is true.
static std::atomic<unsigned*> g_cancel_mirror{nullptr};
static inline void cancel_request(int source) {
    g_cancel_source.store(source, std::memory_order_relaxed);
    const unsigned g = g_cancel_gen.fetch_add(1, std::memory_order_acq_rel) + 1u;
    if (unsigned* m = g_cancel_mirror.load(std::memory_order_acquire)) __atomic_store_n(m, g, __ATOMIC_RELEASE);
}
static thread_local unsigned t_call_gen = 0;           // g_cancel_gen when this thread's current call began
static std::atomic<unsigned> g_cancel_epoch{0};
static struct sigaction g_prev_sigint;
static std::atomic<int> g_sigint_watched{0};
static std::atomic<int> g_in_sigint{0};
static void pcu_on_sigint(int sig, siginfo_t* info, void* uc) {
    // (a handler installed over ours that chains back to ours, and over which ours was re-armed, would recurse: the inner visit returns at once)
    if (g_in_sigint.exchange(1)) return;
    const struct sigaction prev = g_prev_sigint;
    if ((prev.sa_flags & SA_SIGINFO) || prev.sa_handler != SIG_IGN) {                                 // the host ignores SIGINT: so do the calls
        cancel_request(PCU_HIP_CANCEL_BY_SIGINT);
        if (prev.sa_flags & SA_SIGINFO) { if (prev.sa_sigaction && prev.sa_sigaction != pcu_on_sigint) prev.sa_sigaction(sig, info, uc); }
        else if (prev.sa_handler != SIG_DFL && prev.sa_handler) prev.sa_handler(sig);
        else if (prev.sa_handler == SIG_DFL) { g_in_sigint.store(0); signal(SIGINT, SIG_DFL); raise(SIGINT); return; }   // no handler before ours: the default action
    }
    g_in_sigint.store(0);
}
static int sigint_install() {
    struct sigaction sa; memset(&sa, 0, sizeof sa);
    sa.sa_sigaction = pcu_on_sigint; sa.sa_flags = SA_SIGINFO | SA_RESTART; sigemptyset(&sa.sa_mask);
    struct sigaction old;
    if (sigaction(SIGINT, &sa, &old) != 0) return -1;
    g_prev_sigint = old;
    return 0;
}
// Somebody replaced the handler after pcu_hip_watch_sigint(1) (Python's signal.signal() does: ipykernel, a graceful-shutdown hook): chain in
// front of the new one, or Ctrl-C would silently stop reaching the calls. One sigaction() query per compute call (~0.2 us).
static void sigint_rearm() {
    if (!g_sigint_watched.load(std::memory_order_relaxed)) return;
    struct sigaction cur;
    if (sigaction(SIGINT, nullptr, &cur) != 0) return;
    if ((cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == pcu_on_sigint) return;
    (void)sigint_install();
}
__attribute__((destructor)) static void sigint_unchain_at_unload() {       // (dlclose / exit: never leave a handler behind that points into an unmapped library)
    struct sigaction cur;
    if (g_sigint_watched.exchange(0) && sigaction(SIGINT, nullptr, &cur) == 0 && (cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == pcu_on_sigint)
        (void)sigaction(SIGINT, &g_prev_sigint, nullptr);
}
struct pcu_hip_ctx;
static thread_local pcu_hip_ctx* t_call_ctx = nullptr;          // the context of this thread's current call (drain_call)
static void drain_call(hipStream_t s);
static int cancelled(hipStream_t s) {
    drain_call(s);                                 // nothing of the abandoned call may still be running when its buffers are reused or freed
    g_cancel_epoch.fetch_add(1, std::memory_order_relaxed);
    return fail(PCU_HIP_ERR_CANCELLED, "cancelled (pcu_hip_cancel / SIGINT) while waiting for the GPU");
}
static inline bool cancel_requested() { return g_cancel_gen.load(std::memory_order_relaxed) != t_call_gen; }
// What a compute entry point returns: the code its failure was recorded with; and a call that ran to its end while a request was pending --
// its last kernels may have bailed out -- is an abandoned call, whatever its host phases saw.
static int abi_rc(int rc) {
    if (rc < 0) return g_fail_code < 0 ? g_fail_code : rc;
    if (rc == 0 && cancel_requested()) return cancelled(nullptr);
    return rc;
}
// hipStreamSynchronize as a bounded poll: tight for the first 2 ms (short calls keep their latency), then 50 us naps.
static int wait_stream(hipStream_t s) {
    const auto t0 = std::chrono::steady_clock::now();
    bool nap = false;
    for (unsigned it = 0;; ++it) {
        const hipError_t e = hipStreamQuery(s);
        if (e == hipSuccess) return cancel_requested() ? cancelled(s) : 0;       // (a kernel may have bailed out on the request: its results do not count)
        if (e != hipErrorNotReady) return fail(PCU_HIP_ERR_RUNTIME, "hipStreamQuery failed: %s", hipGetErrorString(e));
        if (cancel_requested()) return cancelled(s);
        if (!nap && (it & 0x3f) == 0x3f && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(2)) nap = true;
        if (nap) { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
    }
}
#define HIP_WAIT(s) do { const int w_ = wait_stream(s); if (w_) return w_; } while (0)
// The same for an event recorded on stream s (long launches are enqueued in pieces with at most two in the queue: see kd_search_launch).
static int wait_event(hipEvent_t ev, hipStream_t s) {
    for (unsigned it = 0;; ++it) {
        const hipError_t e = hipEventQuery(ev);
        if (e == hipSuccess) return cancel_requested() ? cancelled(s) : 0;
        if (e != hipErrorNotReady) return fail(PCU_HIP_ERR_RUNTIME, "hipEventQuery failed: %s", hipGetErrorString(e));
        if (cancel_requested()) return cancelled(s);
        if (it > 64) { struct timespec ts = {0, 50000}; nanosleep(&ts, nullptr); }
    }
}

// Host-side profile of a call (PCU_HIP_HOST_PROF=1; diagnostics): wall-clock marks at entry, first launch, last launch, result seen, exit;
// the mean spans of every 1000 calls go to stderr. Tells the Python wrapper's share of a step from the library's (scratch/hostgap.py).
struct HostProf {
    bool on = getenv("PCU_HIP_HOST_PROF") != nullptr;
    std::chrono::steady_clock::time_point t[6]; double acc[6] = {0, 0, 0, 0, 0, 0}; long n = 0; bool have_prev = false;
    void mark(int i) { if (on) t[i] = std::chrono::steady_clock::now(); }
    void done() {
        if (!on) return;
        auto us = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
        if (have_prev) acc[0] += us(t[5], t[0]);          // previous exit -> this entry: the caller (Python wrapper, loop)
        for (int i = 1; i < 5; ++i) acc[i] += us(t[i - 1], t[i]);
        t[5] = t[4]; have_prev = true;
        if (++n % 1000 == 0) {
            fprintf(stderr, "[host prof] per call, us: caller %.2f | entry->first launch %.2f | enqueue %.2f | wait for result %.2f | exit %.2f\n",
                    acc[0] / 999.0 * (999.0 / 1000.0), acc[1] / 1000, acc[2] / 1000, acc[3] / 1000, acc[4] / 1000);
            for (double& a : acc) a = 0;
        }
    }
};
static HostProf g_hprof;

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) holds for the CURRENT device only: a process with contexts on several GPUs has to opt every
// kernel in on each of them. One bit per device and call site; true = this (site, device) has not been set yet.
static bool attr_unset_here(std::atomic<unsigned long long>& mask) {
    int d = 0; (void)hipGetDevice(&d);
    const unsigned long long bit = 1ull << (d & 63);
    return !(mask.fetch_or(bit) & bit);
}
// ------------------------------------------------------------------------------------------------ context
struct pcu_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t spec_stream = nullptr;         // the speculative tie-order tree top (see pcu_hip_ctx_create)
    hipStream_t aux_stream = nullptr;          // second lane for two-sided ops: the two clouds' index builds and the two
                                               // search directions are independent and latency-bound, so they overlap
    hipEvent_t jev[4] = {};                    // fork/join events between the caller's stream and aux_stream
    char* arena = nullptr; size_t arena_cap = 0, arena_off = 0;
    std::vector<void*> extra;                 // overflow allocations of the current call
    size_t extra_bytes = 0;
    size_t extra_hint = 0;                    // largest overflow seen: added to the arena request of later calls
    hipEvent_t ev[8] = {};
    hipEvent_t kev[8] = {};                    // brackets of the main (pass-0) search launches of a call
    hipEvent_t cev[2] = {};                    // progress marks of a launch that is enqueued in pieces (kd_search_launch; created on first use)
    int n_kev = 0;
    double occupancy = 0;                      // <=0: default
    int* h_pinned = nullptr;                   // small pinned readback buffer
    unsigned seq = 0;                                        // sequence number of the result block the epilogue kernel writes to h_pinned
    bool time_phases = false, time_kernels = false;          // HIP-event timing of this call (flags PCU_HIP_TIME_*): each event is a
                                                             // ~6 us bubble in the kernel pipeline, so both are opt-in
    // tie-order resolver: its own grow-only workspace (stable addresses across calls, so the captured
    // level-pair graph below stays valid) and one cached executable graph per scalar type
    char* kd_ws = nullptr; size_t kd_ws_cap = 0, kd_ws_off = 0;
    struct KdGraph { hipGraphExec_t exec = nullptr; hipGraph_t graph = nullptr; const void* key_ptr = nullptr; long long key_m = 0; int key_leaf = 0; } kd_graph[4];   // [type][with second planeSplit loop]
    bool kd_need_ph2 = false;                 // sticky: this context has seen data with elements equal to a cut value
    // Speculative tree top (kd_build_device, mode 1): a call whose predecessor needed the tie-order resolver starts the top levels of
    // the tree -- which do not depend on which queries are tied -- on spec_stream while the searches run; the resolver adopts them.
    struct KdSpec {
        bool active = false;                  // a prefix for (pts, gp, m, leaf, with_ph2) is in flight or done and not yet adopted
        bool pending = false;                 // ev_done recorded and not yet waited for by a later user of the workspace
        const void* pts = nullptr; const void* gp = nullptr; int m = 0, leaf = 0; bool with_ph2 = false; int levels_done = 0;
        hipEvent_t ev_fork = nullptr, ev_init = nullptr, ev_done = nullptr;
    } kd_spec;
    bool kd_spec_hint = false;                // sticky: the last large k_nearest_neighbors call of this context had genuine ties
    // batch entry points: independent pairs are kept in flight on `lanes` (full contexts of their own: stream, workspace, pinned
    // result block), created on first use
    unsigned* tickets = nullptr;              // device words, zero between launches ("last block" tickets; no kernel takes one since round 3)
    double occ_scale[2] = {1.0, 1.0};         // sticky: grid resolution of the first / second cloud of a call relative to the default (rescale_wanted:
                                              // surface-like clouds want finer cells); a k_nearest_neighbors dataset counts as the second cloud
    bool two_pass = false;                    // sticky: a one-pass index build of this context overflowed a bucket slot (grid.h: k_bucket_onepass)
    bool eager_large = false;                 // sticky: this context has met clouds with over-full buckets (surfaces, clusters): launch their
                                              // placement (k_bucket_large) with every build instead of on demand
    unsigned long long* fill2 = nullptr; int fill_parity = 0;   // grid2.h "no memset": two sets of kFillWords bucket fill words; a one-pass build uses set
                                                                 // fill_parity -- left zeroed by its predecessor -- and zeroes the other one for its successor
    char* aux = nullptr; size_t aux_cap = 0;  // grow-only block for operators that run a search as a sub-step (normals): survives the
                                              // sub-call's use of the arena
    // grid2.h: GridGeo -- the layout of this context's last two-sided fused build, per cloud, on the device; what it was computed for, on the host
    struct GeoCache { char* dev = nullptr; bool valid[2] = {false, false}; int n[2] = {0, 0}; double occ = 0, h_want = 0; int max_cells = 0, n_layout = 0, tsize = 0; bool shared = false; } geo;
    bool brick_off = false;                   // sticky: a staged k = 1 pass of this context (search_brick.h) fell back to global scans in more than a quarter of its
                                              // blocks (surfaces, clusters: short uneven rows): its fused calls take k_search1_flat on per-cloud grids again
    unsigned cancel_epoch = 0;                // g_cancel_epoch at this context's last call (ctx_begin: reset of the cross-call device state after an abandoned call)
    std::vector<pcu_hip_ctx*> lanes; int n_lanes_wanted = 4;   // (262k-point pairs, round 4, us per pair at 1 / 2 / 3 / 4 / 5 / 6 / 8 lanes: 83 / 51 / 43 / 41 / 47 / 44 / 41 -- scratch/lanes.py; the host is the limit from 3 on)
    hipEvent_t batch_ev = nullptr;
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Every entry point works on its context's device and leaves the calling thread's current HIP device as it found it
// (a torch user who did torch.cuda.set_device(1) and then calls with numpy arrays -- device 0 -- must not find later torch
// allocations on GPU 0).
struct DeviceGuard {
    int prev = -1, dev = -1;
    explicit DeviceGuard(int device) : dev(device) {
        if (dev < 0) return;
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        if (prev != dev) (void)hipSetDevice(dev);
    }
    ~DeviceGuard() { if (dev >= 0 && prev >= 0 && prev != dev) (void)hipSetDevice(prev); }
    DeviceGuard(const DeviceGuard&) = delete; DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// Compute entry points: the device guard + the call's view of the cancellation counter, a clean failure code, the SIGINT chain re-armed.
struct CallGuard : DeviceGuard {
    explicit CallGuard(pcu_hip_ctx* c) : DeviceGuard(c ? c->device : -1) {
        t_call_ctx = c;
        t_call_gen = g_cancel_gen.load(std::memory_order_relaxed);
        g_fail_code = 0;
        sigint_rearm();
    }
    ~CallGuard() { t_call_ctx = nullptr; }
};
// An abandoned call waits for what it enqueued: the stream it was given, its context's own / auxiliary / speculative streams and its lanes'
// -- not for the whole device (other threads' and the host application's streams keep running).
static void drain_ctx(pcu_hip_ctx* c) {
    for (hipStream_t q : {c->own_stream, c->aux_stream, c->spec_stream}) if (q) (void)hipStreamSynchronize(q);
    for (pcu_hip_ctx* l : c->lanes) drain_ctx(l);
}
static void drain_call(hipStream_t s) {
    if (s) (void)hipStreamSynchronize(s);
    if (t_call_ctx) drain_ctx(t_call_ctx); else (void)hipDeviceSynchronize();
}

struct Arena {
    pcu_hip_ctx* c;
    // Bump allocation out of the context arena; falls back to a tracked hipMalloc when the arena is full
    // (only happens for lazily built coarse grids).
    int alloc(void** out, size_t bytes) {
        bytes = align_up(bytes ? bytes : 16, 256);
        if (c->arena_off + bytes <= c->arena_cap) { *out = c->arena + c->arena_off; c->arena_off += bytes; return 0; }
        void* p = nullptr;
        HIP_TRY(hipMalloc(&p, bytes));
        c->extra.push_back(p); c->extra_bytes += bytes;
        *out = p;
        return 0;
    }
};
template <typename U> static int aalloc(Arena& a, U** out, size_t count) { return a.alloc((void**)out, count * sizeof(U)); }

// Bump allocator over the context's kd workspace (grown, never shrunk; growth invalidates the cached graphs).
struct KdArena {
    pcu_hip_ctx* c;
    template <typename U> int get(U** out, size_t count) {
        size_t bytes = align_up(count * sizeof(U) ? count * sizeof(U) : 16, 256);
        if (c->kd_ws_off + bytes > c->kd_ws_cap) return fail(PCU_HIP_ERR_RUNTIME, "internal: kd workspace overflow");
        *out = reinterpret_cast<U*>(c->kd_ws + c->kd_ws_off); c->kd_ws_off += bytes; return 0;
    }
};
static void kd_graph_drop(pcu_hip_ctx* c) {
    for (auto& g : c->kd_graph) {
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
        if (g.graph) (void)hipGraphDestroy(g.graph);
        g = pcu_hip_ctx::KdGraph();
    }
}
static int kd_ws_reserve(pcu_hip_ctx* c, size_t bytes) {
    if (bytes > c->kd_ws_cap) {
        kd_graph_drop(c);
        if (c->kd_ws) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->kd_ws)); c->kd_ws = nullptr; c->kd_ws_cap = 0; }
        size_t cap = align_up(bytes + (bytes >> 3), 1 << 20);
        HIP_TRY(hipMalloc((void**)&c->kd_ws, cap));
        c->kd_ws_cap = cap;
    }
    c->kd_ws_off = 0;
    return 0;
}

static void ctx_end(pcu_hip_ctx* c);
static int ctx_begin(pcu_hip_ctx* c, size_t want_bytes) {
    ctx_end(c);                                 // drop overflow blocks left by a call that failed midway
    // A call of this process was abandoned (cancelled()) since this context's last one: whatever device state one call leaves for the next
    // may be half-made -- the "no memset" fill words of the one-pass build (grid2.h), a speculative tree top in flight.
    const unsigned ep = g_cancel_epoch.load(std::memory_order_relaxed);
    if (ep != c->cancel_epoch) {
        HIP_TRY(hipDeviceSynchronize());
        if (c->fill2) HIP_TRY(hipMemset(c->fill2, 0, 2 * (size_t)kFillWords * sizeof(unsigned long long)));
        c->fill_parity = 0;
        c->kd_spec.active = c->kd_spec.pending = false;
        c->geo.valid[0] = c->geo.valid[1] = false;
        c->cancel_epoch = ep;
    }
    want_bytes += c->extra_hint;                // what earlier calls had to hipMalloc on top of their estimate (refitted / coarse grids)
    if (want_bytes > c->arena_cap) {
        if (c->arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->arena)); c->arena = nullptr; c->arena_cap = 0; }
        size_t cap = align_up(want_bytes + (want_bytes >> 3), 1 << 20);
        HIP_TRY(hipMalloc((void**)&c->arena, cap));
        c->arena_cap = cap;
    }
    c->arena_off = 0;
    c->n_kev = 0;
    // debugging: poison the workspace (PCU_HIP_DEBUG_POISON=<byte>) so that reads of memory no kernel of this call wrote show up
    static const int poison = getenv("PCU_HIP_DEBUG_POISON") ? atoi(getenv("PCU_HIP_DEBUG_POISON")) : -1;
    if (poison >= 0 && c->arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipMemset(c->arena, poison, c->arena_cap)); HIP_TRY(hipDeviceSynchronize()); }
    return 0;
}
// Sum of the bracketed main-search kernel durations of this call (valid after the final stream sync).
static void collect_kernel_times(pcu_hip_ctx* c, pcu_hip_stats* st) {
    if (!st) return;
    st->ms_kernel_search = 0; st->n_kernel_search = 0;
    for (int i = 0; i + 1 < c->n_kev; i += 2) {
        float ms = 0;
        if (hipEventElapsedTime(&ms, c->kev[i], c->kev[i + 1]) == hipSuccess) { st->ms_kernel_search += ms; st->n_kernel_search++; }
    }
}
static void ctx_end(pcu_hip_ctx* c) {
    // overflow blocks cost a hipMalloc + hipFree (device-synchronising, ~ms) per call: remember how much was needed so that
    // the next call's arena holds it (tight-cluster Chamfer 9.5 -> 4.9 ms once its refit grids stopped overflowing)
    if (c->extra_bytes > c->extra_hint) c->extra_hint = c->extra_bytes + (c->extra_bytes >> 2);
    for (void* p : c->extra) (void)hipFree(p);
    c->extra.clear(); c->extra_bytes = 0;
}

// ------------------------------------------------------------------------------------------------ grid index
template <typename T>
struct GridIndex {
    GridParams<T>* gp = nullptr;
    unsigned* cell_start = nullptr;       // counts, scanned in place
    Pt4<T>* sorted = nullptr;
    unsigned* cell_of = nullptr; unsigned* rank = nullptr; unsigned* block_sums = nullptr;
    T* bbox_partial = nullptr;
    int n = 0, max_cells = 0, scan_blocks = 0;
    unsigned* pos_of = nullptr;           // row -> slot in `sorted` (only when asked for: k_unpermute)
    // bucketed build (grid.h): cells per bucket = 1 << shift; nb_max = host bound on the number of buckets
    bool bucketed = false; int shift = 0, nb_max = 0, n_zero = 0;
    double h_want = 0.0;                  // > 0: cells at least this large (fixed-radius searches)
    Pt4<T>* tmp = nullptr; unsigned *bucket_total = nullptr, *bucket_start = nullptr, *block_base = nullptr, *large_list = nullptr, *n_large = nullptr;
    bool one_pass = false;                // build with k_bucket_onepass (tmp holds nb_max slots of kLargeBucket records); cleared after an overflow
    T* xpartial = nullptr;                // one-pass build, second form (grid2.h): the scatter blocks' bbox partials
    bool lean = false;                    // the Pt4 records of `sorted` are not written (grid2.h: fused k = 1 calls read the coordinate + row-id streams only);
                                          // make_pt4() fills them in when some other kernel needs them
    const T* src = nullptr; double occ_built = 0.0;       // what the index was built from (rebuild after an overflow)
    bool shared_grid = false;             // asked for: this cloud and its partner of a two-sided call are laid over ONE grid (grid2.h: Build2Side::spts1); allocated for
                                          // the larger cloud's plan. Cleared by a build that did not take the second-form one-pass path
};

static int max_cells_for(int64_t n, double occ) {
    double c = (double)n / (occ > 0 ? occ : 1.0) * 1.25 + 64.0;
    if (c > 64.0 * 1024 * 1024) c = 64.0 * 1024 * 1024;
    return (int)c;
}
// Bucketed build: applicable when the cells split into <= kBkMaxBuckets buckets of <= 4096 cells (~kBucketPts = 4096 expected points
// each) and the (block, bucket) reservation table stays small; otherwise (tiny or huge clouds, very coarse grids) the atomic build.
// Workspace of the one-pass variant: every bucket owns a fixed slot of kLargeBucket = 8192 records in `tmp`, i.e. max(n, buckets x 8192)
// records -- about 4x the cloud at 1M points (67 MB instead of 16), bounded by kBkMaxBuckets x 8192 records (2 GiB for float64) beyond
// which it grows with n like everything else; index_bytes() counts exactly what index_alloc() takes.
static bool bucket_plan(int64_t n, double occ, int* shift, int* nb_max) {
    static const bool off = [] { const char* e = getenv("PCU_HIP_INDEX"); return e && strcmp(e, "atomic") == 0; }();
    static const int64_t n_min = getenv("PCU_HIP_BUCKET_MIN") ? atoll(getenv("PCU_HIP_BUCKET_MIN")) : 64;         // below: the atomic build (round 4: 32768; see wave_only_below)
    if (off || n < n_min || occ > 64.0) return false;
    const int mc = max_cells_for(n, occ);
    int sh = 5;
    while (sh < 12 && (double)(2 << sh) * occ <= (double)kBucketPts) ++sh;             // largest bucket with <= ~kBucketPts expected points
    while (sh < 12 && ((mc >> sh) + 1) > kBkMaxBuckets) ++sh;
    const int nb = (mc >> sh) + 1;
    if (nb > kBkMaxBuckets) return false;
    if ((double)(1 << sh) * occ > 0.5 * (double)kLargeBucket) return false;
    const int64_t blocks = (n + kBkBlockPts - 1) / kBkBlockPts;
    if (blocks * (int64_t)nb > 32ll * 1024 * 1024) return false;      // (block, bucket) reservation table: at most 128 MB
    *shift = sh; *nb_max = nb;
    return true;
}
template <typename T>
static size_t index_bytes(int64_t n, double occ) {
    int mc = max_cells_for(n, occ);
    size_t b = align_up(sizeof(GridParams<T>), 256) + align_up((size_t)(mc + 1 + kBkMaxBuckets + 8 + 64) * 4, 256) + align_up(sorted_records_bytes((size_t)n, sizeof(Pt4<T>), sizeof(T)), 256) +
               2 * align_up((size_t)n * 4, 256) + align_up((size_t)(mc / kScanChunk + 2) * 4, 256) + align_up(kBboxBlocks * kBboxStride * sizeof(T), 256);
    int sh = 0, nb = 0;
    if (bucket_plan(n, occ, &sh, &nb))
        b += align_up(std::max((size_t)n, (size_t)nb * kLargeBucket) * sizeof(Pt4<T>), 256) + 2 * align_up((size_t)(nb + 1) * 4, 256) +
             align_up((size_t)((n + kBkBlockPts - 1) / kBkBlockPts) * nb * 4, 256) +
             align_up((size_t)((n + 2047) / 2048) * kXPartStride * sizeof(T), 256);
    return b;
}
template <typename T>
static int index_alloc(Arena& a, GridIndex<T>& g, int64_t n, double occ, bool want_pos = false, bool allow_bucketed = true, bool one_pass = false, int64_t n_plan = 0) {
    // (n_plan: the cell / bucket plan of a larger cloud -- the partner this cloud shares its grid with, see GridIndex::shared_grid)
    const int64_t np = n_plan > n ? n_plan : n;
    g.n = (int)n; g.max_cells = max_cells_for(np, occ); g.scan_blocks = g.max_cells / kScanChunk + 1;
    g.bucketed = allow_bucketed && bucket_plan(np, occ, &g.shift, &g.nb_max);
    if (aalloc(a, &g.gp, 1)) return -1;
    // (cell_start sits 256 bytes INTO its block: the k = 1 / k > 1 lane kernels read the row table of a query in the first cell of the first row
    // from one word BEFORE cell_start (search.h: "uniform four-word tables"; the word is never used, but its address must be mapped -- also when
    // the block is an overflow hipMalloc of its own))
    if (aalloc(a, &g.cell_start, (size_t)g.max_cells + 1 + kBkMaxBuckets + 8 + 64)) return -1;    // + bucket totals + large-bucket count (zeroed together)
    g.cell_start += 64;
    if (a.alloc((void**)&g.sorted, sorted_records_bytes((size_t)n, sizeof(Pt4<T>), sizeof(T)))) return -1;        // + the +inf sentinel records + the coordinates-only copy (pcu_types.h: xyz_of)
    if (aalloc(a, &g.cell_of, (size_t)n)) return -1;
    if (aalloc(a, &g.rank, (size_t)n)) return -1;
    if (aalloc(a, &g.block_sums, (size_t)g.scan_blocks + 1)) return -1;
    if (aalloc(a, &g.bbox_partial, (size_t)kBboxBlocks * kBboxStride)) return -1;
    g.n_zero = g.max_cells + 1;
    if (g.bucketed) {
        g.bucket_total = g.cell_start + g.max_cells + 1; g.n_large = g.bucket_total + g.nb_max; g.n_zero = g.max_cells + 1 + g.nb_max + 1;
        g.one_pass = one_pass;
        if (aalloc(a, &g.tmp, one_pass ? std::max((size_t)n, (size_t)g.nb_max * kLargeBucket) : (size_t)n)) return -1;
        if (aalloc(a, &g.bucket_start, (size_t)g.nb_max + 1) || aalloc(a, &g.large_list, (size_t)g.nb_max + 1)) return -1;
        if (aalloc(a, &g.block_base, (size_t)((n + kBkBlockPts - 1) / kBkBlockPts) * g.nb_max)) return -1;
        if (one_pass && aalloc(a, &g.xpartial, (size_t)((n + 2047) / 2048) * kXPartStride)) return -1;        // (one partial per scatter block: 2048 points at least, grid2.h)
        g.pos_of = want_pos ? g.cell_of : nullptr;      // cell_of is not used by this build
    } else {
        g.pos_of = g.rank;                              // k_scatter turns rank into the slot, in place
    }
    return 0;
}
// Placement of the records of over-full buckets (grid.h): one launch for up to two indexes built back to back.
template <typename T>
static LargeJob<T> large_job(const GridIndex<T>& g) { return LargeJob<T>{g.gp, g.bucket_start, g.large_list, g.n_large, g.tmp, g.rank, g.cell_start, g.sorted, g.pos_of, g.n}; }
template <typename T>
static void index_large_pass(const GridIndex<T>& a, const GridIndex<T>* b, hipStream_t s) {
    const bool ua = a.bucketed, ub = b && b->bucketed;
    if (!ua && !ub) return;
    const LargeJob<T> ja = large_job(ua ? a : *b), jb = large_job(ua && ub ? *b : (ua ? a : *b));
    hipLaunchKernelGGL(k_bucket_large<T>, dim3(4 * kBboxBlocks), dim3(kBlock), 0, s, ja, jb, (ua && ub) ? 2 : 1);        // (grid-strided; 256 blocks left the chip three quarters empty: 111 us on a Gaussian cloud)
    // every record is placed now: searches may use the index (GridParams::has_large)
    if (ua) (void)hipMemsetAsync(reinterpret_cast<char*>(a.gp) + offsetof(GridParams<T>, has_large), 0, sizeof(int), s);
    if (ub) (void)hipMemsetAsync(reinterpret_cast<char*>(b->gp) + offsetof(GridParams<T>, has_large), 0, sizeof(int), s);
}
// Enqueue the build of one or two indexes on `s`: every pass is ONE launch serving both clouds (grid.h: blocks [0, nb0)
// work on the first, the rest on the second). No memset, no host synchronisation. zero2: a small region (the call's
// result block) zeroed on the way by the first launch.
template <typename T>
static BucketSide<T> bucket_side(const GridIndex<T>& g, const T* pts) {
    return BucketSide<T>{pts, g.n, g.gp, g.shift, g.nb_max, g.bucket_total, g.block_base, g.bucket_start, g.tmp, g.cell_start, g.rank,
                         g.sorted, g.pos_of, g.large_list, g.n_large, g.one_pass ? kLargeBucket : 0u};
}
template <typename T>
static int index_build_pair(GridIndex<T>& a, const T* pa, double occa, GridIndex<T>* b, const T* pb, double occb, hipStream_t s,
                            bool defer_large = false, void* zero2 = nullptr, int n_zero2 = 0, pcu_hip_ctx* ctx = nullptr, bool keep_layout = false) {
    a.src = pa; a.occ_built = occa;
    if (b) { b->src = pb; b->occ_built = occb; }
    // one launch set serves both clouds only if they are built the same way
    if (b && a.bucketed && b->bucketed && a.one_pass != b->one_pass) a.one_pass = b->one_pass = false;
    const GridSide<T> g0{a.gp, a.bbox_partial, kBboxBlocks, a.n, occa, a.max_cells, a.sorted + a.n, a.h_want, pa};
    const GridSide<T> g1 = b ? GridSide<T>{b->gp, b->bbox_partial, kBboxBlocks, b->n, occb, b->max_cells, b->sorted + b->n, b->h_want, pb} : g0;
    // When every cloud of the call takes the one-pass bucket build, its blocks lay out the grid themselves (grid.h: k_bucket_onepass)
    // and the k_make_grid launch is skipped. (bbox + grid layout in ONE launch, the last block folding the partials, was measured in
    // round 2: 16.7 us against 8.1 + 4.9 us for the two launches; removed.)
    static const bool grid_kernel = getenv("PCU_HIP_GRID_KERNEL") != nullptr;          // (always the separate k_make_grid launch)
    const bool grid_in_onepass = !grid_kernel && a.bucketed && a.one_pass && (!b || (b->bucketed && b->one_pass));
    // The one-pass build's second form (grid2.h): k_bucket_onepass3 -> k_bucket_sort2, while the bucket tables fit beside the scatter's stage.
    // PCU_HIP_BUILD_V1=1 (and the diagnostics of the first form, PCU_HIP_GRID_KERNEL / PCU_HIP_PROF_BUILD) keep the round-3 chain below.
    static const bool build_v1 = getenv("PCU_HIP_BUILD_V1") != nullptr || getenv("PCU_HIP_PROF_BUILD") != nullptr;
    // A scatter block's (block, bucket) runs must stay long for the staged copies to pay: below ~12 records per run the padding to whole
    // 8-record groups and the hole records the sort then reads cost more than the first form's per-record scatter (4M-point clouds:
    // 0.40 ms against 0.285, config 3).
    const int run_floor = 12 * std::max(a.nb_max, b ? b->nb_max : 0);
    if (grid_in_onepass && !build_v1 && ctx && ctx->fill2 && a.xpartial && (!b || b->xpartial) && a.nb_max <= kStagedMaxBuckets && (!b || b->nb_max <= kStagedMaxBuckets) &&
        kBkThreads * StagedPts<T>::n >= run_floor) {
        unsigned long long* const fw = ctx->fill2 + (size_t)ctx->fill_parity * kFillWords;
        unsigned long long* const fw_next = ctx->fill2 + (size_t)(ctx->fill_parity ^ 1) * kFillWords;
        // points per thread of the scatter blocks: the most (longest runs per (block, bucket), fewest reservations); PCU_HIP_BUILD_PTS fixes it (A/B)
        static const int pts_env = getenv("PCU_HIP_BUILD_PTS") ? atoi(getenv("PCU_HIP_BUILD_PTS")) : 0;
        int pts = StagedPts<T>::n;
        {
            const long long ntot = (long long)a.n + (b ? b->n : 0);
            static const int n_cu = [] { hipDeviceProp_t pr; int d = 0; (void)hipGetDevice(&d); return hipGetDeviceProperties(&pr, d) == hipSuccess ? pr.multiProcessorCount : 256; }();
            // (measured, profiles/r06_build_ab.txt: halving the blocks to get a block per CU -- or two per CU at 2 x 1M -- LOSES: every stage of a
            // block takes as long with half the points, reservations and padding double; only launches of a handful of blocks are cut up)
            while (pts > 2 && (ntot + (long long)kBkThreads * pts - 1) / ((long long)kBkThreads * pts) < n_cu / 16 && kBkThreads * (pts / 2) >= run_floor) pts /= 2;
            if (pts_env == 2 || pts_env == 4 || pts_env == 8) pts = std::min(pts_env, (int)StagedPts<T>::n);
        }
        const int bpts = kBkThreads * pts;
        const int nbcap = (std::max(a.nb_max, b ? b->nb_max : 0) + 63) / 64 * 64;
        auto side = [&](const GridIndex<T>& g, const T* p, double occ, int k) {
            return Build2Side<T>{p, g.n, g.gp, g.shift, occ, g.max_cells, g.h_want, fw + (size_t)k * kStagedMaxBuckets, fw + 2 * kStagedMaxBuckets + k,
                                 g.tmp, kLargeBucket, g.xpartial, (g.n + bpts - 1) / bpts, g.cell_start, g.sorted, g.pos_of, g.lean ? 0 : 1, g.n_large,
                                 k == 0 ? fw_next : nullptr, k == 0 ? kFillWords : 0, k == 0 ? (unsigned*)zero2 : nullptr, k == 0 ? n_zero2 : 0, nullptr,
                                 p, g.n, nullptr, 0, g.n, nullptr, nullptr};
        };
        Build2Side<T> s0 = side(a, pa, occa, 0), s1 = b ? side(*b, pb, occb, 1) : s0;
        // one grid for both clouds (GridIndex::shared_grid): same plan (index_alloc's n_plan), same occupancy, both at least a sample large
        const bool shared = b && a.shared_grid && b->shared_grid && occa == occb && a.max_cells == b->max_cells && a.shift == b->shift && a.nb_max == b->nb_max &&
                            a.h_want == b->h_want && a.n >= kPrepSamples && b->n >= kPrepSamples;
        if (b) a.shared_grid = b->shared_grid = shared; else a.shared_grid = false;
        if (shared) {
            s0.spts0 = s1.spts0 = pa; s0.sn0 = s1.sn0 = a.n; s0.spts1 = s1.spts1 = pb; s0.sn1 = s1.sn1 = b->n;
            s0.n_layout = s1.n_layout = std::max(a.n, b->n);
        }
        // The layout handed down from the context's previous call (grid2.h: GridGeo): two-sided fused calls only (their *_end knows how to restart a
        // call whose layout was refused as stale). The key: everything grid_layout and the sample depend on besides the points themselves.
        static const bool geo_off = getenv("PCU_HIP_NO_GEO_CACHE") != nullptr;
        bool geo_arm = false;
        if (keep_layout && b && !geo_off && ctx->geo.dev && a.n >= kPrepSamples && b->n >= kPrepSamples) {
            pcu_hip_ctx::GeoCache& gc = ctx->geo;
            const bool hit = gc.valid[0] && gc.valid[1] && gc.n[0] == a.n && gc.n[1] == b->n && gc.occ == occa && occa == occb && gc.h_want == a.h_want && a.h_want == b->h_want &&
                             gc.max_cells == a.max_cells && a.max_cells == b->max_cells && gc.shared == shared && gc.n_layout == s0.n_layout && gc.tsize == (int)sizeof(T);
            static_assert(sizeof(GridGeo<T>) <= 256, "two layouts fit the context's block");
            GridGeo<T>* const g0 = reinterpret_cast<GridGeo<T>*>(gc.dev), *const g1 = reinterpret_cast<GridGeo<T>*>(gc.dev + 256);
            s0.geo_out = g0; s1.geo_out = g1;
            if (hit) { s0.geo_in = g0; s1.geo_in = g1; }
            if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[layout] handed down: %d (n %d %d, shared %d)\n", (int)hit, a.n, b->n, (int)shared);
            // (valid again only when both launches are enqueued, below: a build that fails on the way leaves no claim on memory nobody wrote)
            gc.valid[0] = gc.valid[1] = false;
            geo_arm = occa == occb && a.h_want == b->h_want && a.max_cells == b->max_cells;
            gc.n[0] = a.n; gc.n[1] = b->n; gc.occ = occa; gc.h_want = a.h_want; gc.max_cells = a.max_cells; gc.shared = shared; gc.n_layout = s0.n_layout; gc.tsize = (int)sizeof(T);
        }
        const int c0 = s0.n_xpart, c1 = b ? s1.n_xpart : 0;
        static const bool do_prof2 = getenv("PCU_HIP_PROF_BUILD2") != nullptr;
        static long long* prof2 = nullptr;
        if (do_prof2) { if (!prof2) HIP_TRY(hipMalloc((void**)&prof2, 16 * sizeof(long long))); HIP_TRY(hipMemsetAsync(prof2, 0, 16 * sizeof(long long), s)); }
        s0.prof = s1.prof = do_prof2 ? prof2 : nullptr;
        static std::atomic<unsigned long long> attr_set2[2];
        if (attr_unset_here(attr_set2[sizeof(T) == 4 ? 0 : 1])) {
            if (StagedPts<T>::n >= 8) HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_onepass3<T, 8>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)onepass3_lds_bytes<T>(8)));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_onepass3<T, 4>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)onepass3_lds_bytes<T>(4)));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_onepass3<T, 2>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)onepass3_lds_bytes<T>(2)));
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_sort2<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bucket_sort_lds_bytes<T>(kBkMaxCellsPerBucket)));
        }
        Build2Args<T> sa; sa.a[0] = s0; sa.a[1] = s1;
        const size_t lds1 = onepass3_lds_bytes<T>(pts, nbcap);
        if (pts == 8) hipLaunchKernelGGL((k_bucket_onepass3<T, 8>), dim3(c0 + c1), dim3(kBkThreads), lds1, s, sa, c0, nbcap);
        else if (pts == 4) hipLaunchKernelGGL((k_bucket_onepass3<T, 4>), dim3(c0 + c1), dim3(kBkThreads), lds1, s, sa, c0, nbcap);
        else hipLaunchKernelGGL((k_bucket_onepass3<T, 2>), dim3(c0 + c1), dim3(kBkThreads), lds1, s, sa, c0, nbcap);
        if (do_prof2) {
            long long h[16]; HIP_TRY(hipMemcpyAsync(h, prof2, sizeof h, hipMemcpyDeviceToHost, s)); HIP_WAIT(s);
            const double nb = h[15] > 0 ? (double)h[15] * 100.0 : 100.0;
            fprintf(stderr, "[onepass3 prof] blocks %lld | mean us per block: layout %.2f  points in %.2f  keys+ranks %.2f  scan+reservations %.2f  staging %.2f  run copies %.2f  drain %.2f\n",
                    h[15], h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb, h[5] / nb, h[6] / nb);
        }
        const int t0 = a.nb_max, t1 = b ? b->nb_max : 0;
        const int cnt_cap = 1 << std::max(a.shift, b ? b->shift : 0);
        if (do_prof2) { HIP_TRY(hipMemsetAsync(prof2, 0, 16 * sizeof(long long), s)); }
        hipLaunchKernelGGL(k_bucket_sort2<T>, dim3(t0 + t1 + (b ? 2 : 1)), dim3(kSortThreads), bucket_sort_lds_bytes<T>(cnt_cap), s, sa, t0, t1, cnt_cap, b ? 2 : 1);
        ctx->fill_parity ^= 1;        // only now: both launches are enqueued, so the other set WILL be zeroed for the next build (an error return above leaves the parity alone)
        if (do_prof2) {
            long long h[16]; HIP_TRY(hipMemcpyAsync(h, prof2, sizeof h, hipMemcpyDeviceToHost, s)); HIP_WAIT(s);
            const double nb = h[7] > 0 ? (double)h[7] * 100.0 : 100.0;
            fprintf(stderr, "[sort2 prof] blocks %lld | mean us per block: head %.2f  load+rank %.2f  scan %.2f  place %.2f  copies %.2f  drain %.2f\n",
                    h[7], h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb, h[4] / nb, h[5] / nb);
        }
        HIP_TRY(hipGetLastError());
        if (geo_arm) ctx->geo.valid[0] = ctx->geo.valid[1] = true;       // (the sort launch's first blocks will have written this call's layout before the next call's kernels run)
        return 0;
    }
    a.lean = false; if (b) b->lean = false;            // (every other build writes the Pt4 records)
    a.shared_grid = false; if (b) b->shared_grid = false;      // (... and lays every cloud over its own grid)
    {
        const BboxSide<T> s0{pa, a.n, a.bbox_partial, a.cell_start, a.n_zero, (unsigned*)zero2, n_zero2, a.gp};
        const BboxSide<T> s1 = b ? BboxSide<T>{pb, b->n, b->bbox_partial, b->cell_start, b->n_zero, nullptr, 0, b->gp} : s0;
        hipLaunchKernelGGL(k_bbox_partial<T>, dim3(b ? 2 * kBboxBlocks : kBboxBlocks), dim3(kBlock), 0, s, s0, s1, kBboxBlocks);
        if (!grid_in_onepass) hipLaunchKernelGGL(k_make_grid<T>, dim3(b ? 2 : 1), dim3(kBlock), 0, s, g0, g1);
    }
    // bucketed sides share their launches; a side too small / too coarse for buckets takes the atomic passes
    const GridIndex<T>* bs[2]; const T* bp[2]; int nbs = 0;
    if (a.bucketed) { bs[nbs] = &a; bp[nbs] = pa; ++nbs; }
    if (b && b->bucketed) { bs[nbs] = b; bp[nbs] = pb; ++nbs; }
    if (nbs) {
        const BucketSide<T> s0 = bucket_side(*bs[0], bp[0]), s1 = nbs > 1 ? bucket_side(*bs[1], bp[1]) : s0;
        const int c0 = (bs[0]->n + kBkBlockPts - 1) / kBkBlockPts, c1 = nbs > 1 ? (bs[1]->n + kBkBlockPts - 1) / kBkBlockPts : 0;
        static long long* prof = nullptr;       // PCU_HIP_PROF_BUILD: stage times of k_bucket_sort, printed per build (synchronises)
        static const bool do_prof = getenv("PCU_HIP_PROF_BUILD") != nullptr;
        if (do_prof && !prof) HIP_TRY(hipMalloc((void**)&prof, 8 * sizeof(long long)));
        if (do_prof) HIP_TRY(hipMemsetAsync(prof, 0, 8 * sizeof(long long), s));
        const bool one_pass = bs[0]->one_pass;
        if (one_pass) {
            static long long* prof1 = nullptr;
            if (do_prof && !prof1) HIP_TRY(hipMalloc((void**)&prof1, 8 * sizeof(long long)));
            if (do_prof) HIP_TRY(hipMemsetAsync(prof1, 0, 8 * sizeof(long long), s));
            // (the grid sides in the order of the bucket sides: both clouds are bucketed whenever two are built this way)
            hipLaunchKernelGGL(k_bucket_onepass<T>, dim3(c0 + c1), dim3(kBkThreads), 0, s, s0, s1, c0, do_prof ? prof1 : nullptr,
                               bs[0] == &a ? g0 : g1, nbs > 1 ? g1 : (bs[0] == &a ? g0 : g1));
            if (do_prof) {
                long long h[8]; HIP_TRY(hipMemcpyAsync(h, prof1, sizeof h, hipMemcpyDeviceToHost, s)); HIP_WAIT(s);
                const double nb = h[7] > 0 ? (double)h[7] : 1.0;
                fprintf(stderr, "[onepass prof] blocks %lld | mean us per block: zero+loads %.2f  keys+LDS ranks %.2f  slot reservations %.2f  stores %.2f\n", h[7],
                        h[0] / nb / 100.0, h[1] / nb / 100.0, h[2] / nb / 100.0, h[3] / nb / 100.0);
            }
        }
        else {
            hipLaunchKernelGGL(k_bucket_count<T>, dim3(c0 + c1), dim3(kBkThreads), 0, s, s0, s1, c0);
            hipLaunchKernelGGL(k_bucket_scatter<T>, dim3(c0 + c1), dim3(kBkThreads), 0, s, s0, s1, c0);
        }
        const int t0 = bs[0]->nb_max, t1 = nbs > 1 ? bs[1]->nb_max : 0;
        const int cnt_cap = 1 << std::max(bs[0]->shift, nbs > 1 ? bs[1]->shift : 0);
        const size_t lds = bucket_sort_lds_bytes<T>(cnt_cap);
        static std::atomic<unsigned long long> attr_set[2];
        if (attr_unset_here(attr_set[sizeof(T) == 4 ? 0 : 1]))
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_bucket_sort<T>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                        (int)bucket_sort_lds_bytes<T>(kBkMaxCellsPerBucket)));
        // One launch for both clouds, like the other passes (1024 threads / 4096-point buckets: all blocks of both clouds are
        // resident at once; measured 0.174 vs 0.186 ms per step against one launch per cloud).
        hipLaunchKernelGGL(k_bucket_sort<T>, dim3(t0 + t1), dim3(kSortThreads), lds, s, s0, s1, t0, do_prof ? prof : nullptr, cnt_cap);
        if (do_prof) {
            long long h[8]; HIP_TRY(hipMemcpyAsync(h, prof, sizeof h, hipMemcpyDeviceToHost, s)); HIP_WAIT(s);
            const double nb = h[7] > 0 ? (double)h[7] : 1.0;
            fprintf(stderr, "[bucket_sort prof] blocks %lld | mean us per block: head %.2f  zero+sync %.2f  load+rank %.2f  scan %.2f  place %.2f\n", h[7],
                    h[0] / nb / 100.0, h[1] / nb / 100.0, h[2] / nb / 100.0, h[3] / nb / 100.0, h[4] / nb / 100.0);
        }
        if (!defer_large && !one_pass) index_large_pass<T>(a, b, s);        // (a one-pass build has no over-full buckets: it overflows instead)
    }
    for (int side = 0; side < (b ? 2 : 1); ++side) {
        GridIndex<T>& g = side ? *b : a;
        if (g.bucketed) continue;
        const T* d_pts = side ? pb : pa;
        const int n = g.n, nb = (n + kBlock - 1) / kBlock;
        hipLaunchKernelGGL(k_count<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.gp, g.cell_of, g.rank, g.cell_start);
        hipLaunchKernelGGL(k_scan_reduce<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums);
        hipLaunchKernelGGL(k_scan_apply<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums, (unsigned)n);
        hipLaunchKernelGGL(k_scatter<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.cell_of, g.rank, g.cell_start, g.sorted);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int index_build(GridIndex<T>& g, const T* d_pts, double occ, hipStream_t s, bool defer_large = false, void* zero2 = nullptr, int n_zero2 = 0,
                       pcu_hip_ctx* ctx = nullptr) {
    return index_build_pair<T>(g, d_pts, occ, nullptr, nullptr, 0.0, s, defer_large, zero2, n_zero2, ctx);
}

// Refitted grids for unbalanced clouds (grid.h): core range of the cloud by three zooming histogram rounds, then
// uniform grids of a chosen cell count over that range. Everything is enqueued on `s` (no host sync).
template <typename T>
static int core_range_enqueue(Arena& ar, const GridIndex<T>& base, const T* d_pts, hipStream_t s, QuantState<T>** out_qs) {
    QuantState<T>* qs = nullptr; unsigned *partial = nullptr, *hist = nullptr;
    if (aalloc(ar, &qs, 1) || aalloc(ar, &partial, (size_t)kHistBlocks * 3 * (kHistBins + 2)) || aalloc(ar, &hist, 3 * (kHistBins + 2))) return -1;
    hipLaunchKernelGGL(k_quant_init<T>, dim3(1), dim3(64), 0, s, base.gp, qs);
    for (int r = 0; r < 3; ++r) {
        hipLaunchKernelGGL(k_hist_axis<T>, dim3(kHistBlocks), dim3(kBlock), 0, s, d_pts, base.n, qs, partial);
        hipLaunchKernelGGL(k_hist_merge, dim3((3 * (kHistBins + 2) + kBlock - 1) / kBlock), dim3(kBlock), 0, s, partial, kHistBlocks, hist);
        hipLaunchKernelGGL(k_quant_zoom<T>, dim3(3), dim3(64), 0, s, qs, hist, base.n);
    }
    HIP_TRY(hipGetLastError());
    *out_qs = qs;
    return 0;
}
template <typename T>
static int index_build_refit(Arena& ar, GridIndex<T>& g, const GridIndex<T>& base, const T* d_pts, const QuantState<T>* qs,
                             double target_cells, hipStream_t s, bool closed = false, const double* target_dev = nullptr) {
    // closed: sub-box level (only the points inside the box are indexed); target_dev: cell count decided on the device
    const int n = base.n;
    if (target_cells < 1.0) target_cells = 1.0;
    if (index_alloc(ar, g, n, (double)n / target_cells, false, /*allow_bucketed=*/false)) return -1;
    const int nb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_make_grid_refit<T>, dim3(1), dim3(64), 0, s, g.gp, base.gp, qs, target_cells, g.max_cells, g.sorted + n, n,
                       closed ? 1 : 0, target_dev);
    HIP_TRY(hipMemsetAsync(g.cell_start, 0, ((size_t)g.max_cells + 1) * 4, s));
    hipLaunchKernelGGL(k_count<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.gp, g.cell_of, g.rank, g.cell_start);
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums, closed ? 0xffffffffu : (unsigned)n);
    hipLaunchKernelGGL(k_scatter<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.cell_of, g.rank, g.cell_start, g.sorted);
    HIP_TRY(hipGetLastError());
    return 0;
}
// Sub-box level over the heavy cells of `parent` (cells holding more than `thresh` points): enqueue only.
template <typename T>
static int index_build_heavy(Arena& ar, GridIndex<T>& g, const GridIndex<T>& parent, const T* d_pts, double occ, unsigned thresh, hipStream_t s,
                             const double** stats_dev = nullptr) {
    // stats_dev: device pair {cell count chosen for the level, number of points in heavy cells of the parent}
    QuantState<T>* qs = nullptr; T* pbox = nullptr; double *pcnt = nullptr, *target = nullptr;
    if (aalloc(ar, &qs, 1) || aalloc(ar, &pbox, (size_t)kBboxBlocks * 6) || aalloc(ar, &pcnt, (size_t)kBboxBlocks * 2) || aalloc(ar, &target, 2)) return -1;
    // (a bucketed parent keeps no per-point cell ids -- its cell_of storage is the row -> slot table -- the kernel recomputes them)
    hipLaunchKernelGGL(k_heavy_partial<T>, dim3(kBboxBlocks), dim3(kBlock), 0, s, d_pts, parent.n, parent.gp, parent.bucketed ? nullptr : parent.cell_of,
                       parent.cell_start, thresh, pbox, pcnt);
    hipLaunchKernelGGL(k_heavy_finish<T>, dim3(1), dim3(64), 0, s, parent.gp, pbox, pcnt, kBboxBlocks, occ, 16.0 * 1024 * 1024, qs, target);
    if (stats_dev) *stats_dev = target;
    return index_build_refit(ar, g, parent, d_pts, qs, (double)parent.n / occ, s, /*closed=*/true, target);
}

// ------------------------------------------------------------------------------------------------ search driver
static double default_occupancy(int k) {
    // Dataset points per grid cell. More points per cell = more candidates per query in the main pass but fewer
    // queries whose k-th neighbour lies beyond the certified radius (one cell edge) and must be re-run by the
    // wave-per-query passes. Measured optimum on uniform data (profiles/r01_occupancy_sweep.txt): 2.0 for k = 1,
    // 7-8 for k = 16; in between the k-th neighbour radius scales with (k + 3 sqrt k)^(1/3).
    if (k <= 1) return 2.0;
    return std::max(2.0, (k + 3.0 * sqrt((double)k)) / 3.7);
}
static int pow2_at_least(int k) { int p = 1; while (p < k) p <<= 1; return p; }
constexpr int kMaxKLane = 32;       // lane-per-query register slots (K = 64 needs 203-275 VGPRs: those k go wave-per-query from the start)
constexpr int kMaxK = 127;          // the wave-per-query kernel holds k+1 <= 128 slots per lane
// Fewer queries than this: wave-per-query from the start (PCU_HIP_WAVE_ONLY_BELOW overrides). Round 5: 16384 -> 64, together with the one-pass
// index build from 64 points on (bucket_plan; round 4: 32768): a 10k-vs-10k call is then the same four launches as a 1M-vs-1M one (build x 2, lane
// pass, tail) instead of thirteen (bbox, grid, 4 atomic-build passes per cloud, two wave-per-query launches, epilogue): config 1 0.159 -> 0.056 ms,
// the 2 885-query direction of config 5 0.246 -> 0.201 ms, and clouds of 64 .. 2000 points 0.058-0.084 -> 0.044-0.055 ms per Chamfer
// (profiles/r05_small_ab.txt). A handful of blocks of a lane pass do not fill the GPU -- a launch's latency chain does not care.
static int wave_only_below() { static const int v = getenv("PCU_HIP_WAVE_ONLY_BELOW") ? atoi(getenv("PCU_HIP_WAVE_ONLY_BELOW")) : 64; return v; }
#define kWaveOnlyBelow wave_only_below()
constexpr double kSkewFactor = 32.0;    // dataset grid considered unbalanced when sum(count^2)/n > 32 x (occupancy + 1): a lane pass
                                        // costs ~30 us per unit of that ratio at 1M queries, a refit ~3 ms (scratch/skew.py)
// Occupancy rescale. The default occupancy is right for clouds that fill their bounding box. A cloud sampled from a surface puts
// its points into a thin shell of cells: at 1M points the occupied cells hold ~50 points instead of 2 and the lane pass scans
// 9 x 50 candidates (sphere surface 0.39 ms, mesh samples 0.49 ms per Chamfer against 0.17 ms uniform). The balance metric the
// bucket sort already computes (sumsq / n = mean number of cell mates) shows it: above kRescaleAbove x (occupancy + 1) the passes
// give up at once, the host restarts the call with cells finer by sqrt(ratio) in volume (measured optimum on surfaces: 4-8x
// finer, 0.27-0.28 ms) and the context keeps that scale for its next calls; a call that then meets a cloud that is as even as
// a volume-filling one at the finer scale (ratio below kRescaleBelow -- such a cloud would be 20x slower on too fine a grid)
// restarts with the default. Ratios above kSkewFactor still take the refit path. Off when the caller fixes the occupancy
// (pcu_hip_ctx_set_cell_occupancy), for persistent indexes, and with PCU_HIP_NO_RESCALE.
constexpr double kRescaleAbove = 6.0, kRescaleBelow = 3.0, kRescaleMax = 8.0;
constexpr int PCU_RETRY = 1000;         // internal: restart the call (the context's occ_scale changed)
static bool rescale_enabled(const pcu_hip_ctx* c) { static const bool off = getenv("PCU_HIP_NO_RESCALE") != nullptr; return !off && !(c->occupancy > 0); }
static double call_occupancy(const pcu_hip_ctx* c, int k, int role) { return c->occupancy > 0 ? c->occupancy : default_occupancy(k) * c->occ_scale[role]; }
#ifndef PCU_WAVE_BLOCKS
#define PCU_WAVE_BLOCKS 512
#endif
constexpr int kWaveBlocks = PCU_WAVE_BLOCKS;    // fixed grid of the wave-cooperative passes: 2048 waves striding a device-side list

// ONE grid over both clouds of a two-sided call (grid2.h: Build2Side::spts1; what search_brick.h's staged pass needs): clouds of comparable size
// indexed at the same occupancy. PCU_HIP_NO_SHARED_GRID=1: every cloud its own grid, as before round 6.
static bool shared_grid_wanted(const pcu_hip_ctx* c, int64_t nx, int64_t ny, double occ_x, double occ_y) {
    static const bool off = getenv("PCU_HIP_NO_SHARED_GRID") != nullptr;
    return !off && !c->brick_off && occ_x == occ_y && std::min(nx, ny) >= kPrepSamples && std::max(nx, ny) <= 2 * std::min(nx, ny);
}
// whole-call index builds use the one-pass bucket scatter until a cloud of this context overflows a slot (PCU_HIP_TWO_PASS=1: never)
static bool use_one_pass(const pcu_hip_ctx* c) { static const bool off = getenv("PCU_HIP_TWO_PASS") != nullptr; return !off && !c->two_pass; }
// The wave-per-query launch of a call finishes its stragglers itself (search.h: k_search_wave, box round -> ball round); PCU_HIP_NO_ESCALATE=1
// leaves them to the host-driven passes of search_finish (radius 4 ... 16, then coarser grids), the pre-round-3 behaviour and still the
// path of closed sub-box levels.
static int wave_escalates() { static const bool off = getenv("PCU_HIP_NO_ESCALATE") != nullptr; return off ? 0 : 1; }
static bool use_k1_kernel() { static const bool v = getenv("PCU_HIP_NO_K1") == nullptr; return v; }
static int grid8(int nwork, int tb) { return (((nwork + tb - 1) / tb) + 7) / 8 * 8; }       // multiple of 8: XCD-aware block map

template <typename T> static void launch_brick(const SearchArgs2<T>&, int, int, hipStream_t) {}
template <> void launch_brick<float>(const SearchArgs2<float>& p2, int b0, int b1, hipStream_t s) {
    hipLaunchKernelGGL((k_search1_brick<float, kBrickNT>), dim3(b0 + b1), dim3(kBrickNT), 0, s, p2, b0);
}
// Main (lane-per-query) pass of one direction, or -- k = 1 on open indexes -- of both directions of a two-sided call in
// one launch (a1 / nwork1).
template <typename T>
static int launch_search_fast(int K, const SearchArgs<T>& a, int nwork, hipStream_t s, bool open_index,
                              const SearchArgs<T>* a1 = nullptr, int nwork1 = 0) {
    if (nwork <= 0) return 0;
    // (Rejected variants, measured on MI355X in rounds 1-2 and removed from the tree: a one-wave-per-block LDS-tiled kernel, 129 vs 83 us
    // at 1M / k = 1, profiles/r01_search_kernel_ab.txt; a per-wave work-queue variant of the k = 1 pass, 84-87 vs 82-83 us, profiles/r02_ubench.txt.)
    const int tb = kBlock;
    if (K == 1 && use_k1_kernel() && open_index) {        // k = 1 on an open index: the group-wise flat kernel
        SearchArgs2<T> p2; p2.a[0] = a; p2.a[1] = a1 ? *a1 : a;
        if (sizeof(T) == 4 && a.fuse == FUSE_SUM && a.brick && a1 && a1->brick && !a.qlist && !a1->qlist) {       // shared grid: the staged pass (search_brick.h)
            const int b0 = grid8(nwork, kBrickNT), b1 = grid8(nwork1, kBrickNT);
            launch_brick<T>(p2, b0, b1, s);
            HIP_TRY(hipGetLastError());
            return 0;
        }
        const int g0 = grid8(nwork, tb), g1 = a1 ? grid8(nwork1, tb) : 0;
        // (An LDS-staged, block-cooperative variant of this pass -- the north-star's tile design -- was measured again in round 4: 243-593 us
        // against 77 us, profiles/r04_flat_tile_ab.txt; removed.)
#ifndef PCU_FLAT_MINW
#define PCU_FLAT_MINW 8
#endif
#ifndef PCU_FLAT_MINW_ROWS
#define PCU_FLAT_MINW_ROWS 4
#endif
#define PCU_FLAT(FUSE) hipLaunchKernelGGL((k_search1_flat<T, false, sizeof(T) == 4 ? (FUSE == FUSE_SUM ? PCU_FLAT_MINW : PCU_FLAT_MINW_ROWS) : 4, FUSE>), dim3(g0 + g1), dim3(tb), 0, s, p2, g0)
        // (a variant that deals a wave's candidate groups evenly to its lanes -- LDS list + atomic min -- measured 75.6 vs 76.5 us: the loop is
        // not where the instructions are, profiles/r04_flat_deal_ab.txt; removed)
        if (a.fuse == FUSE_SUM) PCU_FLAT(FUSE_SUM);
        else if (a.fuse == FUSE_ARGMAX && a.maxval && (!a1 || a1->maxval)) hipLaunchKernelGGL((k_search1_flat<T, false, sizeof(T) == 4 ? PCU_FLAT_MINW : 4, FUSE_MAXVAL>), dim3(g0 + g1), dim3(tb), 0, s, p2, g0);
        else if (a.fuse == FUSE_ARGMAX) PCU_FLAT(FUSE_ARGMAX);
        else PCU_FLAT(FUSE_NONE);
#undef PCU_FLAT
        HIP_TRY(hipGetLastError());
        return 0;
    }
    if (a1) return fail(PCU_HIP_ERR_RUNTIME, "internal: paired main pass without the k = 1 kernel");
    dim3 grid(grid8(nwork, tb)), block(tb);
    // k > 1 on an open index: the run-list kernel (search.h: k_search_runs, round 5); PCU_HIP_KSEARCH_V1=1 keeps k_search everywhere (A/B, switch test)
    static const bool runs_off = getenv("PCU_HIP_KSEARCH_V1") != nullptr;
    if (K > 1 && open_index && !runs_off) {
#define PCU_CASE(KK) case KK: hipLaunchKernelGGL((k_search_runs<T, KK>), grid, block, 0, s, a); break;
        switch (K) {
            case 2: PCU_CASE(4) PCU_CASE(8) PCU_CASE(16) PCU_CASE(32)
            default: return fail(PCU_HIP_ERR_INVALID, "internal: unsupported K=%d", K);
        }
#undef PCU_CASE
        HIP_TRY(hipGetLastError());
        return 0;
    }
#define PCU_CASE(KK) case KK: hipLaunchKernelGGL((k_search<T, KK>), grid, block, 0, s, a); break;
    switch (K) {
        PCU_CASE(1) case 2: PCU_CASE(4) PCU_CASE(8) PCU_CASE(16) PCU_CASE(32)      // (k = 2 keeps a list of 4: one instantiation pair less, 0.2 MB)
        default: return fail(PCU_HIP_ERR_INVALID, "internal: unsupported K=%d", K);
    }
#undef PCU_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}
template <typename T>
static int launch_search_wave(int K, const SearchArgs<T>& a, hipStream_t s, const SearchArgs<T>* a1 = nullptr) {
    // lists: fixed grid striding a device-side count; whole-cloud passes (a.nq given): one wave per query up to 64k waves
    // (fused calls: the fixed grid always -- the arg-max slots are one per wave of that grid)
    // (round 6: long lists per lane -- K >= 16, registers for 3 waves per SIMD at most -- take twice the grid: config 3's 4.5k stragglers and
    // possible ties 3.45 -> 3.30 ms; at K = 2 the larger grid costs the Gaussian cloud's 33k-straggler pass 3 %, profiles/r06_c3_ab.txt)
    const int list_blocks = (K >= 16 && a.fuse == FUSE_NONE) ? 2 * kWaveBlocks : kWaveBlocks;
    const int blocks = (a.qcount_dev || a.fuse != FUSE_NONE) ? list_blocks : std::max(1, std::min((a.nq + (a1 && !a1->qcount_dev ? a1->nq : 0) + 3) / 4, 16384));
    dim3 grid(blocks), block(kBlock);
#define PCU_CASE(KK) case KK: hipLaunchKernelGGL((k_search_wave<T, KK>), grid, block, 0, s, a, a1 ? *a1 : a, a1 ? 2 : 1, blocks); break;
    switch (K) {
        PCU_CASE(2) PCU_CASE(4) PCU_CASE(8) PCU_CASE(16) PCU_CASE(32) PCU_CASE(64) PCU_CASE(128)
        default: return fail(PCU_HIP_ERR_INVALID, "internal: unsupported K=%d", K);
    }
#undef PCU_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}

// Per-direction lists and counters. Counter slots:
enum { C_U1 = 0, C_T1 = 1, C_U2 = 2, C_U3 = 3, C_TT = 4, C_SPARE = 5, C_SKEW = 6, C_X0 = 7, C_X1 = 8, C_LARGE = 9, C_TF0 = 10, C_TF1 = 11, C_N = 12 };
static_assert(C_LARGE - C_SKEW == kLargeFlag, "search.h addresses the large-bucket flag relative to the skew flag");
template <typename T>
struct SearchScratch {
    int *u1 = nullptr, *u2 = nullptr, *u3 = nullptr, *t1 = nullptr, *tt = nullptr, *x0 = nullptr, *x1 = nullptr;
    T* ub1 = nullptr;               // the lane pass's k-th best of every straggler in u1 (search.h: SearchArgs::ubound)
    int* counters = nullptr;
    int nq = 0;
};
template <typename T>
static size_t scratch_bytes(int64_t nq) { return 7 * align_up((size_t)nq * 4, 256) + align_up((size_t)nq * sizeof(T), 256) + 256; }
template <typename T>
static int scratch_alloc(Arena& a, SearchScratch<T>& sc, int64_t nq, int* counters_ext = nullptr) {
    sc.nq = (int)nq;
    if (aalloc(a, &sc.u1, (size_t)nq) || aalloc(a, &sc.u2, (size_t)nq) || aalloc(a, &sc.u3, (size_t)nq)) return -1;
    if (aalloc(a, &sc.t1, (size_t)nq) || aalloc(a, &sc.tt, (size_t)nq)) return -1;
    if (aalloc(a, &sc.x0, (size_t)nq) || aalloc(a, &sc.x1, (size_t)nq) || aalloc(a, &sc.ub1, (size_t)nq)) return -1;
    sc.counters = counters_ext;
    if (!sc.counters && aalloc(a, &sc.counters, C_N)) return -1;
    return 0;
}

template <typename T>
struct SearchJob {           // one direction: queries of `qidx` against the dataset `ridx`
    GridIndex<T> qidx, ridx;
    const T* d_ref_pts = nullptr;
    double occ = 1.5;
    int k = 1; bool squared = false;
    bool row_out = false;                       // results written straight to the caller's row order (k_nearest_neighbors, k >= 4)
    int leaf_max = 10; bool tie_order = true;   // reference's max_points_per_leaf: defines the order of exact ties
    int n_tt = 0;                               // genuine-tie queries found (filled by search_finish)
    bool skew_check = true;                     // give up early on a badly unbalanced dataset grid (then: refitted finer grids)
    double skew_hi = kSkewFactor, skew_lo = 0.0; // ... thresholds on sumsq / n / (occ + 1); rescale: see kRescaleAbove
    int bad_r = kNfNaN | kNfBothInf, bad_q = 0;  // non-finite input this operator rejects (grid.h: kNf*): k_nearest_neighbors takes any query and every
                                                 // dataset the reference's kd-tree survives; the two-sided metrics reject all non-finite input
    bool may_rescale = false; int role = 1;      // role: which cloud of the call the dataset is (pcu_hip_ctx::occ_scale)
    GridIndex<T> fine[2]; int n_fine = 0;       // finer dataset grids for the dense parts, finest first (unbalanced clouds only)
    int* fine_ties[2] = {nullptr, nullptr};     // possible-tie lists of the lane passes over those levels (counters C_TF0 / C_TF1)
    T* out_d = nullptr; long long* out_i = nullptr;
    SearchScratch<T> sc;
    // fused epilogue (reduce.h): per-block partials of the k = 1 lane pass instead of result rows
    int fuse = FUSE_NONE; int n_flat = 0;
    bool maxval = false;                        // fused arg-max: value-only lane pass, the winner's neighbour resolved by k_fuse_tail (reduce.h: FuseTail::maxval)
    bool brick = false;                         // the lane pass is search_brick.h's staged pass (shared grid, fused sum, float)
    double* f_sum = nullptr; T* f_max_v = nullptr; long long* f_max_k = nullptr;
    unsigned long long* f_limbs = nullptr; double* f_special = nullptr; T* f_wave_v = nullptr; long long* f_wave_k = nullptr;
};

template <typename T>
static SearchArgs<T> base_args(const SearchJob<T>& j, const GridIndex<T>& ridx) {
    SearchArgs<T> a;
    a.gp = ridx.gp; a.ref = ridx.sorted; a.cell_start = ridx.cell_start; a.qsorted = j.qidx.sorted; a.n_ref = (unsigned)ridx.n;
    a.ref_xyz = xyz_of(ridx.sorted, ridx.n); a.ref_idx = idx32_of(ridx.sorted, ridx.n);
    a.q_xyz = xyz_of(j.qidx.sorted, j.qidx.n); a.q_idx = idx32_of(j.qidx.sorted, j.qidx.n);
    a.lean = (ridx.lean || j.qidx.lean) ? 1 : 0;
    a.qlist = nullptr; a.qcount_dev = nullptr; a.nq = 0; a.R = 1; a.kreq = j.k; a.squared = j.squared ? 1 : 0;
    a.qlist2 = nullptr; a.qcount2_dev = nullptr; a.R2 = 0; a.row_out = j.row_out ? 1 : 0;
    a.out_d = j.out_d; a.out_i = j.out_i;
    a.unresolved = nullptr; a.n_unresolved = nullptr; a.ties = nullptr; a.n_ties = nullptr; a.ubound = nullptr; a.qbound2 = nullptr;
    a.skew_limit = 0.f; a.skew_far = 3.0e38f; a.skew_lo = 0.f; a.skew_flag = j.sc.counters + C_SKEW;      // only the first whole-cloud pass checks the balance
    a.qgp = j.qidx.gp;
    // unbalanced clouds (finer sub-box levels exist): a lane next to a heavy cell would scan thousands of candidates serially and hold its
    // wave for hundreds of microseconds (10 % cluster cloud: the base-grid lane pass took 380 us for 0.9M queries) -- hand such queries to
    // the wave-per-query pass, which scans heavy rows with 64 lanes
    a.lane_max_cand = j.n_fine > 0 ? (unsigned)std::max(384.0, 6.0 * 27.0 * j.occ) : (unsigned)std::max(4096.0, 64.0 * 27.0 * j.occ);
    a.fuse = j.fuse; a.f_sum = j.f_sum; a.f_max_v = j.f_max_v; a.f_max_k = j.f_max_k;
    a.f_limbs = j.f_limbs; a.f_special = j.f_special; a.f_wave_v = j.f_wave_v; a.f_wave_k = j.f_wave_k; a.f_accum = 0;
    a.bad_r = j.bad_r; a.bad_q = j.bad_q; a.escalate = 0;
    a.cancel_word = g_cancel_mirror.load(std::memory_order_relaxed); a.cancel_gen = t_call_gen;
    a.brick = j.brick ? 1 : 0; a.n_fallback = j.sc.counters + C_SPARE; a.maxval = j.maxval ? 1 : 0;
    return a;
}

// Enqueue (no host sync): lane-per-query pass at R=1 over all queries, then ONE wave-per-query launch fed by two
// device-side lists: possible ties (radius 1, total order) and stragglers (radius 2). What is still uncertified after
// that (list u2; next to nothing on balanced clouds) is finished by search_finish's host-driven loop.
// what: 1 = the lane passes only, 2 = only the wave-per-query launch that follows them, 3 = both (wave-only jobs: always everything)
template <typename T>
static bool lazy_wave_job(const SearchJob<T>& j);
template <typename T>
static int search_enqueue(pcu_hip_ctx* c, hipStream_t s, const SearchJob<T>& j, pcu_hip_stats* st, bool zero_counters = true, int what = 3) {
    const SearchScratch<T>& sc = j.sc;
    const int KF = pow2_at_least(j.k), KL = std::max(2, pow2_at_least(j.k + 1));
    if (zero_counters) HIP_TRY(hipMemsetAsync(sc.counters, 0, C_N * sizeof(int), s));
    SearchArgs<T> b = base_args(j, j.ridx);
    b.ties = sc.tt; b.n_ties = sc.counters + C_TT;
    // Wave-per-query from the start when lane-per-query would leave the GPU empty (few queries: a 2,885-vertex mesh
    // against 1M samples is 46 waves of long serial scans) or when k exceeds the register top-k of the lane kernel.
    const bool wave_only = j.k > kMaxKLane || j.qidx.n < kWaveOnlyBelow;
    if (!wave_only) {
        // Lane-per-query passes, finest dataset grid first: a query is served by the finest grid that certifies it
        // (dense regions), the rest falls through to the coarser grids (sparse regions) and finally to `ridx`.
        const int* lst = nullptr; const int* cnt = nullptr;
        for (int lv = 0; lv <= ((what & 1) ? j.n_fine : -1); ++lv) {
            const bool last = lv == j.n_fine;
            SearchArgs<T> a = base_args(j, last ? j.ridx : j.fine[lv]);
            a.qlist = lst; a.qcount_dev = cnt; a.nq = j.qidx.n; a.R = 1;
            a.unresolved = last ? sc.u1 : (lv == 0 ? sc.x0 : sc.x1);
            a.n_unresolved = sc.counters + (last ? C_U1 : (lv == 0 ? C_X0 : C_X1));
            if (!last) {
                // a closed level serves only the queries inside its box: split the incoming list first (search.h: k_box_split), the others
                // go straight to the next level's list
                HIP_TRY(hipMemsetAsync(sc.counters + C_U3, 0, sizeof(int), s));
                hipLaunchKernelGGL(k_box_split<T>, dim3((j.qidx.n + kSplitThreads * kSplitPer - 1) / (kSplitThreads * kSplitPer)), dim3(kSplitThreads), 0, s,
                                   j.qidx.sorted, lst, cnt, j.qidx.n, j.fine[lv].gp, sc.u3, sc.counters + C_U3, a.unresolved, a.n_unresolved);
                a.qlist = sc.u3; a.qcount_dev = sc.counters + C_U3;
            }
            if (last) a.ubound = sc.ub1;
            a.ties = sc.t1; a.n_ties = sc.counters + C_T1;
            if (!last) { a.ties = j.fine_ties[lv]; a.n_ties = sc.counters + C_TF0 + lv; }
            // balance limit: mean number of cell mates (sumsq / n) above kSkewFactor x the Poisson value (occupancy + 1)
            if (last && j.skew_check && j.n_fine == 0) { a.skew_limit = (float)(j.skew_hi * (j.occ + 1.0) * (double)j.ridx.n); a.skew_far = (float)(kSkewFactor * (j.occ + 1.0) * (double)j.ridx.n); a.skew_lo = (float)(j.skew_lo * (j.occ + 1.0) * (double)j.ridx.n); }
            const bool time_it = st && c->time_kernels && lv == 0 && c->n_kev + 2 <= 8;
            if (time_it) (void)hipEventRecord(c->kev[c->n_kev], s);
            if (launch_search_fast<T>(KF, a, j.qidx.n, s, /*open_index=*/last)) return -1;
            if (time_it) { (void)hipEventRecord(c->kev[c->n_kev + 1], s); c->n_kev += 2; }
            lst = a.unresolved; cnt = a.n_unresolved;
            if (!last) {
                // Possible ties of a sub-box level are put into total order ON that level: the lane pass certified them there, so every
                // candidate of equal distance lies in the 27 cells it scanned. (A closed level's wave pass does not escalate: what it cannot
                // certify at radius 1 goes to u2 and takes the host-driven passes of search_finish on the base grid, one host round trip each --
                // rare: a certified lane's ties sit inside its 27 cells.) (Handed to the base grid -- round 2 -- each of them, a query
                // inside the cluster, scanned the base cell that holds the whole cluster: 1400 queries x 100k points = 0.45 ms per direction
                // on the tight-cluster Chamfer.) What this launch cannot certify falls to the host-driven passes like any straggler.
                SearchArgs<T> w = base_args(j, j.fine[lv]);
                w.ties = sc.tt; w.n_ties = sc.counters + C_TT;
                w.qlist = j.fine_ties[lv]; w.qcount_dev = sc.counters + C_TF0 + lv; w.R = 1;
                w.unresolved = sc.u2; w.n_unresolved = sc.counters + C_U2;
                if (launch_search_wave<T>(KL, w, s)) return -1;
            }
        }
        b.qlist = sc.t1; b.qcount_dev = sc.counters + C_T1; b.R = 1;             // possible ties -> total order, radius 1
        b.qlist2 = sc.u1; b.qcount2_dev = sc.counters + C_U1; b.R2 = 2;          // stragglers, radius 2
        b.qbound2 = sc.ub1;
        b.unresolved = sc.u2; b.n_unresolved = sc.counters + C_U2;
        b.escalate = wave_escalates();                                           // ... and whatever it takes after that, inside the launch
        if ((what & 2) && launch_search_wave<T>(KL, b, s)) return -1;
        if (st) st->n_passes += ((what & 1) ? 1 + j.n_fine : 0) + ((what & 2) ? 1 : 0);
    } else {
        if (j.fuse) return fail(PCU_HIP_ERR_RUNTIME, "internal: fused epilogue on a wave-only search");
        b.qlist = nullptr; b.qcount_dev = nullptr; b.nq = j.qidx.n; b.R = 1;     // every query, radius 1, total order
        b.unresolved = sc.u1; b.n_unresolved = sc.counters + C_U1;
        if (j.skew_check) { b.skew_limit = (float)(j.skew_hi * (j.occ + 1.0) * (double)j.ridx.n); b.skew_far = (float)(kSkewFactor * (j.occ + 1.0) * (double)j.ridx.n); b.skew_lo = (float)(j.skew_lo * (j.occ + 1.0) * (double)j.ridx.n); }
        if (launch_search_wave<T>(KL, b, s)) return -1;
        b.nq = 0; b.skew_limit = 0.f; b.skew_lo = 0.f;
        b.qlist = sc.u1; b.qcount_dev = sc.counters + C_U1; b.R = 2;             // stragglers, radius 2
        b.unresolved = sc.u2; b.n_unresolved = sc.counters + C_U2;
        b.escalate = wave_escalates();
        if (launch_search_wave<T>(KL, b, s)) return -1;
        if (st) st->n_passes += 2;
    }
    return 0;
}

// Both directions of a two-sided call (k = 1): ONE lane-per-query launch and ONE wave-per-query launch serve both
// (each direction keeps its own lists and counters). Falls back to two search_enqueue calls when a direction does not
// take the k = 1 lane kernel (few queries, tile / generic kernels selected by environment).
template <typename T>
static bool lane_k1_job(const SearchJob<T>& j) { return j.k == 1 && j.qidx.n >= kWaveOnlyBelow && j.n_fine == 0 && use_k1_kernel(); }
// what == 1: only the lane pass; what == 2: only the wave pass; 3: both
// A k = 1 lane job on an open index finishes its stragglers inside the lane launch (search.h: radius 2 by the query's own wave): its wave pass
// can wait until the counters say that something is left.
template <typename T>
static bool lazy_wave_job(const SearchJob<T>& j) { static const bool off = getenv("PCU_HIP_FUSED_WAVE") != nullptr; return !off && lane_k1_job(j); }
template <typename T>
static int search_enqueue_pair(pcu_hip_ctx* c, hipStream_t s, const SearchJob<T>& j0, const SearchJob<T>& j1, pcu_hip_stats* st, int what = 3) {
    if (!(lane_k1_job(j0) && lane_k1_job(j1))) {
        if (j0.fuse || j1.fuse) return fail(PCU_HIP_ERR_RUNTIME, "internal: fused epilogue without the paired k = 1 pass");
        // Two small clouds (wave-per-query from the start, see search_enqueue): both directions share each of the two launches -- such a
        // call is a chain of launch latencies (config 1, 10k-vs-10k: 4 wave launches of 46 + 9 + 43 + 10 us were 60 % of its GPU time).
        auto wave_only = [](const SearchJob<T>& j) { return (j.k > kMaxKLane || j.qidx.n < kWaveOnlyBelow) && j.n_fine == 0; };
        static const bool no_merge = getenv("PCU_HIP_NO_WAVE_MERGE") != nullptr;
        if (!no_merge && what == 3 && wave_only(j0) && wave_only(j1) && j0.k == j1.k) {
            const int KL = std::max(2, pow2_at_least(j0.k + 1));
            SearchArgs<T> b[2];
            for (int d = 0; d < 2; ++d) {
                const SearchJob<T>& j = d ? j1 : j0;
                b[d] = base_args(j, j.ridx);
                b[d].ties = j.sc.tt; b[d].n_ties = j.sc.counters + C_TT;
                b[d].qlist = nullptr; b[d].qcount_dev = nullptr; b[d].nq = j.qidx.n; b[d].R = 1;      // every query, radius 1, total order
                b[d].unresolved = j.sc.u1; b[d].n_unresolved = j.sc.counters + C_U1;
                if (j.skew_check) { b[d].skew_limit = (float)(j.skew_hi * (j.occ + 1.0) * (double)j.ridx.n); b[d].skew_far = (float)(kSkewFactor * (j.occ + 1.0) * (double)j.ridx.n); b[d].skew_lo = (float)(j.skew_lo * (j.occ + 1.0) * (double)j.ridx.n); }
            }
            if (launch_search_wave<T>(KL, b[0], s, &b[1])) return -1;
            for (int d = 0; d < 2; ++d) {
                const SearchJob<T>& j = d ? j1 : j0;
                b[d].nq = 0; b[d].skew_limit = 0.f; b[d].skew_lo = 0.f;
                b[d].qlist = j.sc.u1; b[d].qcount_dev = j.sc.counters + C_U1; b[d].R = 2;             // stragglers, radius 2
                b[d].unresolved = j.sc.u2; b[d].n_unresolved = j.sc.counters + C_U2;
                b[d].escalate = wave_escalates();
            }
            if (launch_search_wave<T>(KL, b[0], s, &b[1])) return -1;
            if (st) st->n_passes += 4;
            return 0;
        }
        if (search_enqueue(c, s, j0, st, /*zero_counters=*/false)) return -1;
        return search_enqueue(c, s, j1, st, false);
    }
    SearchArgs<T> a[2], b[2];
    for (int d = 0; d < 2; ++d) {
        const SearchJob<T>& j = d ? j1 : j0;
        const SearchScratch<T>& sc = j.sc;
        a[d] = base_args(j, j.ridx);
        a[d].nq = j.qidx.n; a[d].R = 1;
        a[d].unresolved = sc.u1; a[d].n_unresolved = sc.counters + C_U1; a[d].ubound = sc.ub1;
        a[d].ties = sc.t1; a[d].n_ties = sc.counters + C_T1;
        if (j.skew_check) { a[d].skew_limit = (float)(j.skew_hi * (j.occ + 1.0) * (double)j.ridx.n); a[d].skew_far = (float)(kSkewFactor * (j.occ + 1.0) * (double)j.ridx.n); a[d].skew_lo = (float)(j.skew_lo * (j.occ + 1.0) * (double)j.ridx.n); }
        b[d] = base_args(j, j.ridx);
        b[d].ties = sc.tt; b[d].n_ties = sc.counters + C_TT;
        b[d].qlist = sc.t1; b[d].qcount_dev = sc.counters + C_T1; b[d].R = 1;            // possible ties -> total order, radius 1
        b[d].qlist2 = sc.u1; b[d].qcount2_dev = sc.counters + C_U1; b[d].R2 = 2;         // stragglers, radius 2
        b[d].qbound2 = sc.ub1;
        b[d].unresolved = sc.u2; b[d].n_unresolved = sc.counters + C_U2;
        b[d].escalate = wave_escalates();
    }
    if (what & 1) {
        const bool time_it = st && c->time_kernels && c->n_kev + 2 <= 8;
        if (time_it) (void)hipEventRecord(c->kev[c->n_kev], s);
        if (launch_search_fast<T>(1, a[0], j0.qidx.n, s, true, &a[1], j1.qidx.n)) return -1;
        if (time_it) { (void)hipEventRecord(c->kev[c->n_kev + 1], s); c->n_kev += 2; }
        if (st) st->n_passes += 2;
    }
    if (what & 2) {
        if (launch_search_wave<T>(2, b[0], s, &b[1])) return -1;
        if (st) st->n_passes += 2;
    }
    return 0;
}

// Exact ties: rebuild nanoflann's kd-tree on the GPU and re-run the tied queries through nanoflann's own
// traversal (kd_order.h). Only called when the grid search reported genuine ties -- or ahead of need, see KdSpec.
constexpr int kKdSpecPairs = 4;          // level pairs started early: the 8 top levels, which hold (nearly) all points whichever queries are tied
constexpr int kKdSpecMinPoints = 262144;
constexpr int kKdFinishGrid = 1024;       // workgroups of k_kd_finish (they stride over the level list)
constexpr int kKdSubGridRoi = 256;        // workgroups of k_kd_subtree when the list's length is not read back
static int kd_finish_max() {
    static const int v = getenv("PCU_HIP_KD_FINISH_MAX") ? atoi(getenv("PCU_HIP_KD_FINISH_MAX")) : 16384;
    return v;
}
template <typename T>
static hipError_t kd_subtree_attr() {
    static std::atomic<unsigned long long> attr_set;
    if (!attr_unset_here(attr_set)) return hipSuccess;
    return hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kd_subtree<T>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)kd_sub_lds_bytes<T>());
}
template <typename T>
static int kd_build_device(pcu_hip_ctx* c, Arena& ar, hipStream_t s, const T* d_pts, int M, const GridParams<T>* gp, int leaf_max,
                           KdBuild<T>& b, int** err_out, int* levels_out, int* n_real_out, const SearchJob<T>* roi_job = nullptr, int n_tied = 0,
                           bool speculative = false, bool need_depth = true, int** max_depth_out = nullptr) {
    // speculative: only the first kKdSpecPairs levels, on c->spec_stream behind the event c->kd_spec.ev_fork (recorded by the caller
    // on its stream once gp is final), no host synchronisation; a later normal call for the same input continues from there.
    pcu_hip_ctx::KdSpec& sp = c->kd_spec;
    if (speculative) s = c->spec_stream;
    b.leaf_max = leaf_max;
    b.sub_max = (int)std::min<long long>(KdSub<T>::S, (long long)KdSub<T>::CAP * (leaf_max + 1));
    // level lists only ever hold nodes with more than sub_max elements
    const size_t max_level = (size_t)M / (size_t)(b.sub_max + 1) + 4;
    // node ids: < 2*max_level for the top levels, plus 2n reserved by each LDS sub-tree block (sum of n <= M)
    const size_t max_nodes = 2 * (size_t)M + 4 * max_level + 16;
    const size_t max_items = (size_t)M / kKdChunk + 3 * max_level + 2;      // (+ stubs: up to two per level node, one chunk each or more)
    const size_t n_sub_cap = 2 * max_level + (size_t)M / (size_t)(b.sub_max / 2 + 1) + 16;
    int* counters = nullptr;
    (void)ar;
    {
        auto al = [](size_t v) { return align_up(v, 256); };
        const size_t need = al((size_t)M * sizeof(Pt4<T>)) + al(max_nodes * sizeof(KdNode<T>)) + al(64) + 2 * al(3 * max_level * 4) +
                            2 * al((3 * max_level + 1) * 4) + 2 * al(max_items * 4) + 2 * al((size_t)M * 4) + al(n_sub_cap * 4) + al(16 * 8) + 8192;
        if (kd_ws_reserve(c, need)) return -1;
    }
    KdArena ka{c};
    if (ka.get(&b.E, (size_t)M) || ka.get(&b.nodes, max_nodes) || ka.get(&counters, 16)) return -1;
    if (ka.get(&b.level_nodes, 3 * max_level) || ka.get(&b.next_nodes, 3 * max_level)) return -1;
    if (ka.get(&b.level_cbase, 3 * max_level + 1) || ka.get(&b.next_cbase, 3 * max_level + 1)) return -1;
    if (ka.get(&b.chunk_bl, max_items) || ka.get(&b.chunk_br, max_items)) return -1;
    if (ka.get(&b.BLpos, (size_t)M) || ka.get(&b.BRpos, (size_t)M)) return -1;
    if (ka.get(&b.sub_nodes, n_sub_cap)) return -1;
    T* roi = nullptr; int* n_roi = nullptr;
    if (ka.get(&roi, 4 * kKdMaxRoi) || ka.get(&n_roi, 16)) return -1;
    b.roi = roi; b.n_roi = n_roi;
    b.n_nodes = counters; b.n_items = counters + 2;
    *err_out = counters + 3;
    b.n_sub = counters + 4; b.max_depth = counters + 5; b.n_real = counters + 6;
    b.n_cur = counters + 7; b.n_next = counters + 8; b.need_ph2 = counters + 9; b.level_ph2 = counters + 10;
    b.prof = nullptr;
    if (getenv("PCU_HIP_PROF_KD")) { if (ka.get(&b.prof, 16)) return -1; HIP_TRY(hipMemsetAsync(b.prof, 0, 16 * sizeof(long long), s)); }
    // planeSplit's second loop (elements EQUAL to the cut value) has no work on generic data, and its three launches per
    // level are pure launch floor: the level graph is first replayed without them; if some node turns out to hold such
    // elements (device flag), the build is redone with them and the context remembers (duplicated points, lattices).
    int hcnt[16] = {0};
    bool sub_launched = false;          // the LDS sub-trees were enqueued without a read-back of their count
    // whoever uses the workspace next waits for an earlier speculative prefix (adopted or not)
    if (sp.pending) { HIP_TRY(hipStreamWaitEvent(s, sp.ev_done, 0)); sp.pending = false; }
    bool adopt = !speculative && sp.active && sp.pts == (const void*)d_pts && sp.gp == (const void*)gp && sp.m == M && sp.leaf == leaf_max &&
                 sp.with_ph2 == c->kd_need_ph2 && M > b.sub_max;
    if (!speculative) sp.active = false;
    for (int rebuild = 0; rebuild < 2; ++rebuild) {
    const bool with_ph2 = c->kd_need_ph2;
    int levels_done = 0;
    if (adopt && rebuild == 0) {
        // the top levels are there (built without regions of interest: complete); the regions apply from here on
        levels_done = sp.levels_done;
        static const bool no_roi = getenv("PCU_HIP_KD_FULL") != nullptr;
        if (roi_job && n_tied > 0 && n_tied <= kKdMaxRoi && !no_roi)
            hipLaunchKernelGGL(k_kd_roi<T>, dim3(1), dim3(kKdMaxRoi), 0, s, roi_job->qidx.sorted, roi_job->sc.tt, n_tied, roi_job->out_d, roi_job->k,
                               roi_job->squared ? 1 : 0, roi_job->row_out ? 1 : 0, roi, n_roi);
    } else {
    if (speculative) HIP_TRY(hipStreamWaitEvent(s, sp.ev_fork, 0));
    HIP_TRY(hipMemsetAsync(counters, 0, 16 * sizeof(int), s));
    // few tied queries: build only the part of the tree their traversals can touch (kd_order.h, KdBuild::roi)
    static const bool no_roi = getenv("PCU_HIP_KD_FULL") != nullptr;
    if (!speculative && roi_job && n_tied > 0 && n_tied <= kKdMaxRoi && !no_roi)
        hipLaunchKernelGGL(k_kd_roi<T>, dim3(1), dim3(kKdMaxRoi), 0, s, roi_job->qidx.sorted, roi_job->sc.tt, n_tied, roi_job->out_d, roi_job->k,
                           roi_job->squared ? 1 : 0, roi_job->row_out ? 1 : 0, roi, n_roi);
    else HIP_TRY(hipMemsetAsync(n_roi, 0, sizeof(int), s));
    hipLaunchKernelGGL(k_kd_init_elems<T>, dim3((M + kBlock - 1) / kBlock), dim3(kBlock), 0, s, d_pts, M, b.E);
    hipLaunchKernelGGL(k_kd_root<T>, dim3(1), dim3(64), 0, s, b, gp, M);
    if (speculative) HIP_TRY(hipEventRecord(sp.ev_init, s));          // the caller's buffers (points, grid parameters) are not read after this
    }
    // Level passes only while the nodes are large: below kd_finish_max() elements one workgroup per node finishes the rest
    // (k_kd_finish). 0 = level passes all the way down to the LDS sub-trees (the round-3 flow).
    const int fin_max = kd_finish_max();
    int levels_sync = 0;
    if (fin_max > 0) for (long long m = M; m > fin_max; m = (m + 1) >> 1) ++levels_sync;
    // grid of a level pass: a level above levels_sync has at most 2^level nodes, i.e. M / chunk full work items + one partial per node;
    // the general bound (max_items: every later level of the round-3 flow) is 5x that at 4M points, and blocks without a work item
    // still wait for a slot beside the searches
    const int items_ub = fin_max > 0 ? (int)std::min<size_t>(max_items, (size_t)M / kKdChunk + ((size_t)1 << std::min(levels_sync, 24)) + 2) : (int)max_items;
    // One level = 9 short launches; the count of nodes per level lives on the device and kernels of an exhausted
    // level exit at once, so levels need no host decision. Two consecutive levels (the ping-pong of the level
    // lists has period 2) are captured ONCE into a hipGraph and replayed: the expected log2(M / sub_max) + 2
    // levels first, then two at a time until the device reports an empty level. Replay removes most of the
    // per-launch host cost, which dominated this launch-bound phase.
    auto enqueue_one_level = [&](KdBuild<T>& bb, bool with_ph2) {           // (leaves bb with the roles of the two level lists exchanged)
        hipLaunchKernelGGL(k_kd_minmax<T>, dim3(items_ub), dim3(kBlock), 0, s, bb);
        hipLaunchKernelGGL(k_kd_count<T>, dim3(items_ub), dim3(kBlock), 0, s, bb);
        for (int ph = 0; ph < (with_ph2 ? 2 : 1); ++ph) {
            if (ph == 1) hipLaunchKernelGGL(k_kd_bad_count<T>, dim3(items_ub), dim3(kBlock), 0, s, bb, ph);      // (loop 1 ranks from the count pass: k_kd_lists)
            hipLaunchKernelGGL(k_kd_lists<T>, dim3(items_ub), dim3(kBlock), 0, s, bb, ph);
            hipLaunchKernelGGL(k_kd_swap<T>, dim3(items_ub), dim3(kBlock), 0, s, bb, ph);
        }
        hipLaunchKernelGGL(k_kd_advance<T>, dim3(1), dim3(kBlock), 0, s, bb);
        std::swap(bb.level_nodes, bb.next_nodes);
        std::swap(bb.level_cbase, bb.next_cbase);
        std::swap(bb.n_cur, bb.n_next);
    };
    auto enqueue_level_pair = [&](KdBuild<T> bb, bool with_ph2) { enqueue_one_level(bb, with_ph2); enqueue_one_level(bb, with_ph2); };
    int expected = 2;
    for (long long m = M; m > b.sub_max; m >>= 1) ++expected;
    const bool roi_mode = !speculative && roi_job && n_tied > 0 && n_tied <= kKdMaxRoi && getenv("PCU_HIP_KD_FULL") == nullptr;
    sub_launched = false;
    if (M > b.sub_max) {
        auto& G = c->kd_graph[(sizeof(T) == 4 ? 0 : 2) + (with_ph2 ? 1 : 0)];
        const bool use_graph = getenv("PCU_HIP_NO_GRAPH") == nullptr && s != nullptr;      // (the legacy NULL stream cannot be captured: eager launches there)
        if (use_graph && (!G.exec || G.key_ptr != (const void*)b.E || G.key_m != M || G.key_leaf != leaf_max)) {
            if (G.exec) { (void)hipGraphExecDestroy(G.exec); G.exec = nullptr; }
            if (G.graph) { (void)hipGraphDestroy(G.graph); G.graph = nullptr; }
            HIP_TRY(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
            enqueue_level_pair(b, with_ph2);
            HIP_TRY(hipStreamEndCapture(s, &G.graph));
            HIP_TRY(hipGraphInstantiate(&G.exec, G.graph, nullptr, nullptr, 0));
            G.key_ptr = (const void*)b.E; G.key_m = M; G.key_leaf = leaf_max;
        }
        // levels [from, to): pairs replay the graph (after an even number of levels the two level lists have their original roles), an
        // odd level is enqueued directly; `cur` ends as the build with the roles the next consumer must see
        KdBuild<T> cur = b;
        auto run_levels = [&](int from, int to) -> int {
            if (from & 1) { std::swap(cur.level_nodes, cur.next_nodes); std::swap(cur.level_cbase, cur.next_cbase); std::swap(cur.n_cur, cur.n_next); }
            for (int done = from; done < to;) {
                if (!(done & 1) && to - done >= 2) { if (use_graph) HIP_TRY(hipGraphLaunch(G.exec, s)); else enqueue_level_pair(cur, with_ph2); done += 2; }
                else { enqueue_one_level(cur, with_ph2); done += 1; }
            }
            return 0;
        };
        if (speculative) {
            const int nl = fin_max > 0 ? std::min(2 * kKdSpecPairs, levels_sync) : 2 * std::min(kKdSpecPairs, (expected + 1) / 2);
            if (run_levels(0, nl)) return -1;
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipEventRecord(sp.ev_done, s));
            sp.active = true; sp.pending = true; sp.pts = (const void*)d_pts; sp.gp = (const void*)gp; sp.m = M; sp.leaf = leaf_max;
            sp.with_ph2 = with_ph2; sp.levels_done = nl;
            return 0;
        }
        if (fin_max > 0) {
            if (run_levels(levels_done, std::max(levels_done, levels_sync))) return -1;
            hipLaunchKernelGGL(k_kd_finish<T>, dim3(kKdFinishGrid), dim3(kFinThreads), 0, s, cur);
            HIP_TRY(hipGetLastError());
            if (roi_mode) {
                // few nodes are alive: the LDS sub-trees follow without a host read-back of their count (a fixed grid strides over the
                // list); a top level that met elements equal to its cut value without the second planeSplit loop is reported through
                // the counters the caller reads with the traversal's error flag (tie_order_resolve)
                HIP_TRY(kd_subtree_attr<T>());
                hipLaunchKernelGGL(k_kd_subtree<T>, dim3(kKdSubGridRoi), dim3(kSubThreads), kd_sub_lds_bytes<T>(), s, b);
                HIP_TRY(hipGetLastError());
                sub_launched = true;
            } else {
                HIP_TRY(hipMemcpyAsync(hcnt, counters, sizeof hcnt, hipMemcpyDeviceToHost, s));
                HIP_WAIT(s);
            }
        } else {
        int pairs = std::max(1, (expected + 1) / 2 - levels_done / 2);
        for (int guard = 0; guard < 100000; ++guard) {
            for (int i = 0; i < pairs; ++i) {
                if (use_graph) HIP_TRY(hipGraphLaunch(G.exec, s)); else enqueue_level_pair(b, with_ph2);
            }
            HIP_TRY(hipGetLastError());
            HIP_TRY(hipMemcpyAsync(hcnt, counters, sizeof hcnt, hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
            if (hcnt[b.n_cur - counters] == 0) break;      // after an even number of levels the roles are as at the start
            pairs = 1;
        }
        }
    }
    if (speculative) return 0;          // (a tree small enough for the LDS sub-tree kernel alone: nothing to start early)
    if (!with_ph2 && hcnt[9] && !sub_launched) { c->kd_need_ph2 = true; continue; }
    break;
    }
    (void)c;
    // finish every small node inside one workgroup's LDS (the level loop's last read-back already holds the count)
    if (!(M > b.sub_max)) {
        HIP_TRY(hipMemcpyAsync(hcnt, counters, sizeof hcnt, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
    }
    const int n_sub = sub_launched ? 0 : hcnt[4];
    if (n_sub > 0) {
        HIP_TRY(kd_subtree_attr<T>());
        hipLaunchKernelGGL(k_kd_subtree<T>, dim3(n_sub), dim3(kSubThreads), kd_sub_lds_bytes<T>(), s, b);
        HIP_TRY(hipGetLastError());
    }
    if (max_depth_out) *max_depth_out = counters + 5;
    if (!need_depth && !b.prof && !n_real_out) { *levels_out = 0; return 0; }      // (the caller sizes its stacks by a bound: one host round trip less)
    HIP_TRY(hipMemcpyAsync(hcnt, counters, sizeof hcnt, hipMemcpyDeviceToHost, s));
    HIP_WAIT(s);
    *levels_out = hcnt[5] + 1;      // tree depth (root = 0) + 1
    if (b.prof) {
        long long hp[16]; HIP_TRY(hipMemcpy(hp, b.prof, sizeof hp, hipMemcpyDeviceToHost));
        fprintf(stderr, "[kd prof] n_sub=%d nodes=%d depth=%d | ticks(100MHz): load %lld S1 %lld S2 %lld S3/4 %lld S5 %lld S6 %lld S7 %lld store %lld\n", n_sub, hcnt[6], hcnt[5], hp[0], hp[1], hp[2], hp[3], hp[4], hp[5], hp[6], hp[7]);
    }
    if (n_real_out) *n_real_out = hcnt[6];
    return 0;
}

// Start the tree's top levels ahead of need (see pcu_hip_ctx::KdSpec): called between a call's index build and its searches when the
// context's previous large call had genuine ties. Costs GPU cycles, not wall time, when this call has none.
// Two steps: the fork event is recorded on the caller's stream right after the index build, the tree's kernels are enqueued (on the
// second stream, behind that event) only AFTER the call's searches: enqueued first they took the chip for themselves -- the lane pass of
// a 4M-point call started 1.1 ms late (profiles/r03_c3_timeline.txt) -- and co-residency is impossible anyway (the k = 16 lane kernel
// fills the register file at 4 waves per SIMD). Now the searches start at once and the tree fills their tail and the wave pass.
template <typename T>
static bool kd_speculate_fork(pcu_hip_ctx* c, hipStream_t s, const SearchJob<T>& j) {
    static const bool off = getenv("PCU_HIP_NO_KD_SPEC") != nullptr;
    if (off || !c->kd_spec_hint || !j.tie_order || j.ridx.n < kKdSpecMinPoints) return false;
    pcu_hip_ctx::KdSpec& sp = c->kd_spec;
    if (!sp.ev_fork) {
        if (hipEventCreateWithFlags(&sp.ev_fork, hipEventDisableTiming) != hipSuccess || hipEventCreateWithFlags(&sp.ev_init, hipEventDisableTiming) != hipSuccess ||
            hipEventCreateWithFlags(&sp.ev_done, hipEventDisableTiming) != hipSuccess) return false;
    }
    return hipEventRecord(sp.ev_fork, s) == hipSuccess;             // the dataset's grid parameters (bbox = the root's) are final
}
template <typename T>
static int kd_speculate(pcu_hip_ctx* c, Arena& ar, hipStream_t s, const SearchJob<T>& j) {
    KdBuild<T> b; int* err = nullptr; int levels = 0;
    return kd_build_device(c, ar, s, j.d_ref_pts, j.ridx.n, j.ridx.gp, j.leaf_max, b, &err, &levels, nullptr, (const SearchJob<T>*)nullptr, 0, /*speculative=*/true);
}
// ... and, at the end of that call: the early kernels read the caller's points and the call's grid parameters -- both may be recycled once
// the call returns -- so wait for them (long done by then); a prefix nobody adopted means the prediction was wrong.
static void kd_speculate_end(pcu_hip_ctx* c, bool call_completed) {
    pcu_hip_ctx::KdSpec& sp = c->kd_spec;
    if (!sp.pending && !sp.active) return;
    if (sp.ev_init) (void)hipEventSynchronize(sp.ev_init);
    if (sp.active) { sp.active = false; if (call_completed) c->kd_spec_hint = false; }
}

// The tie-order traversal of n queries (one wave each). A call whose queries are nearly all tied -- a cloud against a line, a lattice -- runs
// hundreds of milliseconds here, and a kernel cannot be abandoned: beyond kKdSearchPiece queries the launch is enqueued in pieces, at most two
// in the queue, the host waiting (cancellably) for piece i - 1 before it enqueues piece i + 1. The GPU never idles between pieces; a
// cancellation request is honoured within about one piece (~30 ms of traversal).
constexpr int kKdSearchPiece = 16384;
template <typename T>
static int kd_search_launch(pcu_hip_ctx* c, hipStream_t s, KdSearchArgs<T> a, int n) {
    if (n <= kKdSearchPiece) {
        a.t0 = 0;
        hipLaunchKernelGGL(k_kd_search<T>, dim3(n), dim3(64), 0, s, a);
        HIP_TRY(hipGetLastError());
        return 0;
    }
    for (auto& e : c->cev) if (!e) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (int t0 = 0, i = 0; t0 < n; t0 += kKdSearchPiece, ++i) {
        a.t0 = t0;
        hipLaunchKernelGGL(k_kd_search<T>, dim3(std::min(kKdSearchPiece, n - t0)), dim3(64), 0, s, a);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipEventRecord(c->cev[i & 1], s));
        if (i >= 1) { if (int w = wait_event(c->cev[(i - 1) & 1], s)) return w; }
    }
    return 0;
}

template <typename T>
static int tie_order_resolve(pcu_hip_ctx* c, Arena& ar, hipStream_t s, SearchJob<T>& j, int n_tt, pcu_hip_stats* st) {
    c->kd_spec_hint = j.ridx.n >= kKdSpecMinPoints;          // genuine ties: calls like this one will probably have them again
    hipEvent_t e0 = c->ev[4], e1 = c->ev[5];
    const bool timed = st && c->time_phases;
    if (timed) (void)hipEventRecord(e0, s);
    // attempt 0: tree restricted to the tied queries' regions of interest (few ties); attempt 1: the whole tree, if a
    // traversal wanted to walk into a big unbuilt part (rare; results are exact either way, this is about time)
    for (int attempt = 0; attempt < 2; ++attempt) {
        KdBuild<T> b; int* err = nullptr; int levels = 0; int* depth_dev = nullptr;
        // few tied queries: their traversal stacks are sized by a bound on the tree depth instead of waiting for the exact one
        // (a deeper tree raises the traversal's error flag 1: the traversal alone is then repeated with the exact depth)
        constexpr int kDepthBound = 96;
        const bool lazy = n_tt <= 4096;
        if (kd_build_device(c, ar, s, j.d_ref_pts, j.ridx.n, j.ridx.gp, j.leaf_max, b, &err, &levels, nullptr, attempt == 0 ? &j : nullptr, n_tt,
                            false, /*need_depth=*/!lazy, &depth_dev)) return -1;
        if (lazy) levels = kDepthBound;
        KdSearchArgs<T> a;
        a.E = b.E; a.nodes = b.nodes; a.qsorted = j.qidx.sorted; a.qlist = j.sc.tt; a.qcount_dev = j.sc.counters + C_TT;
        a.k = j.k; a.squared = j.squared ? 1 : 0; a.row_out = j.row_out ? 1 : 0; a.out_d = j.out_d; a.out_i = j.out_i; a.error_flag = err;
        a.qraw = nullptr; a.nq_raw = 0; a.rs_d = nullptr; a.rs_i = nullptr;
        a.cancel_word = g_cancel_mirror.load(std::memory_order_relaxed); a.cancel_gen = t_call_gen;
        KdFrame<T>* frames = nullptr;
        a.stack_cap = levels + 2;
        if (aalloc(ar, &frames, (size_t)n_tt * a.stack_cap)) return -1;
        a.stack = frames;
        if (kd_search_launch(c, s, a, n_tt)) return -1;
        int hc[16] = {0};                  // the build's counters: [3] = the traversal's error flag, [9] = a node held elements equal to its cut value
        HIP_TRY(hipMemcpyAsync(hc, err - 3, sizeof hc, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        if (hc[9] && !c->kd_need_ph2) {     // top levels ran without planeSplit's second loop and needed it (kd_build_device, regions of
            c->kd_need_ph2 = true;          // interest: no read-back before the traversal): once more, the context remembers
            --attempt; continue;
        }
        int herr = hc[3];
        if (herr == 1 && lazy) {            // deeper than the bound: once more with the exact depth
            int depth = 0;
            HIP_TRY(hipMemcpy(&depth, depth_dev, sizeof(int), hipMemcpyDeviceToHost));
            HIP_TRY(hipMemsetAsync(err, 0, sizeof(int), s));
            a.stack_cap = depth + 3;
            if (aalloc(ar, &frames, (size_t)n_tt * a.stack_cap)) return -1;
            a.stack = frames;
            if (kd_search_launch(c, s, a, n_tt)) return -1;
            HIP_TRY(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
        }
        if (herr == 2 && attempt == 0) continue;
        if (herr) return fail(PCU_HIP_ERR_RUNTIME, "internal: kd tie-order traversal exceeded the tree depth (%d)", levels);
        break;
    }
    if (timed) { (void)hipEventRecord(e1, s); HIP_WAIT(s); float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1); st->ms_tie += ms; }
    return 0;
}

// Thresholds of a whole-call job (not for persistent indexes): see kRescaleAbove.
template <typename T>
static void job_rescale_setup(const pcu_hip_ctx* c, SearchJob<T>& j, bool allowed, int role) {
    j.may_rescale = allowed && rescale_enabled(c); j.role = role;
    if (!j.may_rescale) return;
    if (c->occ_scale[role] >= 1.0) j.skew_hi = kRescaleAbove; else j.skew_lo = kRescaleBelow;
}
// A pass gave up on the balance check: is it a case for a different grid resolution (then c->occ_scale is updated and the caller
// restarts the call) or for the refit machinery (false)?
template <typename T>
static bool rescale_wanted(pcu_hip_ctx* c, const SearchJob<T>& j, hipStream_t s, int skew_flag) {
    if (!j.may_rescale) return false;
    // flag value 2 (search.h, skew_far): unbalanced beyond kSkewFactor -- a case for the refit path whatever the exact ratio is
    if (skew_flag == 2 && c->occ_scale[j.role] >= 1.0) return false;
    GridParams<T> hg;
    if (hipMemcpyAsync(&hg, j.ridx.gp, sizeof hg, hipMemcpyDeviceToHost, s) != hipSuccess || hipStreamSynchronize(s) != hipSuccess) return false;
    const double r = (double)hg.sumsq / (double)j.ridx.n / (j.occ + 1.0);
    double& scale = c->occ_scale[j.role];
    if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[rescale] cloud %d n=%d occ=%.3f scale=%.3f ratio=%.2f\n", j.role, j.ridx.n, j.occ, scale, r);
    if (scale >= 1.0) {
        if (r > kRescaleAbove && r <= kSkewFactor) { scale = 1.0 / std::min(kRescaleMax, sqrt(r)); return true; }
        return false;
    }
    // a finer grid kept from earlier calls: back to the default when this cloud is even at that scale, or so uneven that it is a
    // case for the refit path (which works from the default resolution)
    if (r < kRescaleBelow || r > kSkewFactor) { scale = 1.0; return true; }
    return false;
}

// Non-finite input an operator does not take (grid.h: kNf*; search.h: index_not_ready): a ValueError on the Python side.
// What the reference answers stably is answered the same way: rows of a QUERY / SOURCE cloud with a non-finite
// coordinate find nothing (-1 / -1.0, src/point_cloud_distance.cpp:90-93) and take part in the metrics as the reference's tails make
// them (a -1.0 never wins Hausdorff's max, :223; Chamfer gathers through index -1 = numpy's last row, __init__.py:112-113); rows of a
// DATASET / TARGET cloud with infinities of one sign per axis are never anybody's neighbour. Only a cloud that is searched IN and
// holds a NaN, or +inf and -inf along one axis, has no stable answer in the reference (its kd-tree bounds become NaN).
constexpr int kNfHard = kNfNaN | kNfBothInf;
constexpr int PCU_NONFINITE = -1000;    // internal: a search met non-finite input its operator rejects (the caller words the error)
static int nonfinite_error(bool metric) {
    if (metric) return fail(PCU_HIP_ERR_INVALID, "Invalid input: a point cloud that is searched in (the target; both clouds of hausdorff_distance / chamfer_distance) "
                            "contains non-finite coordinates: NaN, or both +inf and -inf along one axis. The reference builds its kd-tree over NaN bounds for such data "
                            "and its result depends on the traversal; not supported (non-finite source rows and single-signed infinities are handled as the reference handles them).");
    return fail(PCU_HIP_ERR_INVALID, "Invalid input: dataset_points contains NaN coordinates, or both +inf and -inf along one axis. The reference builds its kd-tree over "
                "NaN bounds for such data and returns rows that depend on the traversal; not supported (non-finite query points and single-signed "
                "infinities in the dataset are handled as the reference handles them).");
}
// After the call's single read-back + stream sync: look at the counters; finish whatever is still unresolved with coarser dataset grids
// (host-driven, one sync per pass; only far-away / isolated queries ever get here).
// Returns 1 if extra passes ran (callers then redo dependent reductions), 0 if not, 3 if the call is to be restarted (rescale_wanted),
// <0 on error.
// Two sub-box levels over the heavy cells (more than 8x the wanted occupancy) of `from`, and once more over what is still heavy in the
// first (tight clusters inside blobs); the passes then run finest level first.
template <typename T>
static int skew_add_levels(Arena& ar, hipStream_t s, SearchJob<T>& j, pcu_hip_stats* st, const GridIndex<T>& from, const double** hs_dev) {
    const unsigned thresh = (unsigned)(8.0 * j.occ + 8.0);
    GridIndex<T> sub1, sub2;
    if (index_build_heavy(ar, sub1, from, j.d_ref_pts, j.occ, thresh, s, hs_dev)) return -1;
    if (index_build_heavy(ar, sub2, sub1, j.d_ref_pts, j.occ, thresh, s)) return -1;
    if (st) st->n_grid_builds += 2;
    j.fine[0] = sub2; j.fine[1] = sub1; j.n_fine = 2;
    for (int lv = 0; lv < 2; ++lv) if (!j.fine_ties[lv] && aalloc(ar, &j.fine_ties[lv], (size_t)j.qidx.n)) return -1;
    return 0;
}
// A direction's refit attempt enqueued ahead of search_finish (skew_prelaunch: both directions of a two-sided call at once, on two
// streams), with what its read-backs delivered.
struct SkewPre {
    bool on = false;
    int hc_redo[C_N]; double hs[2] = {0, 0}; const double* hs_dev = nullptr;
    int d_passes = 0, d_builds = 0;             // what the attempt added to the call's statistics (taken back if it is dropped)
};
template <typename T>
static int search_finish(pcu_hip_ctx* c, Arena& ar, hipStream_t s, SearchJob<T>& j, pcu_hip_stats* st, const int* hc, SkewPre* pre = nullptr) {
    // hc: host copy of j.sc.counters, read back by the caller together with the call's scalar results
    // (one D2H copy + one stream sync for the whole call in the common case)
    int hc_redo[C_N], hc_large[C_N];
    bool redone = false;
    if (hc[C_LARGE] & 4) return PCU_NONFINITE;
    if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[finish] role=%d n=%d occ=%.3f may=%d skew=%d large=%d u1=%d u2=%d t1=%d\n", j.role, j.ridx.n, j.occ, (int)j.may_rescale, hc[C_SKEW], hc[C_LARGE], hc[C_U1], hc[C_U2], hc[C_T1]);
    if (hc[C_LARGE]) {
        // Every pass gave up at once because an index was not ready (GridParams::has_large).
        if (hc[C_LARGE] & 2) {
            // A one-pass build overflowed a bucket slot (uneven data): rebuild with the two-pass pipeline, which has k_bucket_large
            // for such buckets, and keep to it in this context.
            c->two_pass = true; c->eager_large = true;
            GridIndex<T>* ov[2] = {&j.qidx, &j.ridx};
            for (GridIndex<T>* g : ov)
                if (g->one_pass) { g->one_pass = false; if (index_build<T>(*g, g->src, g->occ_built, s)) return -1; if (st) st->n_grid_builds += 1; }
        }
        // Over-full buckets of a bucketed index were still unplaced (their placement, k_bucket_large, is only launched on demand).
        // Place them -- for the query cloud and the dataset -- and run the passes again.
        index_large_pass<T>(j.qidx, &j.ridx, s);
        c->eager_large = true;                   // data of this kind will come again (mesh samples: +0.18 ms per call for the round trip)
        if (search_enqueue(c, s, j, st)) return -1;
        HIP_TRY(hipMemcpyAsync(hc_large, j.sc.counters, sizeof hc_large, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        hc = hc_large; redone = true;
    }
    if (hc[C_SKEW] && rescale_wanted(c, j, s, hc[C_SKEW])) return 3;          // 3: restart the call at another grid resolution
    if (hc[C_SKEW]) {
        // The dataset grid is badly unbalanced (clusters, blobs, a far outlier inflating the bbox): every pass gave up
        // at once. Refit: same cell count over the core range of the cloud (replaces `ridx`), then up to two finer
        // grids sized by how unbalanced the previous one still is; the passes then run finest grid first.
        if (!(pre && pre->on))
        index_large_pass<T>(j.qidx, &j.ridx, s);        // (the balance check comes first in the kernels: unplaced over-full buckets
                                                        // of the query index would stop the passes below as well)
        // heavy cells (more than 8x the wanted occupancy): a sub-box grid over them, sized by how overfull they are;
        // and once more over what is still heavy in that one (tight clusters inside blobs)
        GridIndex<T> base;
        GridParams<T> hb;
        // First on the grid as built: its range is already the robust one (grid.h: make_grid_body clips at 3 sigma), so a cloud that
        // is uneven because of clusters keeps it and only gains the sub-box levels -- tight-cluster Chamfer at 1M: 2 x (125 us of
        // range histograms + 100 us of base rebuild) less. Only if most points sit in heavy cells (outliers so far out that they
        // stretched even the clipped range: the grid itself is useless) is the base refitted to the cloud's core range first.
        static const bool always_refit = getenv("PCU_HIP_REFIT_BASE") != nullptr;
        bool keep_base = !always_refit;
        auto add_levels = [&](const GridIndex<T>& from, const double** hs_dev) -> int { return skew_add_levels(ar, s, j, st, from, hs_dev); };
        j.skew_check = false;
        if (keep_base && pre && pre->on) {
            // enqueued and read back by the caller (skew_prelaunch), together with the other direction's
            memcpy(hc_redo, pre->hc_redo, sizeof hc_redo);
            keep_base = pre->hs[1] <= 0.5 * (double)j.ridx.n;
            if (!keep_base && st) { st->n_passes -= pre->d_passes; st->n_grid_builds -= pre->d_builds; }
            if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[skew] n=%d heavy points %.0f, level cells %.0f: %s (both directions enqueued together)\n", j.ridx.n, pre->hs[1], pre->hs[0], keep_base ? "grid kept" : "base refit");
        } else if (keep_base) {
            // everything enqueued at once -- both sub-box levels, the passes, the read-back of the heavy-cell statistics next to the counters:
            // ONE host round trip for the direction (the balance metric that brought us here, > 32 x the even value, already says that
            // levels are needed). If the statistics then say "base refit", what was enqueued is dropped.
            const double* hs_dev = nullptr; double hs[2] = {0, 0};
            pcu_hip_stats before; if (st) before = *st;
            if (add_levels(j.ridx, &hs_dev)) return -1;
            if (search_enqueue(c, s, j, st)) return -1;
            HIP_TRY(hipMemcpyAsync(hs, hs_dev, sizeof hs, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(hc_redo, j.sc.counters, sizeof hc_redo, hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
            keep_base = hs[1] <= 0.5 * (double)j.ridx.n;
            if (!keep_base && st) *st = before;          // (the attempt is dropped: its builds and passes are not the call's)
            if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[skew] n=%d heavy points %.0f, level cells %.0f: %s\n", j.ridx.n, hs[1], hs[0], keep_base ? "grid kept" : "base refit");
        }
        if (!keep_base) {
            QuantState<T>* qs = nullptr;
            if (core_range_enqueue(ar, j.ridx, j.d_ref_pts, s, &qs)) return -1;
            if (index_build_refit(ar, base, j.ridx, j.d_ref_pts, qs, (double)j.ridx.n / j.occ, s)) return -1;
            HIP_TRY(hipMemcpyAsync(&hb, base.gp, sizeof hb, hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
            if (st) st->n_grid_builds += 1;
            j.ridx = base; j.n_fine = 0;
            if ((double)hb.sumsq / (double)j.ridx.n > 4.0 * (j.occ + 1.0)) {       // still unbalanced after clipping the outliers
                if (add_levels(base, nullptr)) return -1;
            }
            if (search_enqueue(c, s, j, st)) return -1;
            HIP_TRY(hipMemcpyAsync(hc_redo, j.sc.counters, sizeof hc_redo, hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
        }
        hc = hc_redo; redone = true;
        if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[skew] n=%d lists: after finest %d, after mid %d, after base %d; ties %d\n", j.qidx.n, hc[C_X0], hc[C_X1], hc[C_U1], hc[C_T1]);
    }
    if (st) { st->n_escalated += hc[C_U1]; st->n_tie_flagged += hc[C_T1] + hc[C_TF0] + hc[C_TF1]; }
    int n_left = hc[C_U2];
    if (n_left == 0) {
        if (st) st->n_tie_true += hc[C_TT];
        j.n_tt = hc[C_TT];
        if (hc[C_TT] > 0 && j.tie_order) { if (tie_order_resolve(c, ar, s, j, hc[C_TT], st)) return -1; return redone ? 1 : 2; }   // 2: only the rows in the tie list changed
        return redone ? 1 : 0;
    }
    const int KL = std::max(2, pow2_at_least(j.k + 1));
    int* cur = j.sc.u2; int* nxt = j.sc.u1;
    double occ = j.occ;
    GridIndex<T> ridx = j.ridx;
    GridParams<T> hg;
    HIP_TRY(hipMemcpy(&hg, ridx.gp, sizeof hg, hipMemcpyDeviceToHost));
    int R = 2;                                            // already done on the fine grid
    for (int pass = 0; pass < 64 && n_left > 0; ++pass) {
        const int gmax = std::max(hg.G[0], std::max(hg.G[1], hg.G[2]));
        if (R >= gmax) return fail(PCU_HIP_ERR_RUNTIME, "internal: search did not certify with the whole grid scanned");
        // Coarser dataset grid (cell edge x8, restart at R=1) once the radius on the current grid gets expensive: a wave
        // scans (2R+1)^2 rows, so for a few left-over queries radius 8 and 16 on the fine grid are cheaper than a build
        // (Gaussian tails: 6.9 -> 1.4 ms), for many (whole clouds far apart) the coarse grid is.
        if (R >= (n_left > 4096 ? 4 : 16) && gmax > 8) {
            occ *= 512.0;
            GridIndex<T> coarse;
            if (index_alloc(ar, coarse, ridx.n, occ, false, false)) return -1;
            if (index_build(coarse, j.d_ref_pts, occ, s)) return -1;
            if (st) st->n_grid_builds++;
            ridx = coarse; R = 1;
            HIP_TRY(hipMemcpyAsync(&hg, ridx.gp, sizeof hg, hipMemcpyDeviceToHost, s));
        } else {
            R *= 2;
        }
        HIP_TRY(hipMemsetAsync(j.sc.counters + C_SPARE, 0, sizeof(int), s));
        SearchArgs<T> b = base_args(j, ridx);
        b.qlist = cur; b.nq = n_left; b.R = R;
        b.unresolved = nxt; b.n_unresolved = j.sc.counters + C_SPARE; b.ties = j.sc.tt; b.n_ties = j.sc.counters + C_TT;
        b.f_accum = j.fuse != FUSE_NONE;        // a fused call's stragglers (fused_continue): their share joins the accumulators
        if (launch_search_wave<T>(KL, b, s)) return -1;
        int left = 0;
        HIP_TRY(hipMemcpyAsync(&left, j.sc.counters + C_SPARE, sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        if (st) st->n_passes++;
        n_left = left;
        int* done = cur; cur = nxt; nxt = done;
    }
    if (n_left > 0) return fail(PCU_HIP_ERR_RUNTIME, "internal: too many search passes");
    int tt = 0;
    HIP_TRY(hipMemcpy(&tt, j.sc.counters + C_TT, sizeof(int), hipMemcpyDeviceToHost));
    if (st) st->n_tie_true += tt;
    j.n_tt = tt;
    if (tt > 0 && j.tie_order && tie_order_resolve(c, ar, s, j, tt, st)) return -1;
    return 1;
}

// One 256-byte device block per call holds everything the host must read back: both directions' counters and
// the scalar results. It is copied to pinned host memory with a single hipMemcpyAsync.
struct ResultBlock {
    int counters[2][C_N];        //   0..95   bytes
    double sums[2];              //  96..111
    double vals[2];              // 112..127  (T-typed values stored in the first sizeof(T) bytes of each slot pair)
    long long ij[4];             // 128..159
    int pad[24];
};
static_assert(sizeof(ResultBlock) == 256, "ResultBlock layout");

// Row-order restore of a job's cell-ordered results (either destination may be null).
template <typename T>
static int unpermute_enqueue(hipStream_t s, const SearchJob<T>& j, T* dst_d, long long* dst_i,
                             const ResultBlock* rb = nullptr, int* host_block = nullptr, unsigned seq = 0) {
    const long long n_elems = (long long)j.qidx.n * j.k;
    hipLaunchKernelGGL(k_unpermute<T>, dim3((unsigned)((n_elems + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                       j.qidx.pos_of, j.out_d, j.out_i, dst_d, dst_i, n_elems, j.k, reinterpret_cast<const int*>(rb), host_block, seq,
                       j.sc.counters + C_SKEW);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ validation
static int validate_sizes(int64_t nq, int64_t nr, const char* qname, const char* rname) {
    if (nq <= 0 || nr <= 0)
        return fail(PCU_HIP_ERR_INVALID,
                    "Invalid input set with zero elements: %s and %s must have shape (n, 3) and (m, 3). "
                    "Got %s.shape = (%lld, 3), %s.shape = (%lld, 3).", qname, rname, qname, (long long)nq, rname, (long long)nr);
    if (nq > 0x07fffff0ll || nr > 0x07fffff0ll)      // record byte offsets are 32-bit (32-byte f64 records)
        return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^27-16 rows are not supported");
    return 0;
}

struct Timer {
    pcu_hip_ctx* c; hipStream_t s; pcu_hip_stats* st;
    void mark(int i) { if (st && c->time_phases) (void)hipEventRecord(c->ev[i], s); }
    float span(int i, int j) { float ms = 0; if (st && c->time_phases) (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[j]); return ms; }
};

// Input staging: returns device pointer to the cloud (copying from host if needed).
template <typename T>
static int stage_in(Arena& ar, const T* p, int64_t n, bool on_dev, hipStream_t s, const T** out) {
    if (on_dev) { *out = p; return 0; }
    T* d = nullptr;
    if (aalloc(ar, &d, (size_t)n * 3)) return -1;
    HIP_TRY(hipMemcpyAsync(d, p, (size_t)n * 3 * sizeof(T), hipMemcpyHostToDevice, s));
    *out = d;
    return 0;
}

// ------------------------------------------------------------------------------------------------ knn
// A dataset kept on the GPU together with its grid index (pcu_hip_index_*): memory owned by the object, not by a call's arena.
struct pcu_hip_index {
    int elem_size = 0;            // 4: float, 8: double
    int device = 0;
    int64_t n = 0;
    double occ = 0;
    void* pts = nullptr;          // (n,3) device copy of the dataset
    void* mem = nullptr;          // block holding the index buffers
    GridIndex<float> g32; GridIndex<double> g64;
};
template <typename T> static const GridIndex<T>& index_grid(const pcu_hip_index* p);
template <> const GridIndex<float>& index_grid<float>(const pcu_hip_index* p) { return p->g32; }
template <> const GridIndex<double>& index_grid<double>(const pcu_hip_index* p) { return p->g64; }

// Operators that do not run the grid searches (which carry the check, search.h: index_not_ready) look at the flags themselves: one read-back.
template <typename T>
static int check_nonfinite(const GridParams<T>* gp, int mask, hipStream_t s) {
    int nf = 0;
    HIP_TRY(hipMemcpyAsync(&nf, reinterpret_cast<const char*>(gp) + offsetof(GridParams<T>, nonfinite), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_WAIT(s);
    return (nf & mask) ? nonfinite_error(false) : 0;
}
// k beyond the grid search's capacity (k > 127): the reference's own algorithm for every query -- its kd-tree, rebuilt on the GPU
// (kd_order.h), and its traversal with a wave-cooperative result set of k slots (k_kd_search_all). Any k > 0 is answered, as
// the reference does (src/point_cloud_distance.cpp:133-135); slots beyond the dataset size are padded (-1, -1.0) (:90-93).
template <typename T>
static int knn_big_k(pcu_hip_ctx* c, const T* query, int64_t nq, const T* dataset, int64_t nr, int k, int max_leaf,
                     T* out_d, int64_t* out_i, unsigned flags, void* stream, pcu_hip_stats* st, const pcu_hip_index* pidx) {
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE, squared = flags & PCU_HIP_SQUARED;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    c->time_phases = false; c->time_kernels = false;
    const size_t slot_bytes = (size_t)k * (sizeof(T) + 4);
    const bool rs_lds = slot_bytes <= 64 * 1024;
    int grid = (int)std::min<int64_t>(nq, 8192);
    if (!rs_lds) grid = (int)std::max<size_t>(1, std::min<size_t>((size_t)grid, ((size_t)1 << 30) / slot_bytes));
    const size_t out_bytes = align_up((size_t)nq * k * sizeof(T), 256) + align_up((size_t)nq * k * 8, 256);
    size_t need = (pidx ? 0 : index_bytes<T>(nr, 2.0)) + (rs_lds ? 0 : 2 * align_up((size_t)grid * slot_bytes, 256)) +
                  align_up((size_t)grid * 512 * sizeof(KdFrame<T>), 256) + 65536;
    if (!on_dev) need += align_up((size_t)nq * 3 * sizeof(T), 256) + (pidx ? 0 : align_up((size_t)nr * 3 * sizeof(T), 256)) + out_bytes;
    if (ctx_begin(c, need)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T *dq, *dr;
        if ((rc = stage_in(ar, query, nq, on_dev, s, &dq))) break;
        if (pidx) dr = static_cast<const T*>(pidx->pts);
        else if ((rc = stage_in(ar, dataset, nr, on_dev, s, &dr))) break;
        T* dd = out_d; long long* di = (long long*)out_i;
        if (!on_dev) { if ((rc = aalloc(ar, &dd, (size_t)nq * k))) break; if ((rc = aalloc(ar, &di, (size_t)nq * k))) break; }
        GridIndex<T> gi;                          // only its exact bounding box is used (root of the kd-tree)
        if (pidx) gi = index_grid<T>(pidx);
        else { if ((rc = index_alloc(ar, gi, nr, 2.0))) break; if ((rc = index_build(gi, dr, 2.0, s))) break; if (st) st->n_grid_builds = 1; }
        if ((rc = check_nonfinite<T>(gi.gp, kNfNaN | kNfBothInf, s))) break;
        KdBuild<T> b; int* err = nullptr; int levels = 0;
        if ((rc = kd_build_device(c, ar, s, dr, (int)nr, gi.gp, max_leaf > 0 ? max_leaf : 10, b, &err, &levels, nullptr))) break;
        KdSearchArgs<T> a;
        a.E = b.E; a.nodes = b.nodes; a.qsorted = nullptr; a.qlist = nullptr; a.qcount_dev = nullptr;
        a.k = k; a.squared = squared ? 1 : 0; a.row_out = 1; a.out_d = dd; a.out_i = di; a.error_flag = err;
        a.qraw = dq; a.nq_raw = (int)nq; a.rs_d = nullptr; a.rs_i = nullptr;
        a.cancel_word = g_cancel_mirror.load(std::memory_order_relaxed); a.cancel_gen = t_call_gen;
        if (!rs_lds && ((rc = aalloc(ar, &a.rs_d, (size_t)grid * k)) || (rc = aalloc(ar, &a.rs_i, (size_t)grid * k)))) break;
        KdFrame<T>* frames = nullptr;
        a.stack_cap = levels + 2;
        if ((rc = aalloc(ar, &frames, (size_t)grid * a.stack_cap))) break;
        a.stack = frames;
        static std::atomic<unsigned long long> attr_set[2];
        if (attr_unset_here(attr_set[sizeof(T) == 4 ? 0 : 1]))
            HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void*>(&k_kd_search_all<T>), hipFuncAttributeMaxDynamicSharedMemorySize, 64 * 1024));
        hipLaunchKernelGGL(k_kd_search_all<T>, dim3(grid), dim3(64), rs_lds ? slot_bytes : 0, s, a);
        HIP_TRY(hipGetLastError());
        int herr = 0;
        HIP_TRY(hipMemcpyAsync(&herr, err, sizeof(int), hipMemcpyDeviceToHost, s));
        if (!on_dev) {
            HIP_WAIT(s);        // (the traversal first, cancellably: copies to pageable memory block the host until they are done -- hundreds of MB here)
            HIP_TRY(hipMemcpyAsync(out_d, dd, (size_t)nq * k * sizeof(T), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_i, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_WAIT(s);
        if (herr) { rc = fail(PCU_HIP_ERR_RUNTIME, "internal: kd traversal exceeded the tree depth (%d)", levels); break; }
        if (st) { st->n_queries = nq; st->n_passes = 1; }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

template <typename T> static int job_unlean(hipStream_t s, SearchJob<T>& j);
template <typename T>
static int knn_attempt(pcu_hip_ctx* c, const T* query, int64_t nq, const T* dataset, int64_t nr, int k, int max_leaf,
                       T* out_d, int64_t* out_i, unsigned flags, void* stream, pcu_hip_stats* st, const pcu_hip_index* pidx, int restarts) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (pidx) {
        if (pidx->elem_size != (int)sizeof(T)) return fail(PCU_HIP_ERR_INVALID, "the index was built for the other floating-point type");
        if (pidx->device != c->device) return fail(PCU_HIP_ERR_INVALID, "the index lives on device %d, the context on device %d", pidx->device, c->device);
        nr = pidx->n;
    }
    if (k <= 0) return fail(PCU_HIP_ERR_INVALID, "Invalid value for k (%d) must be greater than 0.", k);
    if (validate_sizes(nq, nr, "query_points", "dataset_points")) return PCU_HIP_ERR_INVALID;
    if (k > kMaxK) return knn_big_k<T>(c, query, nq, dataset, nr, k, max_leaf, out_d, out_i, flags, stream, st, pidx);
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE, squared = flags & PCU_HIP_SQUARED;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    c->time_phases = flags & PCU_HIP_TIME_PHASES; c->time_kernels = flags & PCU_HIP_TIME_KERNELS;
    const double occ = pidx ? pidx->occ : call_occupancy(c, k, 1);
    const double occ_q = 2.0;
    // (one grid over query cloud and dataset when they are of comparable size and occupancy -- k = 1: the lanes of a wave then look at neighbouring
    // cells of one dataset row, profiles/r06_flat_aligned_ab.txt; both indexes are planned for the larger cloud)
    const bool share = !pidx && shared_grid_wanted(c, nq, nr, occ_q, occ);
    const int64_t n_plan = share ? std::max(nq, nr) : 0;
    size_t need = (pidx ? 0 : index_bytes<T>(std::max(nr, n_plan), occ)) + index_bytes<T>(std::max(nq, n_plan), occ_q) + scratch_bytes<T>(nq) + 8192 +
                  align_up((size_t)nq * k * sizeof(T), 256) + align_up((size_t)nq * k * 8, 256);     // cell-ordered results
    if (!on_dev) need += align_up((size_t)nq * 3 * sizeof(T), 256) + (pidx ? 0 : align_up((size_t)nr * 3 * sizeof(T), 256)) +
                         align_up((size_t)nq * k * sizeof(T), 256) + align_up((size_t)nq * k * 8, 256);
    if (ctx_begin(c, need)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    Timer tm{c, s, st};
    int rc = 0;
    do {
        const T *dq, *dr;
        if ((rc = stage_in(ar, query, nq, on_dev, s, &dq))) break;
        if (pidx) dr = static_cast<const T*>(pidx->pts);
        else if ((rc = stage_in(ar, dataset, nr, on_dev, s, &dr))) break;
        T* dd = out_d; long long* di = (long long*)out_i;
        if (!on_dev) { if ((rc = aalloc(ar, &dd, (size_t)nq * k))) break; if ((rc = aalloc(ar, &di, (size_t)nq * k))) break; }
        SearchJob<T> job;
        // Every kernel writes a result row straight to the query's ORIGINAL row: no cell-ordered copy, no row-order restore pass, no row -> slot
        // table from the index build. For k >= 4 (rows >= 48 bytes) that was always so; for k < 4 the rows used to be written coalesced in the
        // queries' cell order and gathered back by k_unpermute -- two scattered line fetches per query (185 MB for a 12 MB result at 1M, k = 1) plus
        // 37 MB of scattered row -> slot stores in the build. The scattered 4 / 8-byte row stores cost 67 MB instead (profiles/r04_c2_ab.txt):
        // config 2 0.184 -> 0.154 ms. PCU_HIP_ROW_OUT_MIN_K=4 restores the old split.
        static const int row_out_min_k = getenv("PCU_HIP_ROW_OUT_MIN_K") ? atoi(getenv("PCU_HIP_ROW_OUT_MIN_K")) : 1;
        const bool row_out = k >= row_out_min_k;
        job.row_out = row_out;
        if (pidx) job.ridx = index_grid<T>(pidx);
        else if ((rc = index_alloc(ar, job.ridx, nr, occ, false, true, use_one_pass(c), n_plan))) break;
        if ((rc = index_alloc(ar, job.qidx, nq, occ_q, /*want_pos=*/!row_out, true, use_one_pass(c), n_plan))) break;
        job.ridx.shared_grid = job.qidx.shared_grid = share;
        job.qidx.src = dq; job.qidx.occ_built = occ_q;
        if (!pidx) { job.ridx.src = dr; job.ridx.occ_built = occ; }
        ResultBlock* rb = nullptr;
        if ((rc = aalloc(ar, &rb, 1))) break;
        if ((rc = scratch_alloc(ar, job.sc, nq, rb->counters[0]))) break;
        job.d_ref_pts = dr; job.occ = occ; job.k = k; job.squared = squared;
        job_rescale_setup(c, job, !pidx && restarts < 2, 1);
        if (row_out) { job.out_d = dd; job.out_i = di; }
        else {
            if ((rc = aalloc(ar, &job.out_d, (size_t)nq * k))) break;
            if ((rc = aalloc(ar, &job.out_i, (size_t)nq * k))) break;
        }
        job.leaf_max = max_leaf > 0 ? max_leaf : 10; job.tie_order = !(flags & PCU_HIP_NO_TIE_ORDER);
        {   // k = 1 on a fresh pair of indexes: the lane kernel reads the coordinate and row-id streams only, so the Pt4 records are not
            // written (grid2.h; 16 MB less per million points); whatever runs after the first read-back fills them in first (job_unlean)
            static const bool no_lean = getenv("PCU_HIP_NO_LEAN") != nullptr;
            const bool lean = !pidx && !no_lean && row_out && lane_k1_job(job) && job.ridx.bucketed && job.qidx.bucketed && job.ridx.one_pass && job.qidx.one_pass;
            job.ridx.lean = job.qidx.lean = lean;
        }
        tm.mark(0);
        if (pidx) { if ((rc = index_build<T>(job.qidx, dq, occ_q, s, /*defer_large=*/!c->eager_large, rb, (int)(sizeof(ResultBlock) / 4), c))) break; }
        else if ((rc = index_build_pair<T>(job.ridx, dr, occ, &job.qidx, dq, occ_q, s, !c->eager_large, rb, (int)(sizeof(ResultBlock) / 4), c))) break;
        if (st) st->n_grid_builds += pidx ? 1 : 2;
        tm.mark(1);
        const bool spec = kd_speculate_fork(c, s, job);
        const bool lazy_wave = lazy_wave_job(job);          // (k = 1: the wave pass only if the lane launch leaves something, see below)
        if ((rc = search_enqueue(c, s, job, st, /*zero_counters=*/false, lazy_wave ? 1 : 3))) break;
        if (spec && (rc = kd_speculate(c, ar, s, job))) break;
        if (row_out) { hipLaunchKernelGGL(k_result_block_to_host, dim3(1), dim3(64), 0, s, reinterpret_cast<const int*>(rb), c->h_pinned, ++c->seq); HIP_TRY(hipGetLastError()); }
        else if ((rc = unpermute_enqueue(s, job, dd, di, rb, c->h_pinned, ++c->seq))) break;   // optimistic: redone below if stragglers / ties remain
        tm.mark(2);
        HIP_WAIT(s);         // the per-row outputs must be complete, so this call waits for the stream, not for the word
        if ((unsigned)*(volatile int*)(c->h_pinned + 63) != c->seq) { rc = fail(PCU_HIP_ERR_RUNTIME, "internal: the result block did not arrive"); break; }
        if (job.ridx.lean || job.qidx.lean) {
            const int* hc0 = ((ResultBlock*)c->h_pinned)->counters[0];
            bool clean = true;
            for (int i = 0; i < C_N; ++i) clean = clean && hc0[i] == 0;
            if (!clean && (rc = job_unlean(s, job))) break;      // stragglers, ties, give-ups: everything from here on may read Pt4 records
        }
        if (lazy_wave) {
            const int* hc0 = ((ResultBlock*)c->h_pinned)->counters[0];
            if ((hc0[C_U1] > 0 || hc0[C_T1] > 0) && !hc0[C_SKEW] && !hc0[C_LARGE]) {       // stragglers beyond radius 2, possible ties, deferred lanes: the wave pass now
                if ((rc = search_enqueue(c, s, job, st, false, 2))) break;
                hipLaunchKernelGGL(k_result_block_to_host, dim3(1), dim3(64), 0, s, reinterpret_cast<const int*>(rb), c->h_pinned, ++c->seq); HIP_TRY(hipGetLastError());
                HIP_WAIT(s);
                if ((unsigned)*(volatile int*)(c->h_pinned + 63) != c->seq) { rc = fail(PCU_HIP_ERR_RUNTIME, "internal: the result block did not arrive"); break; }
            }
        }
        if ((rc = search_finish(c, ar, s, job, st, ((ResultBlock*)c->h_pinned)->counters[0])) < 0) { if (rc == PCU_NONFINITE) rc = nonfinite_error(false); break; }
        if (rc == 3) { rc = PCU_RETRY; break; }
        if (row_out) { if (rc > 0) tm.mark(2); }
        else if (rc == 1) { if ((rc = unpermute_enqueue(s, job, dd, di))) break; tm.mark(2); }
        else if (rc == 2) {       // the resolver rewrote only the tied queries' rows: restore just those
            hipLaunchKernelGGL(k_unpermute_rows<T>, dim3((unsigned)((job.n_tt * (long long)k + kBlock - 1) / kBlock)), dim3(kBlock), 0, s,
                               job.sc.tt, job.n_tt, job.qidx.sorted, job.out_d, job.out_i, dd, di, k);
            HIP_TRY(hipGetLastError());
            tm.mark(2);
        }
        rc = 0;
        if (!on_dev) {
            HIP_TRY(hipMemcpyAsync(out_d, dd, (size_t)nq * k * sizeof(T), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_i, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_WAIT(s);
        if (st) { st->n_queries = nq; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 2); collect_kernel_times(c, st); }
    } while (0);
    kd_speculate_end(c, rc != PCU_RETRY);
    ctx_end(c);
    if (rc == PCU_RETRY) return PCU_RETRY;
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename T>
static int knn_impl(pcu_hip_ctx* c, const T* query, int64_t nq, const T* dataset, int64_t nr, int k, int max_leaf,
                    T* out_d, int64_t* out_i, unsigned flags, void* stream, pcu_hip_stats* st, const pcu_hip_index* pidx = nullptr) {
    for (int restarts = 0;; ++restarts) {          // (restarts: occupancy rescale, see kRescaleAbove)
        const int rc = knn_attempt<T>(c, query, nq, dataset, nr, k, max_leaf, out_d, out_i, flags, stream, st, pidx, restarts);
        if (rc != PCU_RETRY) return rc;
    }
}

// ------------------------------------------------------------------------------------------------ two-sided ops
// Shared front end of hausdorff / chamfer: both clouds indexed once; x->y and y->x searches with k = 1, both
// enqueued back to back (each direction owns its lists and counters), epilogues enqueued behind them, ONE
// stream synchronisation for the whole call unless some query needs the host-driven coarse-grid loop.
//
// Fused calls (Chamfer p = 2 without indices, Hausdorff; reduce.h "fused epilogues"): the searches reduce on the fly and the
// wave-per-query launch ends the call, so a step is index build + 2 launches and no result rows exist. Whenever the fused
// attempt cannot stand -- the dataset grid is unbalanced (refit path), some query is still uncertified after radius 2, or
// (Hausdorff) the arg-max row has exactly tied neighbours whose order matters -- the call is redone through the row-based
// path below, which handles all of that.
struct CallBlock {               // one per call, zeroed by the first index-build launch
    ResultBlock rb;              // pad[0] = epilogue ticket, pad[2..3] = tie bit of the fused arg-max winner per direction
    unsigned long long limbs[2][kAccLimbs];     // exact sums of the wave-per-query pass (reduce.h)
    double special[2];
};
template <typename T>
struct PairState {
    const T *dx = nullptr, *dy = nullptr;
    SearchJob<T> xy, yx;                                // x rows searched in y / y rows searched in x
    T* pv = nullptr; long long* pi = nullptr; double* pd = nullptr;   // reduction partials (per direction: 2 x kRedBlocks)
    T* res_v = nullptr; long long* res_ij = nullptr; double* res_s = nullptr;
    ResultBlock* rb = nullptr; CallBlock* cb = nullptr;
    bool two = true;
    bool allow_rescale = true;                          // cleared on the last restart of a call (occupancy rescale)
    int fuse = FUSE_NONE; FuseTail<T> tail;             // fused attempt (tail: arguments of the launch that ends the call)
    bool wave_pending = false;                          // the fused attempt's wave-per-query pass has not been launched (pair_search_enqueue)
    int* tie_hit = nullptr;
};
template <typename T>
static size_t pair_bytes(int64_t nx, int64_t ny, double occ_x, double occ_y, bool on_dev) {
    // (index_bytes of the LARGER cloud for both: a shared grid is planned for it, index_alloc's n_plan)
    size_t b = index_bytes<T>(std::max(nx, ny), occ_x) + index_bytes<T>(std::max(nx, ny), occ_y) + scratch_bytes<T>(nx) + scratch_bytes<T>(ny) +
               align_up((size_t)nx * sizeof(T), 256) + align_up((size_t)ny * sizeof(T), 256) +
               align_up((size_t)nx * 8, 256) + align_up((size_t)ny * 8, 256) +
               6 * align_up((size_t)kRedBlocks * 8, 256) + 8192 +
               3 * (align_up((size_t)grid8((int)nx, kBlock) * 8, 256) + align_up((size_t)grid8((int)ny, kBlock) * 8, 256)) +
               4 * align_up((size_t)kWaveBlocks * (kBlock / 64) * 8, 256) + align_up(sizeof(CallBlock), 256) + 1024;
    if (!on_dev) b += align_up((size_t)nx * 3 * sizeof(T), 256) + align_up((size_t)ny * 3 * sizeof(T), 256) +
                      align_up((size_t)nx * 8, 256) + align_up((size_t)ny * 8, 256);
    return b;
}
// A fused two-sided attempt is lane pass + fold: the k = 1 lane kernel finishes its own stragglers (search.h: radius 2 inside the lane), so the
// wave-per-query pass -- a launch on the critical path of every call for a few hundred queries -- runs only when a direction's lists are not
// empty afterwards (fused_needs_wave; then: wave pass + a second fold). PCU_HIP_FUSED_WAVE=1 launches it up front as before.
static int wait_result_block(pcu_hip_ctx* c, hipStream_t s);
static bool fused_wave_upfront() { static const bool v = getenv("PCU_HIP_FUSED_WAVE") != nullptr; return v; }
template <typename T>
static int pair_search_enqueue(pcu_hip_ctx* c, hipStream_t s, PairState<T>& P, pcu_hip_stats* st) {
    const bool lazy_wave = P.fuse && P.two && !fused_wave_upfront() && lane_k1_job(P.xy) && lane_k1_job(P.yx);
    if (P.two) { if (search_enqueue_pair(c, s, P.xy, P.yx, st, lazy_wave ? 1 : 3)) return -1; }
    else if (search_enqueue(c, s, P.xy, st, /*zero_counters=*/false)) return -1;
    P.wave_pending = lazy_wave;
    if (P.fuse) {               // the launch that ends a fused call (reduce.h)
        P.tail.nwaves = lazy_wave ? 0 : kWaveBlocks * (kBlock / 64);       // (no wave pass yet: its per-wave slots hold nothing)
        hipLaunchKernelGGL(k_fuse_tail<T>, dim3(1), dim3(kTailThreads), 0, s, P.tail);
        HIP_TRY(hipGetLastError());
    }
    return 0;
}
// The first fold of a lazy fused attempt has arrived: do some queries still need the wave-per-query pass (stragglers beyond radius 2, lanes that
// deferred next to a heavy cell)? Then run it now, fold again, and wait for that result block.
template <typename T>
static int fused_wave_if_needed(pcu_hip_ctx* c, hipStream_t s, PairState<T>& P, pcu_hip_stats* st, ResultBlock& host) {
    // The layout handed down from the previous call no longer fits these clouds (grid2.h: kGeoStale; every pass gave up): once more, laid out afresh.
    if ((host.counters[0][C_LARGE] | (P.two ? host.counters[1][C_LARGE] : 0)) & kGeoStale) { c->geo.valid[0] = c->geo.valid[1] = false; return PCU_RETRY; }
    if (P.xy.brick) {           // the staged pass's report: blocks that scanned from global memory (search_brick.h)
        const long long fb = (long long)host.counters[0][C_SPARE] + host.counters[1][C_SPARE], nb = (long long)P.xy.n_flat + P.yx.n_flat;
        if (4 * fb > nb) c->brick_off = true;
    }
    if (!P.wave_pending) return 0;
    P.wave_pending = false;
    bool need = false, broken = false;
    for (int d = 0; d < 2; ++d) {
        need = need || host.counters[d][C_U1] > 0 || host.counters[d][C_T1] > 0;
        broken = broken || host.counters[d][C_SKEW] || host.counters[d][C_LARGE];
    }
    if (!need || broken) return 0;            // (a pass that gave up listed nothing: the row-based path takes the call over)
    if (search_enqueue_pair(c, s, P.xy, P.yx, st, 2)) return -1;
    P.tail.nwaves = kWaveBlocks * (kBlock / 64); P.tail.seq = ++c->seq;
    hipLaunchKernelGGL(k_fuse_tail<T>, dim3(1), dim3(kTailThreads), 0, s, P.tail);
    HIP_TRY(hipGetLastError());
    if (wait_result_block(c, s)) return -1;
    memcpy(&host, c->h_pinned, sizeof host);
    return 0;
}
template <typename T>
static int pair_setup(pcu_hip_ctx* c, Arena& ar, hipStream_t s, const T* x, int64_t nx, const T* y, int64_t ny, bool on_dev,
                      bool squared, double occ_x, double occ_y, long long* ext_ixy, long long* ext_iyx, PairState<T>& P, Timer& tm,
                      pcu_hip_stats* st, bool two_sided, int max_leaf, bool tie_order_xy, bool tie_order_yx, int fuse_mode = FUSE_NONE) {
    // (ext_ixy / ext_iyx: device arrays of the caller that take the correspondences directly. Result rows of the row-based path are in the
    // caller's ROW order -- the search kernels write them there, see knn_attempt -- so no row -> slot table is built and no restore pass runs.)
    constexpr bool want_pos_x = false, want_pos_y = false;
    P.xy.row_out = P.yx.row_out = true;
    P.two = two_sided;
    P.xy.leaf_max = P.yx.leaf_max = max_leaf > 0 ? max_leaf : 10; P.xy.tie_order = tie_order_xy; P.yx.tie_order = tie_order_yx;
    if (stage_in(ar, x, nx, on_dev, s, &P.dx)) return -1;
    if (stage_in(ar, y, ny, on_dev, s, &P.dy)) return -1;
    GridIndex<T> ix, iy;
    const bool share = two_sided && shared_grid_wanted(c, nx, ny, occ_x, occ_y);
    const int64_t n_plan = share ? std::max(nx, ny) : 0;
    if (index_alloc(ar, ix, nx, occ_x, want_pos_x, true, use_one_pass(c), n_plan) || index_alloc(ar, iy, ny, occ_y, want_pos_y, true, use_one_pass(c), n_plan)) return -1;
    ix.shared_grid = iy.shared_grid = share;
    ix.src = P.dx; iy.src = P.dy; ix.occ_built = occ_x; iy.occ_built = occ_y;        // (the jobs below hold copies: what a rebuild after a slot overflow starts from)
    if (ix.bucketed && iy.bucketed && ix.one_pass != iy.one_pass) ix.one_pass = iy.one_pass = false;
    P.xy.qidx = ix; P.xy.ridx = iy; P.xy.d_ref_pts = P.dy;
    P.yx.qidx = iy; P.yx.ridx = ix; P.yx.d_ref_pts = P.dx;
    P.xy.occ = occ_y; P.yx.occ = occ_x;            // a direction's occupancy is its dataset's
    P.xy.k = P.yx.k = 1; P.xy.squared = P.yx.squared = squared;
    // Non-finite input (nonfinite_error): a cloud that is searched in must not hold NaN / both infinities along an axis; query rows may hold
    // anything. (Both clouds of a two-sided call are searched in, each by the other direction's job.) A fused attempt, whose epilogues know
    // nothing of unmatched rows, takes finite clouds only: it gives up on any non-finite coordinate and the call is redone row-based (below).
    P.xy.bad_r = P.yx.bad_r = kNfHard; P.xy.bad_q = P.yx.bad_q = 0;
    job_rescale_setup(c, P.xy, P.allow_rescale, 1); job_rescale_setup(c, P.yx, P.allow_rescale, 0);
    if (aalloc(ar, &P.cb, 1)) return -1;
    P.rb = &P.cb->rb;
    if (scratch_alloc(ar, P.xy.sc, nx, P.rb->counters[0]) || scratch_alloc(ar, P.yx.sc, ny, P.rb->counters[1])) return -1;
    if (aalloc(ar, &P.xy.out_d, (size_t)nx) || aalloc(ar, &P.yx.out_d, (size_t)ny)) return -1;
    P.xy.out_i = ext_ixy; P.yx.out_i = ext_iyx;
    if ((!P.xy.out_i && aalloc(ar, &P.xy.out_i, (size_t)nx)) || (!P.yx.out_i && aalloc(ar, &P.yx.out_i, (size_t)ny))) return -1;
    if (aalloc(ar, &P.pv, (size_t)2 * kRedBlocks) || aalloc(ar, &P.pi, (size_t)2 * kRedBlocks) || aalloc(ar, &P.pd, (size_t)2 * kRedBlocks)) return -1;
    if (aalloc(ar, &P.tie_hit, 16)) return -1;
    P.res_v = reinterpret_cast<T*>(P.rb->vals); P.res_ij = P.rb->ij; P.res_s = P.rb->sums;
    // (indexing cloud y on a second stream beside cloud x was measured in round 1: ~3 %, less than sharing every launch; removed)
    // fused attempt: only when every direction takes the k = 1 lane-per-query kernel on one stream
    static const bool no_fuse = getenv("PCU_HIP_NO_FUSE") != nullptr;
    P.fuse = FUSE_NONE;
    if (fuse_mode != FUSE_NONE && !no_fuse && lane_k1_job(P.xy) && (!two_sided || lane_k1_job(P.yx))) {
        P.fuse = fuse_mode;
        P.xy.bad_r = P.xy.bad_q = P.yx.bad_r = P.yx.bad_q = kNfNaN | kNfBothInf | kNfAnyInf;
        FuseTail<T>& t = P.tail; memset(&t, 0, sizeof t);
        t.mode = fuse_mode; t.njobs = two_sided ? 2 : 1; t.nwaves = kWaveBlocks * (kBlock / 64);
        for (int d = 0; d < t.njobs; ++d) {
            SearchJob<T>& J = d ? P.yx : P.xy;
            J.fuse = fuse_mode; J.n_flat = grid8(J.qidx.n, kBlock);
            if (aalloc(ar, &J.f_sum, (size_t)J.n_flat) || aalloc(ar, &J.f_max_v, (size_t)J.n_flat) || aalloc(ar, &J.f_max_k, (size_t)J.n_flat)) return -1;
            if (aalloc(ar, &J.f_wave_v, (size_t)t.nwaves) || aalloc(ar, &J.f_wave_k, (size_t)t.nwaves)) return -1;
            J.f_limbs = &P.cb->limbs[d][0]; J.f_special = &P.cb->special[d];
            t.flat_sum[d] = J.f_sum; t.flat_v[d] = J.f_max_v; t.flat_k[d] = J.f_max_k; t.nflat[d] = J.n_flat;
            t.wave_v[d] = J.f_wave_v; t.wave_k[d] = J.f_wave_k; t.limbs[d] = J.f_limbs; t.special[d] = J.f_special;
        }
        {   // Hausdorff: the value-only lane pass + the tail's resolution of the one winning query (PCU_HIP_NO_MAXVAL=1: the winner-tracking lane pass)
            static const bool maxval_off = getenv("PCU_HIP_NO_MAXVAL") != nullptr;
            t.maxval = (fuse_mode == FUSE_ARGMAX && !maxval_off) ? 1 : 0; t.squared = squared ? 1 : 0;
            P.xy.maxval = P.yx.maxval = t.maxval != 0;
        }
        {   // PCU_HIP_PROF_TAIL=1: stage timers of k_fuse_tail, printed every 256 calls (diagnostics)
            static const bool prof_tail = getenv("PCU_HIP_PROF_TAIL") != nullptr;
            static long long* tail_prof = nullptr; static long n_tail = 0;
            if (prof_tail) {
                if (!tail_prof) { HIP_TRY(hipMalloc((void**)&tail_prof, 8 * sizeof(long long))); HIP_TRY(hipMemset(tail_prof, 0, 8 * sizeof(long long))); }
                if (++n_tail % 256 == 0) {
                    long long h[8]; HIP_TRY(hipMemcpy(h, tail_prof, sizeof h, hipMemcpyDeviceToHost)); HIP_TRY(hipMemset(tail_prof, 0, sizeof h));
                    const double nb = h[7] > 0 ? (double)h[7] * 100.0 : 100.0;
                    fprintf(stderr, "[fuse_tail prof] calls %lld | mean us: partials %.2f  winners %.2f  resolve %.2f  finish %.2f\n", h[7], h[0] / nb, h[1] / nb, h[2] / nb, h[3] / nb);
                }
                t.prof = tail_prof;
            }
        }
        t.result_block = reinterpret_cast<const int*>(P.rb); t.host_block = c->h_pinned; t.seq = ++c->seq;
        t.w_sums = (int)(offsetof(ResultBlock, sums) / 4); t.w_vals = (int)(offsetof(ResultBlock, vals) / 4);
        t.w_ij = (int)(offsetof(ResultBlock, ij) / 4); t.w_tie = (int)(offsetof(ResultBlock, pad) / 4) + 2;
    }
    // A fused attempt reads the clouds' coordinate + row-id streams only: its index build leaves the Pt4 records unwritten (grid2.h: 12 bytes per
    // point less to write); whoever takes the call over when the attempt does not stand fills them in first (pair_unlean).
    {
        static const bool no_lean = getenv("PCU_HIP_NO_LEAN") != nullptr;
        const bool lean = P.fuse != FUSE_NONE && !no_lean && ix.bucketed && iy.bucketed && ix.one_pass && iy.one_pass && !want_pos_x && !want_pos_y;
        ix.lean = iy.lean = P.xy.qidx.lean = P.xy.ridx.lean = P.yx.qidx.lean = P.yx.ridx.lean = lean;
    }
    tm.mark(0);
    g_hprof.mark(1);
    // (the first build's first kernel also zeroes the call block: both directions' counters, the epilogue's ticket, the exact sums)
    // (defer_large: the placement of over-full buckets is launched only if a search reports them, see search_finish)
    if (index_build_pair<T>(ix, P.dx, occ_x, &iy, P.dy, occ_y, s, /*defer_large=*/!c->eager_large, P.cb, (int)(sizeof(CallBlock) / 4), c, /*keep_layout=*/P.fuse != FUSE_NONE)) return -1;
    P.xy.qidx.lean = P.yx.ridx.lean = ix.lean; P.xy.ridx.lean = P.yx.qidx.lean = iy.lean;      // (the build says what it wrote)
    P.xy.qidx.shared_grid = P.yx.ridx.shared_grid = ix.shared_grid; P.xy.ridx.shared_grid = P.yx.qidx.shared_grid = iy.shared_grid;
    {   // shared grid + fused sum + float: the staged lane pass (search_brick.h); its per-block partials follow its block size
        // (measured: 118 us against k_search1_flat's 60.5 on the same shared grid, profiles/r06_flat_aligned_ab.txt -- OFF unless PCU_HIP_BRICK=1)
        static const bool brick_env_on = getenv("PCU_HIP_BRICK") && atoi(getenv("PCU_HIP_BRICK")) != 0;
        const bool brick = sizeof(T) == 4 && P.fuse == FUSE_SUM && two_sided && ix.shared_grid && iy.shared_grid && brick_env_on && !c->brick_off;
        P.xy.brick = P.yx.brick = brick;
        if (brick) for (int d = 0; d < 2; ++d) { SearchJob<T>& J = d ? P.yx : P.xy; J.n_flat = grid8(J.qidx.n, kBrickNT); P.tail.nflat[d] = J.n_flat; }
    }
    if (P.fuse == FUSE_ARGMAX) for (int d = 0; d < (two_sided ? 2 : 1); ++d) {       // what the tail needs to resolve a winner: the direction's dataset index and query stream
        const SearchJob<T>& J = d ? P.yx : P.xy;
        P.tail.r_gp[d] = J.ridx.gp; P.tail.r_cs[d] = J.ridx.cell_start; P.tail.r_xyz[d] = xyz_of(J.ridx.sorted, J.ridx.n); P.tail.r_idx[d] = idx32_of(J.ridx.sorted, J.ridx.n);
        P.tail.q_xyz[d] = xyz_of(J.qidx.sorted, J.qidx.n);
    }
    if (st) st->n_grid_builds += 2;
    tm.mark(1);
    if (pair_search_enqueue(c, s, P, st)) return -1;
    tm.mark(2);
    return 0;
}
// The epilogue kernel's last block writes the result block into pinned host memory and then the call's sequence number
// into its last word: the host spins on that word (a few hundred ns after the store) instead of paying a stream
// synchronisation's wake-up latency (tens of us per call). Everything enqueued before that kernel has completed by then
// (in-order stream). With event timing on, or if the word does not arrive, fall back to hipStreamSynchronize.
static int wait_result_block(pcu_hip_ctx* c, hipStream_t s) {
    static const bool no_spin = getenv("PCU_HIP_NO_SPIN") != nullptr;
    if (!no_spin && !c->time_phases && !c->time_kernels) {
        volatile int* flag = c->h_pinned + 63;
        const auto t0 = std::chrono::steady_clock::now();
        for (unsigned it = 0;; ++it) {
            if ((unsigned)*flag == c->seq) { std::atomic_thread_fence(std::memory_order_acquire); return cancel_requested() ? cancelled(s) : 0; }
            if ((it & 0x3ff) == 0x3ff) {
                if (cancel_requested()) return cancelled(s);
                if (std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(200)) break;   // long call or fault: poll the stream instead
            }
        }
    }
    HIP_WAIT(s);
    if ((unsigned)*(volatile int*)(c->h_pinned + 63) != c->seq) return fail(PCU_HIP_ERR_RUNTIME, "internal: the epilogue kernel did not deliver its result block");
    return 0;
}
// A fused attempt stands when nothing needs the row-based machinery (see the head of this section).
template <typename T>
static bool fused_ok(const PairState<T>& P, const ResultBlock& h, bool tie_matters, bool stragglers_done = false) {
    for (int d = 0; d < (P.two ? 2 : 1); ++d) {
        if (h.counters[d][C_SKEW] || h.counters[d][C_LARGE] || (h.counters[d][C_U2] > 0 && !stragglers_done)) return false;
        if (tie_matters && P.fuse == FUSE_ARGMAX && h.pad[2 + d]) return false;
    }
    return true;
}
// A fused attempt that failed on the balance check: restart at another grid resolution? (rescale_wanted updates the context)
template <typename T>
static bool fused_rescale(pcu_hip_ctx* c, hipStream_t s, const PairState<T>& P, const ResultBlock& h) {
    bool want = false;
    for (int d = 0; d < (P.two ? 2 : 1) && !want; ++d)
        if (h.counters[d][C_SKEW]) want = rescale_wanted(c, d ? P.yx : P.xy, s, h.counters[d][C_SKEW]);
    return want;
}
// A fused attempt whose only flaw is that some queries are still uncertified after radius 2 (sparse tails, outliers): finish
// those with the host-driven passes of search_finish -- in fused mode the wave-per-query kernel adds every query it certifies to
// the direction's exact sum / arg-max slots -- and fold again. Returns 1 if the call is complete (host block refreshed), 0 if
// the row-based path has to take over, < 0 on error.
template <typename T>
static int fused_continue(pcu_hip_ctx* c, Arena& ar, hipStream_t s, PairState<T>& P, pcu_hip_stats* st, ResultBlock& host, bool tie_matters) {
    static const bool off = getenv("PCU_HIP_NO_FUSED_CONTINUE") != nullptr;
    if (off) return 0;
    const int nd = P.two ? 2 : 1;
    bool any = false;
    for (int d = 0; d < nd; ++d) {
        if (host.counters[d][C_SKEW] || host.counters[d][C_LARGE]) return 0;
        any = any || host.counters[d][C_U2] > 0;
    }
    if (!any) return 0;                         // (a tied arg-max row: rows are needed)
    for (int d = 0; d < nd; ++d) {
        if (host.counters[d][C_U2] <= 0) { if (st) { st->n_escalated += host.counters[d][C_U1]; st->n_tie_flagged += host.counters[d][C_T1]; } continue; }
        const int r = search_finish(c, ar, s, d ? P.yx : P.xy, st, host.counters[d]);
        if (r < 0) return r;
    }
    P.tail.seq = ++c->seq;
    hipLaunchKernelGGL(k_fuse_tail<T>, dim3(1), dim3(kTailThreads), 0, s, P.tail);
    HIP_TRY(hipGetLastError());
    if (wait_result_block(c, s)) return -1;
    memcpy(&host, c->h_pinned, sizeof host);
    return fused_ok(P, host, tie_matters, /*stragglers_done=*/true) ? 1 : 0;
}
// The Pt4 records of a pair's lean indexes (grid2.h), for everything that is not the fused fast path.
template <typename T>
static int pair_unlean(hipStream_t s, PairState<T>& P) {
    GridIndex<T>& ix = P.xy.qidx; GridIndex<T>& iy = P.xy.ridx;
    if (!ix.lean && !iy.lean) return 0;
    const Pt4Side<T> a{ix.sorted, ix.n}, b{iy.sorted, iy.n};
    const int nb0 = ix.lean ? std::min((ix.n + 8 + kBlock - 1) / kBlock, 2048) : 0, nb1 = iy.lean ? std::min((iy.n + 8 + kBlock - 1) / kBlock, 2048) : 0;
    hipLaunchKernelGGL(k_make_pt4<T>, dim3(nb0 + nb1), dim3(kBlock), 0, s, a, b, nb0);
    HIP_TRY(hipGetLastError());
    P.xy.qidx.lean = P.xy.ridx.lean = P.yx.qidx.lean = P.yx.ridx.lean = false;
    return 0;
}
// The same for a single job (k_nearest_neighbors, k = 1).
template <typename T>
static int job_unlean(hipStream_t s, SearchJob<T>& j) {
    GridIndex<T>& ix = j.qidx; GridIndex<T>& iy = j.ridx;
    if (!ix.lean && !iy.lean) return 0;
    const Pt4Side<T> a{ix.sorted, ix.n}, b{iy.sorted, iy.n};
    const int nb0 = ix.lean ? std::min((ix.n + 8 + kBlock - 1) / kBlock, 2048) : 0, nb1 = iy.lean ? std::min((iy.n + 8 + kBlock - 1) / kBlock, 2048) : 0;
    hipLaunchKernelGGL(k_make_pt4<T>, dim3(nb0 + nb1), dim3(kBlock), 0, s, a, b, nb0);
    HIP_TRY(hipGetLastError());
    ix.lean = iy.lean = false;
    return 0;
}
// A pass refused non-finite input (counter bit 4): the classification of both clouds (GridParams::nonfinite; [0] = x / source, [1] = y / target).
template <typename T>
static int pair_nonfinite_flags(hipStream_t s, const PairState<T>& P, int (&nf)[2]) {
    const GridParams<T>* gp[2] = {P.xy.qidx.gp, P.xy.ridx.gp};
    for (int d = 0; d < 2; ++d)
        HIP_TRY(hipMemcpyAsync(&nf[d], reinterpret_cast<const char*>(gp[d]) + offsetof(GridParams<T>, nonfinite), sizeof(int), hipMemcpyDeviceToHost, s));
    HIP_WAIT(s);
    return 0;
}
// ... and is it input the reference answers stably (see nonfinite_error)? Then the jobs take it from here on (row-based path).
template <typename T>
static bool pair_nonfinite_ok(PairState<T>& P, const int (&nf)[2]) {
    if ((nf[1] & kNfHard) || (P.two && (nf[0] & kNfHard))) return false;
    P.xy.bad_r = P.yx.bad_r = kNfHard; P.xy.bad_q = P.yx.bad_q = 0;
    return true;
}
template <typename T>
static bool pair_refused_nonfinite(const PairState<T>& P, const ResultBlock& h) { return ((h.counters[0][C_LARGE] | (P.two ? h.counters[1][C_LARGE] : 0)) & 4) != 0; }
// Redo a fused call's searches through the row-based path (everything the fused attempt left behind is reset).
template <typename T>
static int unfuse_and_research(pcu_hip_ctx* c, hipStream_t s, PairState<T>& P, pcu_hip_stats* st, bool skip_search = false) {
    P.fuse = FUSE_NONE; P.xy.fuse = P.yx.fuse = FUSE_NONE;
    HIP_TRY(hipMemsetAsync(P.cb, 0, sizeof(CallBlock), s));
    if (st) { st->n_passes = 0; }
    // skip_search: every direction of the fused attempt stopped at the balance check (skewed_everywhere) -- the same passes would stop
    // there again; the caller goes straight to search_finish's refit path with the counters it already holds
    return skip_search ? 0 : pair_search_enqueue(c, s, P, st);
}
// Did every direction of a failed fused attempt give up on the balance check alone?
template <typename T>
static bool skewed_everywhere(const PairState<T>& P, const ResultBlock& h) {
    for (int d = 0; d < (P.two ? 2 : 1); ++d) if (!h.counters[d][C_SKEW] || h.counters[d][C_LARGE]) return false;
    return true;
}
// The first step of search_finish's refit path (levels over the grid as built + the passes + the read-backs), enqueued without the
// host round trip, when the direction's counters say that this is what search_finish would do first.
template <typename T>
static bool skew_prelaunch_wanted(const pcu_hip_ctx* c, const SearchJob<T>& j, const int* hc) {
    static const bool off = getenv("PCU_HIP_REFIT_BASE") != nullptr || getenv("PCU_HIP_NO_SKEW_OVERLAP") != nullptr;
    return !off && !hc[C_LARGE] && hc[C_SKEW] && (!j.may_rescale || (hc[C_SKEW] == 2 && c->occ_scale[j.role] >= 1.0)) && !c->time_phases && !c->time_kernels;
}
template <typename T>
static int skew_prelaunch(pcu_hip_ctx* c, Arena& ar, hipStream_t s, SearchJob<T>& j, pcu_hip_stats* st, SkewPre& pre) {
    const int p0 = st ? st->n_passes : 0, b0 = st ? st->n_grid_builds : 0;
    // (no index_large_pass here: skew_prelaunch_wanted requires C_LARGE == 0, i.e. no deferred buckets in either index -- and the two directions run
    // this function side by side on two streams over the SAME two indexes with the roles swapped, where a placement pass would race with itself)
    j.skew_check = false;
    const double* hs_dev = nullptr;
    if (skew_add_levels(ar, s, j, st, j.ridx, &hs_dev)) return -1;
    if (search_enqueue(c, s, j, st)) return -1;
    pre.hs_dev = hs_dev;                         // (read back by the caller once BOTH directions are enqueued: a copy to pageable memory blocks the host)
    pre.on = true;
    if (st) { pre.d_passes = st->n_passes - p0; pre.d_builds = st->n_grid_builds - b0; }
    return 0;
}

// Sync + finish stragglers. Returns 1 if the epilogue must be re-enqueued, 0 if not, 3 if the call is to be restarted, <0 on error.
template <typename T>
static int pair_finish(pcu_hip_ctx* c, Arena& ar, hipStream_t s, PairState<T>& P, pcu_hip_stats* st, ResultBlock* host, bool copied_by_kernel = false,
                       bool host_given = false) {
    if (host_given) {}                          // (*host holds the counters to act on: nothing was enqueued since they were read)
    else if (!copied_by_kernel) { HIP_TRY(hipMemcpyAsync(c->h_pinned, P.rb, sizeof(ResultBlock), hipMemcpyDeviceToHost, s)); HIP_WAIT(s); }
    else if (wait_result_block(c, s)) return -1;
    if (!host_given) memcpy(host, c->h_pinned, sizeof(ResultBlock));
    // Both directions unbalanced (clustered clouds): their refits -- dozens of short launches each -- are enqueued side by side on the
    // context's two streams and share ONE host round trip, instead of one direction after the other (round 3: 2 x 0.8 ms).
    SkewPre pre[2];
    if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[pair_finish] two=%d wanted=%d,%d (skew %d,%d large %d,%d may %d,%d scale %.2f,%.2f time %d%d)\n", (int)P.two, (int)skew_prelaunch_wanted(c, P.xy, host->counters[0]), (int)skew_prelaunch_wanted(c, P.yx, host->counters[1]), host->counters[0][C_SKEW], host->counters[1][C_SKEW], host->counters[0][C_LARGE], host->counters[1][C_LARGE], (int)P.xy.may_rescale, (int)P.yx.may_rescale, c->occ_scale[0], c->occ_scale[1], (int)c->time_phases, (int)c->time_kernels);
    if (P.two && skew_prelaunch_wanted(c, P.xy, host->counters[0]) && skew_prelaunch_wanted(c, P.yx, host->counters[1])) {
        HIP_TRY(hipEventRecord(c->jev[0], s));
        HIP_TRY(hipStreamWaitEvent(c->aux_stream, c->jev[0], 0));
        if (skew_prelaunch(c, ar, s, P.xy, st, pre[0]) || skew_prelaunch(c, ar, c->aux_stream, P.yx, st, pre[1])) return -1;
        HIP_TRY(hipEventRecord(c->jev[1], c->aux_stream));
        HIP_TRY(hipStreamWaitEvent(s, c->jev[1], 0));
        // (into pinned memory: four copies in flight and one wait, instead of four blocking copies to pageable memory)
        static_assert(2 * (sizeof(double) * 2 + sizeof(int) * C_N) <= 64 * sizeof(int), "pair_finish's read-backs fit the second half of h_pinned");
        char* const hp = reinterpret_cast<char*>(c->h_pinned + 64);
        for (int d = 0; d < 2; ++d) {
            HIP_TRY(hipMemcpyAsync(hp + 16 * d, pre[d].hs_dev, 16, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(hp + 32 + sizeof(int) * C_N * d, (d ? P.yx : P.xy).sc.counters, sizeof(int) * C_N, hipMemcpyDeviceToHost, s));
        }
        HIP_WAIT(s);
        for (int d = 0; d < 2; ++d) { memcpy(pre[d].hs, hp + 16 * d, 16); memcpy(pre[d].hc_redo, hp + 32 + sizeof(int) * C_N * d, sizeof(int) * C_N); }
    }
    int r1 = search_finish(c, ar, s, P.xy, st, host->counters[0], &pre[0]);
    if (r1 < 0 || r1 == 3) return r1;
    int r2 = P.two ? search_finish(c, ar, s, P.yx, st, host->counters[1], &pre[1]) : 0;
    if (r2 < 0 || r2 == 3) return r2;
    return (r1 | r2) ? 1 : 0;       // (2 = "only tied rows changed" matters to k_nearest_neighbors only)
}

// Arg-max epilogue of one or both directions (+ copy of the result block to pinned host memory): one launch.
template <typename T>
static int argmax_enqueue(pcu_hip_ctx* c, hipStream_t s, PairState<T>& P, bool two_sided) {
    auto side = [&](const SearchJob<T>& j) {
        const int n = j.qidx.n;
        return ArgmaxSide<T>{j.out_d, j.row_out ? nullptr : j.qidx.sorted, j.out_i, n, std::min((n + kBlock - 1) / kBlock, kRedBlocksFused)};
    };
    const ArgmaxSide<T> a = side(P.xy);
    ArgmaxSide<T> b = a; b.n = 0; b.nb = 0;
    if (two_sided) b = side(P.yx);
    hipLaunchKernelGGL(k_argmax_pair<T>, dim3(a.nb + b.nb), dim3(kBlock), 0, s, a, b, P.pv, P.pi, P.res_v, P.res_ij,
                       reinterpret_cast<unsigned*>(P.rb->pad), reinterpret_cast<const int*>(P.rb), c->h_pinned, ++c->seq);
    HIP_TRY(hipGetLastError());
    return 0;
}

// A two-sided (or one-sided) k = 1 call between its enqueue half and its finish half. The batch entry points keep one of
// these in flight per lane (pcu_hip_ctx) so that the short kernels of independent pairs overlap on the GPU.
template <typename T>
struct PendingPair {
    PairState<T> P;
    Arena ar; Timer tm; hipStream_t s = nullptr;
    const T *x = nullptr, *y = nullptr; int64_t nx = 0, ny = 0;
    bool on_dev = false, squared = false, two_sided = true;
    unsigned flags = 0; int max_leaf = 10; pcu_hip_stats* st = nullptr;
    double p_norm = 2.0; int64_t *out_cxy = nullptr, *out_cyx = nullptr;      // chamfer
    int restarts = 0;                       // how often this call has been restarted at another grid resolution (PCU_RETRY)
};

template <typename T>
static int hausdorff_begin(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, bool two_sided, int max_leaf,
                           unsigned flags, void* stream, pcu_hip_stats* st, PendingPair<T>& pp) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (validate_sizes(nx, ny, "source", "targets")) return PCU_HIP_ERR_INVALID;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE, squared = flags & PCU_HIP_SQUARED;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    if (st) { const int builds = pp.restarts ? st->n_grid_builds : 0; memset(st, 0, sizeof *st); st->n_grid_builds = builds; }      // (a restarted call reports the builds of its abandoned attempts too)
    c->time_phases = flags & PCU_HIP_TIME_PHASES; c->time_kernels = flags & PCU_HIP_TIME_KERNELS;
    const double occ_x = call_occupancy(c, 1, 0), occ_y = call_occupancy(c, 1, 1);
    if (ctx_begin(c, pair_bytes<T>(nx, ny, occ_x, occ_y, on_dev))) return PCU_HIP_ERR_RUNTIME;
    pp.P.allow_rescale = pp.restarts < 2;
    if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[pair begin] restarts=%d nx=%lld ny=%lld occ_x=%.3f occ_y=%.3f\n", pp.restarts, (long long)nx, (long long)ny, occ_x, occ_y);
    pp.ar = Arena{c}; pp.tm = Timer{c, s, st}; pp.s = s; pp.x = x; pp.y = y; pp.nx = nx; pp.ny = ny;
    pp.on_dev = on_dev; pp.squared = squared; pp.two_sided = two_sided; pp.flags = flags; pp.max_leaf = max_leaf; pp.st = st;
    int rc = pair_setup(c, pp.ar, s, x, nx, y, ny, on_dev, squared, occ_x, occ_y, nullptr, nullptr, pp.P, pp.tm, st, two_sided, max_leaf, false, false, FUSE_ARGMAX);
    if (rc) { ctx_end(c); return rc < 0 ? rc : PCU_HIP_ERR_RUNTIME; }
    return 0;
}
template <typename T>
static int hausdorff_end(pcu_hip_ctx* c, PendingPair<T>& pp, T* out_d, int64_t* out_i, int64_t* out_j) {
    PairState<T>& P = pp.P; Arena& ar = pp.ar; Timer& tm = pp.tm; hipStream_t s = pp.s; pcu_hip_stats* st = pp.st;
    const bool two_sided = pp.two_sided; const unsigned flags = pp.flags;
    const bool tie_matters = !(flags & PCU_HIP_NO_TIE_ORDER);
    int rc = 0;
    do {
        ResultBlock host;
        bool done = false, skewed = false;
        if (P.fuse) {
            tm.mark(3);
            if ((rc = wait_result_block(c, s))) break;
            memcpy(&host, c->h_pinned, sizeof host);
            if ((rc = fused_wave_if_needed(c, s, P, st, host))) break;
            if (pair_refused_nonfinite(P, host)) {          // non-finite coordinates: row-based, if the reference has a stable answer
                int nf[2];
                if ((rc = pair_nonfinite_flags(s, P, nf))) break;
                if (!pair_nonfinite_ok(P, nf)) { rc = nonfinite_error(true); break; }
                if ((rc = pair_unlean(s, P)) || (rc = unfuse_and_research(c, s, P, st))) break;
            }
            else if (fused_ok(P, host, tie_matters)) {
                for (int d = 0; d < (two_sided ? 2 : 1); ++d) if (st) { st->n_escalated += host.counters[d][C_U1]; st->n_tie_flagged += host.counters[d][C_T1]; }
                done = true;
            } else if (fused_rescale(c, s, P, host)) { rc = PCU_RETRY; break; }
            else if ((rc = pair_unlean(s, P))) break;
            else if ((rc = fused_continue(c, ar, s, P, st, host, tie_matters)) != 0) { if (rc < 0) break; rc = 0; done = true; }
            else { skewed = skewed_everywhere(P, host); if ((rc = unfuse_and_research(c, s, P, st, skewed))) break; }
        }
        if (!done) {
            for (int attempt = 0; attempt < 2; ++attempt) {
                const bool given = attempt == 0 && skewed;       // no searches were re-run: no epilogue to run either, straight to the refit path
                if (!given && (rc = argmax_enqueue(c, s, P, two_sided))) break;
                tm.mark(3);
                if (attempt == 0) { rc = pair_finish(c, ar, s, P, st, &host, /*copied_by_kernel=*/true, given); if (rc == PCU_NONFINITE) rc = nonfinite_error(true); if (given && rc == 0) rc = 1; if (rc == 3) rc = PCU_RETRY; if (rc <= 0 || rc == PCU_RETRY) break; rc = 0; }   // syncs; 1 => redo epilogue
                else { HIP_TRY(hipMemcpyAsync(c->h_pinned, P.rb, sizeof(ResultBlock), hipMemcpyDeviceToHost, s)); HIP_WAIT(s); memcpy(&host, c->h_pinned, sizeof host); }
            }
            if (rc) break;
            // The value never depends on the order of exact ties, and (i, j) only does if the arg-max source row i itself
            // has tied nearest neighbours: only then is that direction's tie order resolved (kd_order.h) and j re-read.
            // Whether row i is in a direction's true-tie list is checked on the device (one launch, one 4-byte read-back).
            if (tie_matters && (P.xy.n_tt > 0 || (two_sided && P.yx.n_tt > 0))) {
                HIP_TRY(hipMemsetAsync(P.tie_hit, 0, 2 * sizeof(int), s));
                for (int dir = 0; dir < (two_sided ? 2 : 1); ++dir) {
                    SearchJob<T>& J = dir ? P.yx : P.xy;
                    if (J.n_tt <= 0) continue;
                    hipLaunchKernelGGL(k_tie_hit<T>, dim3(std::min((J.n_tt + kBlock - 1) / kBlock, 1024)), dim3(kBlock), 0, s,
                                       J.sc.tt, J.n_tt, J.qidx.sorted, P.res_ij + 2 * dir, P.tie_hit + dir);
                }
                HIP_TRY(hipGetLastError());
                int hit[2] = {0, 0};
                HIP_TRY(hipMemcpyAsync(hit, P.tie_hit, sizeof hit, hipMemcpyDeviceToHost, s));
                HIP_WAIT(s);
                bool redo = false;
                for (int dir = 0; dir < (two_sided ? 2 : 1) && !rc; ++dir) {
                    SearchJob<T>& J = dir ? P.yx : P.xy;
                    if (!hit[dir]) continue;
                    if (tie_order_resolve(c, ar, s, J, J.n_tt, st)) { rc = -1; break; }
                    redo = true;
                }
                if (rc) break;
                if (redo) {
                    if ((rc = argmax_enqueue(c, s, P, two_sided))) break;
                    HIP_TRY(hipMemcpyAsync(c->h_pinned, P.rb, sizeof(ResultBlock), hipMemcpyDeviceToHost, s));
                    HIP_WAIT(s);
                    memcpy(&host, c->h_pinned, sizeof host);
                }
            }
        }
        const T* hv = reinterpret_cast<const T*>(host.vals); const long long* hij = host.ij;
        const int nres = two_sided ? 2 : 1;
        for (int r = 0; r < nres; ++r) { out_d[r] = hv[r]; out_i[r] = hij[2 * r]; out_j[r] = hij[2 * r + 1]; }
        if (st) { st->n_queries = two_sided ? pp.nx + pp.ny : pp.nx; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 3); collect_kernel_times(c, st); }
    } while (0);
    ctx_end(c);
    if (rc == PCU_RETRY) return PCU_RETRY;
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename T>
static int hausdorff_impl(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, bool two_sided, int max_leaf,
                          T* out_d, int64_t* out_i, int64_t* out_j, unsigned flags, void* stream, pcu_hip_stats* st) {
    for (int restarts = 0;; ++restarts) {          // (restarts: occupancy rescale, see kRescaleAbove)
        PendingPair<T> pp; pp.restarts = restarts;
        if (int rc = hausdorff_begin(c, x, nx, y, ny, two_sided, max_leaf, flags, stream, st, pp)) return rc;
        const int rc = hausdorff_end(c, pp, out_d, out_i, out_j);
        if (rc != PCU_RETRY) return rc;
    }
}

static int pcode_of(double p) {
    if (p == 2.0) return P_TWO;
    if (p == 1.0) return P_ONE;
    if (isinf(p)) return p > 0 ? P_INF : P_NINF;
    if (p == 0.0) return P_ZERO;
    return P_GEN;
}

template <typename T>
static int chamfer_begin(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, double p_norm, int max_leaf,
                         int64_t* out_cxy, int64_t* out_cyx, unsigned flags, void* stream, pcu_hip_stats* st, PendingPair<T>& pp) {
    g_hprof.mark(0);
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (validate_sizes(nx, ny, "query_points", "dataset_points")) return PCU_HIP_ERR_INVALID;
    if (isnan(p_norm)) return fail(PCU_HIP_ERR_INVALID, "p_norm is NaN");
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    if (st) { const int builds = pp.restarts ? st->n_grid_builds : 0; memset(st, 0, sizeof *st); st->n_grid_builds = builds; }      // (a restarted call reports the builds of its abandoned attempts too)
    c->time_phases = flags & PCU_HIP_TIME_PHASES; c->time_kernels = flags & PCU_HIP_TIME_KERNELS;
    const double occ_x = call_occupancy(c, 1, 0), occ_y = call_occupancy(c, 1, 1);
    if (ctx_begin(c, pair_bytes<T>(nx, ny, occ_x, occ_y, on_dev))) return PCU_HIP_ERR_RUNTIME;
    pp.P.allow_rescale = pp.restarts < 2;
    if (getenv("PCU_HIP_DEBUG_SKEW")) fprintf(stderr, "[pair begin] restarts=%d nx=%lld ny=%lld occ_x=%.3f occ_y=%.3f\n", pp.restarts, (long long)nx, (long long)ny, occ_x, occ_y);
    pp.ar = Arena{c}; pp.tm = Timer{c, s, st}; pp.s = s; pp.x = x; pp.y = y; pp.nx = nx; pp.ny = ny;
    pp.on_dev = on_dev; pp.two_sided = true; pp.flags = flags; pp.max_leaf = max_leaf; pp.st = st;
    pp.p_norm = p_norm; pp.out_cxy = out_cxy; pp.out_cyx = out_cyx;
    // Which of two exactly tied neighbours is picked changes a direction's contribution only through the returned
    // indices, or through a p != 2 norm of the difference vector; the p = 2 value is the tied distance itself.
    const bool tie_any = !(flags & PCU_HIP_NO_TIE_ORDER);
    const bool tie_xy = tie_any && (out_cxy != nullptr || p_norm != 2.0), tie_yx = tie_any && (out_cyx != nullptr || p_norm != 2.0);
    // p = 2 without indices: the value is the sum of the nearest-neighbour distances -> fused epilogue, no result rows
    const int fuse = (p_norm == 2.0 && !out_cxy && !out_cyx) ? FUSE_SUM : FUSE_NONE;
    int rc = pair_setup(c, pp.ar, s, x, nx, y, ny, on_dev, /*squared=*/false, occ_x, occ_y, on_dev ? (long long*)out_cxy : nullptr, on_dev ? (long long*)out_cyx : nullptr,
                        pp.P, pp.tm, st, true, max_leaf, tie_xy, tie_yx, fuse);
    g_hprof.mark(2);
    if (rc) { ctx_end(c); return rc < 0 ? rc : PCU_HIP_ERR_RUNTIME; }
    return 0;
}
template <typename T>
static int chamfer_end(pcu_hip_ctx* c, PendingPair<T>& pp, double* out_mean2) {
    PairState<T>& P = pp.P; Arena& ar = pp.ar; Timer& tm = pp.tm; hipStream_t s = pp.s; pcu_hip_stats* st = pp.st;
    const int64_t nx = pp.nx, ny = pp.ny; const bool on_dev = pp.on_dev;
    int64_t *out_cxy = pp.out_cxy, *out_cyx = pp.out_cyx; const double p_norm = pp.p_norm;
    int rc = 0;
    // A NaN coordinate in either cloud makes the reference's value NaN whatever its kd-tree does with it: the row's own term
    // norm(other[corr] - row) is NaN for every corr (__init__.py:112-113), and so are the mean and the sum (:114-115) -- for every ord but 0,
    // which counts NaN as a non-zero. So the VALUE is stable although the correspondences are not: returned as the reference returns it
    // (no indices asked for). Everything else a search refuses (nonfinite_error) has no stable answer.
    bool nan_result = false;
    auto nan_rule = [&](const int (&nf)[2]) { return ((nf[0] | nf[1]) & kNfNaN) && !out_cxy && !out_cyx && p_norm != 0.0; };
    auto refused = [&]() -> int {           // a row-based search refused its dataset: the NaN value, or the error
        int nf[2];
        if (int r = pair_nonfinite_flags(s, P, nf)) return r;
        if (nan_rule(nf)) { nan_result = true; return 0; }
        return nonfinite_error(true);
    };
    do {
        ResultBlock host;
        bool done = false, skewed = false;
        if (P.fuse) {
            tm.mark(3);
            if ((rc = wait_result_block(c, s))) break;
            g_hprof.mark(3);
            memcpy(&host, c->h_pinned, sizeof host);
            if ((rc = fused_wave_if_needed(c, s, P, st, host))) break;
            if (pair_refused_nonfinite(P, host)) {          // non-finite coordinates: see below (nan_rule) and hausdorff_end
                int nf[2];
                if ((rc = pair_nonfinite_flags(s, P, nf))) break;
                if (!pair_nonfinite_ok(P, nf)) { if (nan_rule(nf)) { rc = 0; nan_result = true; } else rc = nonfinite_error(true); break; }
                if ((rc = pair_unlean(s, P)) || (rc = unfuse_and_research(c, s, P, st))) break;
            }
            else if (fused_ok(P, host, false)) {
                for (int d = 0; d < 2; ++d) if (st) { st->n_escalated += host.counters[d][C_U1]; st->n_tie_flagged += host.counters[d][C_T1]; }
                done = true;
            } else if (fused_rescale(c, s, P, host)) { rc = PCU_RETRY; break; }
            else if ((rc = pair_unlean(s, P))) break;
            else if ((rc = fused_continue(c, ar, s, P, st, host, false)) != 0) { if (rc < 0) break; rc = 0; done = true; }
            else { skewed = skewed_everywhere(P, host); if ((rc = unfuse_and_research(c, s, P, st, skewed))) break; }
        }
        if (!done) {
            const int pc = pcode_of(p_norm);
            // __init__.py:112: norm(x[corrs_y_to_x] - y).mean() -> queries y, targets x ; :113 the other way round
            const int nbx = std::min((int)((nx + kBlock - 1) / kBlock), kRedBlocksFused), nby = std::min((int)((ny + kBlock - 1) / kBlock), kRedBlocksFused);
            for (int attempt = 0; attempt < 2; ++attempt) {
                const bool given = attempt == 0 && skewed;       // no searches were re-run: no epilogue to run either, straight to the refit path
                if (given) { rc = pair_finish(c, ar, s, P, st, &host, true, true); if (rc == PCU_NONFINITE) { rc = refused(); break; } if (rc == 0) rc = 1; if (rc == 3) rc = PCU_RETRY; if (rc <= 0 || rc == PCU_RETRY) break; rc = 0; continue; }
                // both directions' norms + final sums + the copy of the result block to pinned host memory: one launch
                // (the rows are in the caller's row order: the queries are the clouds themselves)
                const PnormSide<T> sx{nullptr, P.dx, P.dy, P.xy.out_i, P.xy.out_d, (int)nx, nbx, P.xy.sc.counters + C_SKEW, (long long)ny},
                                   sy{nullptr, P.dy, P.dx, P.yx.out_i, P.yx.out_d, (int)ny, nby, P.yx.sc.counters + C_SKEW, (long long)nx};
                hipLaunchKernelGGL(k_pnorm_pair<T>, dim3(nbx + nby), dim3(kBlock), 0, s, sx, sy, pc, p_norm, P.pd, P.res_s,
                                   reinterpret_cast<unsigned*>(P.rb->pad), reinterpret_cast<const int*>(P.rb), c->h_pinned, ++c->seq);
                HIP_TRY(hipGetLastError());
                tm.mark(3);
                if (attempt == 0) { rc = pair_finish(c, ar, s, P, st, &host, /*copied_by_kernel=*/true); if (rc == PCU_NONFINITE) { rc = refused(); break; } if (rc == 3) rc = PCU_RETRY; if (rc <= 0 || rc == PCU_RETRY) break; rc = 0; }
                else { HIP_TRY(hipMemcpyAsync(c->h_pinned, P.rb, sizeof(ResultBlock), hipMemcpyDeviceToHost, s)); HIP_WAIT(s); memcpy(&host, c->h_pinned, sizeof host); }
            }
            if (rc || nan_result) break;
            if (!on_dev) {
                if (out_cxy) HIP_TRY(hipMemcpyAsync(out_cxy, P.xy.out_i, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
                if (out_cyx) HIP_TRY(hipMemcpyAsync(out_cyx, P.yx.out_i, (size_t)ny * 8, hipMemcpyDeviceToHost, s));
                HIP_WAIT(s);
            }
        }
        const double* hs = host.sums;
        out_mean2[0] = hs[0] / (double)nx;
        out_mean2[1] = hs[1] / (double)ny;
        if (st) { st->n_queries = nx + ny; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 3); collect_kernel_times(c, st); }
    } while (0);
    if (!rc && nan_result) { out_mean2[0] = out_mean2[1] = std::numeric_limits<double>::quiet_NaN(); if (st) st->n_queries = nx + ny; }
    ctx_end(c);
    g_hprof.mark(4); g_hprof.done();
    if (rc == PCU_RETRY) return PCU_RETRY;
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename T>
static int chamfer_impl(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, double p_norm, int max_leaf, double* out_mean2,
                        int64_t* out_cxy, int64_t* out_cyx, unsigned flags, void* stream, pcu_hip_stats* st) {
    for (int restarts = 0;; ++restarts) {
        PendingPair<T> pp; pp.restarts = restarts;
        if (int rc = chamfer_begin(c, x, nx, y, ny, p_norm, max_leaf, out_cxy, out_cyx, flags, stream, st, pp)) return rc;
        const int rc = chamfer_end(c, pp, out_mean2);
        if (rc != PCU_RETRY) return rc;
    }
}


// ------------------------------------------------------------------------------------------------ normals
static int aux_reserve(pcu_hip_ctx* c, size_t bytes) {
    if (bytes > c->aux_cap) {
        if (c->aux) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->aux)); c->aux = nullptr; c->aux_cap = 0; }
        const size_t cap = align_up(bytes + (bytes >> 3), 1 << 20);
        HIP_TRY(hipMalloc((void**)&c->aux, cap));
        c->aux_cap = cap;
    }
    return 0;
}
static int validate_normals_input(int64_t n, int64_t n_dirs) {
    if (n <= 0) return fail(PCU_HIP_ERR_INVALID, "Invalid point set with zero elements: points must have shape (n, 3), but got ot points.shape = (%lld, 3).", (long long)n);
    if (n > 0x07fffff0ll) return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^27-16 rows are not supported");
    if (n_dirs != 0 && n_dirs != n)
        return fail(PCU_HIP_ERR_INVALID, "Invalid view directions does not match the number of points. If view directions are passed in, they must have the same "
                    "shape as points. Got points.shape = (%lld, 3), and view_dirs.shape = (%lld, 3).", (long long)n, (long long)n_dirs);
    return 0;
}
// estimate_point_cloud_normals_knn_internal (src/point_cloud_normals.cpp:375-411): self-KNN (the search of this library, so the
// neighbour sets and their order are the reference's), then one plane fit per point (normals.h).
template <typename T>
static int normals_knn_impl(pcu_hip_ctx* c, const T* points, int64_t n, const T* dirs, int k, int max_leaf, double drop,
                            T* out_n, uint8_t* out_keep, unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (k <= 0) return fail(PCU_HIP_ERR_INVALID, "Invalid number of neighbors (%d) must be greater than 0.", k);
    if (validate_normals_input(n, dirs ? n : 0)) return PCU_HIP_ERR_INVALID;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    auto al = [](size_t v) { return align_up(v, 256); };
    const size_t b_pts = al((size_t)n * 3 * sizeof(T));
    const size_t need = (on_dev ? 0 : 2 * b_pts) + al((size_t)n * k * 8) + al((size_t)n * k * sizeof(T)) + (on_dev ? 0 : b_pts + al((size_t)n));
    if (aux_reserve(c, need + 4096)) return PCU_HIP_ERR_RUNTIME;
    char* p = c->aux;
    auto take = [&](size_t bytes) { char* r = p; p += al(bytes); return r; };
    const T* d_pts = points; const T* d_dirs = dirs;
    if (!on_dev) {
        T* t = (T*)take((size_t)n * 3 * sizeof(T)); HIP_TRY(hipMemcpyAsync(t, points, (size_t)n * 3 * sizeof(T), hipMemcpyHostToDevice, s)); d_pts = t;
        if (dirs) { T* u = (T*)take((size_t)n * 3 * sizeof(T)); HIP_TRY(hipMemcpyAsync(u, dirs, (size_t)n * 3 * sizeof(T), hipMemcpyHostToDevice, s)); d_dirs = u; }
    }
    long long* d_nbr = (long long*)take((size_t)n * k * 8);
    T* d_dist = (T*)take((size_t)n * k * sizeof(T));
    T* d_out = out_n; uint8_t* d_keep = out_keep;
    if (!on_dev) { d_out = (T*)take((size_t)n * 3 * sizeof(T)); d_keep = (uint8_t*)take((size_t)n); }
    const unsigned kflags = (flags | PCU_HIP_PTRS_ON_DEVICE | PCU_HIP_SQUARED | PCU_HIP_STREAM_GIVEN) & ~(unsigned)(PCU_HIP_TIME_PHASES | PCU_HIP_TIME_KERNELS);
    if (int rc = knn_impl<T>(c, d_pts, n, d_pts, n, k, max_leaf, d_dist, (int64_t*)d_nbr, kflags, (void*)s, st)) return rc;
    const NormalsKnnArgs<T> a{d_pts, d_dirs, d_nbr, (int)n, k, drop, d_out, d_keep};
    hipLaunchKernelGGL(k_normals_knn<T>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
    HIP_TRY(hipGetLastError());
    if (!on_dev) {
        HIP_TRY(hipMemcpyAsync(out_n, d_out, (size_t)n * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(out_keep, d_keep, (size_t)n, hipMemcpyDeviceToHost, s));
    }
    HIP_WAIT(s);
    return 0;
}
// estimate_point_cloud_normals_ball_internal (:305-372): every point inside the ball (radiusSearch semantics of :75, see normals.h).
template <typename T>
static int normals_ball_impl(pcu_hip_ctx* c, const T* points, int64_t n, const T* dirs, double ball_radius, int min_pts, int max_pts,
                             int weight_rbf, double drop, T* out_n, uint8_t* out_keep, unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (!(ball_radius > 0.0)) return fail(PCU_HIP_ERR_INVALID, "Invalid radius (%f) must be greater than 0.", ball_radius);
    if (min_pts < 3) return fail(PCU_HIP_ERR_INVALID, "Invalid min_pts_per_ball (%d) must be greater than 3.", min_pts);
    if (max_pts > 0 && max_pts < 3) return fail(PCU_HIP_ERR_INVALID, "Invalid max_pts_per_ball (%d) must either be negative (no max) or a number greater than 3.", max_pts);
    if (validate_normals_input(n, dirs ? n : 0)) return PCU_HIP_ERR_INVALID;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    c->time_phases = false; c->time_kernels = false;
    const double occ = 8.0;
    size_t need = index_bytes<T>(n, occ) + 8192;
    if (!on_dev) need += 3 * align_up((size_t)n * 3 * sizeof(T), 256) + align_up((size_t)n, 256);
    if (ctx_begin(c, need)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T *d_pts, *d_dirs = dirs;
        if ((rc = stage_in(ar, points, n, on_dev, s, &d_pts))) break;
        if (dirs && (rc = stage_in(ar, dirs, n, on_dev, s, &d_dirs))) break;
        T* d_out = out_n; uint8_t* d_keep = out_keep;
        if (!on_dev) { if ((rc = aalloc(ar, &d_out, (size_t)n * 3))) break; if ((rc = aalloc(ar, &d_keep, (size_t)n))) break; }
        GridIndex<T> gi;
        if ((rc = index_alloc(ar, gi, n, occ, false, /*allow_bucketed=*/false))) break;
        const T radius_t = (T)ball_radius;                          // RadiusResultSet<DistanceType = T>(radius, ...)
        gi.h_want = sqrt((double)radius_t) / 1.98;                  // cells of about half the true search radius: 5^3 cells per point
        if ((rc = index_build(gi, d_pts, occ, s))) break;
        if (st) { st->n_grid_builds = 1; st->n_queries = n; st->n_passes = 1; }
        const NormalsBallArgs<T> a{gi.gp, gi.sorted, gi.cell_start, d_dirs, (int)n, radius_t, ball_radius, min_pts, max_pts, weight_rbf, drop, d_out, d_keep};
        hipLaunchKernelGGL(k_normals_ball<T>, dim3((unsigned)((n + kBlock - 1) / kBlock)), dim3(kBlock), 0, s, a);
        HIP_TRY(hipGetLastError());
        if (!on_dev) {
            HIP_TRY(hipMemcpyAsync(out_n, d_out, (size_t)n * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_keep, d_keep, (size_t)n, hipMemcpyDeviceToHost, s));
        }
        HIP_WAIT(s);
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ------------------------------------------------------------------------------------------------ batches of pairs
// BASELINE config 4 (256 independent 256k-vs-256k pairs): clouds this small leave every kernel at the launch floor, so a
// batch keeps several pairs in flight, each on a lane of its own (stream + workspace + pinned result block): the fused calls
// above need no host decision between their enqueue and their result, so the host enqueues pair p + 1 .. p + L - 1 while
// pair p runs and the short kernels of different pairs overlap on the GPU. Results are written in pair order.
static int batch_lanes(pcu_hip_ctx* c, int n_pairs, void* stream, unsigned flags) {
    const int want = std::max(1, std::min(std::min(c->n_lanes_wanted, 16), n_pairs));
    while ((int)c->lanes.size() < want) {
        pcu_hip_ctx* l = nullptr;
        if (pcu_hip_ctx_create(c->device, &l)) return -1;
        l->occupancy = c->occupancy; l->occ_scale[0] = c->occ_scale[0]; l->occ_scale[1] = c->occ_scale[1];
        c->lanes.push_back(l);
    }
    for (pcu_hip_ctx* l : c->lanes) l->occupancy = c->occupancy;
    // inputs produced on the caller's stream must be complete before a lane reads them
    if ((flags & PCU_HIP_PTRS_ON_DEVICE) && (stream || (flags & PCU_HIP_STREAM_GIVEN))) {
        if (!c->batch_ev) HIP_TRY(hipEventCreateWithFlags(&c->batch_ev, hipEventDisableTiming));
        HIP_TRY(hipEventRecord(c->batch_ev, (hipStream_t)stream));
        for (int i = 0; i < want; ++i) HIP_TRY(hipStreamWaitEvent(c->lanes[i]->own_stream, c->batch_ev, 0));
    }
    return want;
}
static void stats_add(pcu_hip_stats* tot, const pcu_hip_stats& s) {
    if (!tot) return;
    tot->n_queries += s.n_queries; tot->n_escalated += s.n_escalated; tot->n_tie_flagged += s.n_tie_flagged; tot->n_tie_true += s.n_tie_true;
    tot->n_passes += s.n_passes; tot->n_grid_builds += s.n_grid_builds;
}
template <typename T, typename Begin, typename End>
static int batch_run(pcu_hip_ctx* c, int n_pairs, unsigned flags, void* stream, pcu_hip_stats* st, Begin begin, End end) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (n_pairs < 0) return fail(PCU_HIP_ERR_INVALID, "negative number of pairs");
    if (st) memset(st, 0, sizeof *st);
    if (n_pairs == 0) return 0;
    // One host thread drives all lanes. (Several host threads, each with lanes of its own, were measured in round 2 on 32 pairs of
    // 262k points: what counts is the number of pairs in flight -- 3 is the optimum either way, 55.9 vs 58.8 us per pair; removed.)
    const int L = batch_lanes(c, n_pairs, stream, flags);
    if (L < 0) return PCU_HIP_ERR_RUNTIME;
    const unsigned lflags = flags & ~(unsigned)(PCU_HIP_STREAM_GIVEN | PCU_HIP_TIME_PHASES | PCU_HIP_TIME_KERNELS);
    std::vector<PendingPair<T>> pend((size_t)L);
    std::vector<pcu_hip_stats> lst((size_t)L);
    std::vector<int> cur((size_t)L, -1);
    int rc = 0; std::string err;
    int k = 0;
    for (int it = 0; it < n_pairs + L; ++it) {
        const int lane = it % L;
        if (cur[lane] >= 0) {
            int r = end(c->lanes[lane], pend[lane], cur[lane]);
            for (int restarts = 1; r == PCU_RETRY; ++restarts) {         // occupancy rescale: the lane's context has a new scale
                pend[lane] = PendingPair<T>(); pend[lane].restarts = restarts;
                pcu_hip_stats again;
                r = begin(c->lanes[lane], pend[lane], cur[lane], lflags, &again);
                if (!r) r = end(c->lanes[lane], pend[lane], cur[lane]);
            }
            if (r && !rc) { rc = r; err = g_err; }
            stats_add(st, lst[lane]);
            cur[lane] = -1;
        }
        if (k < n_pairs && !rc) {
            const int p = k++;
            pend[lane] = PendingPair<T>();
            const int r = begin(c->lanes[lane], pend[lane], p, lflags, &lst[lane]);
            if (r) { rc = r; err = g_err; } else cur[lane] = p;
        }
    }
    if (rc) g_err = err;
    return rc;
}

// ------------------------------------------------------------------------------------------------ persistent index
template <typename T> static GridIndex<T>& index_grid_mut(pcu_hip_index* p);
template <> GridIndex<float>& index_grid_mut<float>(pcu_hip_index* p) { return p->g32; }
template <> GridIndex<double>& index_grid_mut<double>(pcu_hip_index* p) { return p->g64; }

static void index_free(pcu_hip_index* p) {
    if (!p) return;
    if (p->pts) (void)hipFree(p->pts);
    if (p->mem) (void)hipFree(p->mem);
    delete p;
}
template <typename T>
static int index_create_impl(pcu_hip_ctx* c, const T* dataset, int64_t nr, int k_hint, unsigned flags, void* stream, pcu_hip_index** out) {
    if (!c || !out) return fail(PCU_HIP_ERR_INVALID, "null context / output");
    *out = nullptr;
    if (nr <= 0) return fail(PCU_HIP_ERR_INVALID, "Invalid input set with zero elements: dataset_points must have shape (m, 3). Got dataset_points.shape = (%lld, 3).", (long long)nr);
    if (nr > 0x07fffff0ll) return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^27-16 rows are not supported");
    if (k_hint <= 0) k_hint = 1;
    if (k_hint > kMaxK) k_hint = kMaxK;
    hipStream_t s = (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
    pcu_hip_index* p = new pcu_hip_index();
    p->elem_size = (int)sizeof(T); p->device = c->device; p->n = nr;
    p->occ = c->occupancy > 0 ? c->occupancy : default_occupancy(k_hint);
    int rc = 0;
    do {
        if (hipMalloc(&p->pts, (size_t)nr * 3 * sizeof(T)) != hipSuccess) { rc = fail(PCU_HIP_ERR_RUNTIME, "out of device memory for the dataset copy"); break; }
        if (hipMemcpyAsync(p->pts, dataset, (size_t)nr * 3 * sizeof(T), (flags & PCU_HIP_PTRS_ON_DEVICE) ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice, s) != hipSuccess) {
            rc = fail(PCU_HIP_ERR_RUNTIME, "copy of the dataset failed"); break;
        }
        const size_t bytes = index_bytes<T>(nr, p->occ) + 4096;
        if (hipMalloc(&p->mem, bytes) != hipSuccess) { rc = fail(PCU_HIP_ERR_RUNTIME, "out of device memory for the index"); break; }
        pcu_hip_ctx holder;                         // only its arena fields are used: a bump allocator over the index's own block
        holder.device = c->device; holder.arena = static_cast<char*>(p->mem); holder.arena_cap = bytes; holder.arena_off = 0;
        Arena ar{&holder};
        GridIndex<T>& g = index_grid_mut<T>(p);
        if (index_alloc(ar, g, nr, p->occ) || !holder.extra.empty()) {
            for (void* q : holder.extra) (void)hipFree(q);
            rc = fail(PCU_HIP_ERR_RUNTIME, "internal: index block too small"); break;
        }
        if ((rc = index_build<T>(g, static_cast<const T*>(p->pts), p->occ, s))) break;
        if ((rc = check_nonfinite<T>(g.gp, kNfNaN | kNfBothInf, s))) break;
    } while (0);
    if (rc) { index_free(p); return rc < 0 ? rc : PCU_HIP_ERR_RUNTIME; }
    *out = p;
    return 0;
}

// Diagnostic hook for tests: builds the nanoflann-faithful kd-tree of `pts` on the GPU and returns its
// permutation (vAcc) and node count, so the tree itself can be compared with the oracle's.
template <typename T>
static int debug_kd(pcu_hip_ctx* c, const T* pts, int64_t n, int leaf_max, int64_t* out_vacc, int64_t* out_nnodes) {
    if (!c || n <= 0) return fail(PCU_HIP_ERR_INVALID, "bad arguments");
    hipStream_t s = c->own_stream;
    if (ctx_begin(c, index_bytes<T>(n, 2.0) + (size_t)n * 3 * sizeof(T) + 8192)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T* dp;
        if ((rc = stage_in(ar, pts, n, false, s, &dp))) break;
        GridIndex<T> gi;
        if ((rc = index_alloc(ar, gi, n, 2.0))) break;
        if ((rc = index_build(gi, dp, 2.0, s))) break;
        KdBuild<T> b; int* err = nullptr; int levels = 0, nn = 0;
        if ((rc = kd_build_device(c, ar, s, dp, (int)n, gi.gp, leaf_max > 0 ? leaf_max : 10, b, &err, &levels, &nn))) break;
        std::vector<Pt4<T>> h((size_t)n);
        HIP_TRY(hipMemcpy(h.data(), b.E, (size_t)n * sizeof(Pt4<T>), hipMemcpyDeviceToHost));
        for (int64_t i = 0; i < n; ++i) out_vacc[i] = (int64_t)h[(size_t)i].idx;
        *out_nnodes = nn;
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

#include "voxel_host.h"

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* pcu_hip_last_error(void) { return g_err.c_str(); }
const char* pcu_hip_version(void) { return "pcu_hip 0.1 (gfx950)"; }

int pcu_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

void pcu_hip_cancel(void) { cancel_request(PCU_HIP_CANCEL_BY_REQUEST); }
int pcu_hip_cancel_source(void) { return g_cancel_source.load(std::memory_order_relaxed); }
int pcu_hip_watch_sigint(int enable) {
    if (enable && !g_sigint_watched.exchange(1)) {
        if (sigint_install() != 0) { g_sigint_watched.store(0); return fail(PCU_HIP_ERR_RUNTIME, "sigaction(SIGINT) failed"); }
    } else if (!enable && g_sigint_watched.exchange(0)) {
        struct sigaction cur;
        // (only if ours is still the installed handler: somebody who installed theirs after us keeps it)
        if (sigaction(SIGINT, nullptr, &cur) == 0 && (cur.sa_flags & SA_SIGINFO) && cur.sa_sigaction == pcu_on_sigint) (void)sigaction(SIGINT, &g_prev_sigint, nullptr);
    }
    return 0;
}
int pcu_hip_ctx_create(int device, pcu_hip_ctx** out_ctx) {
    if (!out_ctx) return fail(PCU_HIP_ERR_INVALID, "null out_ctx");
    int n = pcu_hip_device_count();
    if (n <= 0) return fail(PCU_HIP_ERR_NO_DEVICE, "no HIP device visible: the gfx950 path has no CPU fallback");
    if (device < 0 || device >= n) return fail(PCU_HIP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    DeviceGuard dg(device);
    if (!g_cancel_mirror.load(std::memory_order_acquire)) {
        unsigned* m = nullptr;
        HIP_TRY(hipHostMalloc((void**)&m, 64, hipHostMallocPortable | hipHostMallocMapped | hipHostMallocCoherent));
        *m = g_cancel_gen.load(std::memory_order_relaxed);
        unsigned* expected = nullptr;
        if (!g_cancel_mirror.compare_exchange_strong(expected, m)) (void)hipHostFree(m);        // (another thread's context was first)
        else __atomic_store_n(m, g_cancel_gen.load(std::memory_order_acquire), __ATOMIC_RELEASE);
    }
    pcu_hip_ctx* c = new pcu_hip_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    HIP_TRY(hipStreamCreateWithFlags(&c->aux_stream, hipStreamNonBlocking));
    {   // The speculative top of the tie-order tree runs beside a call's searches on a stream of its own. PCU_HIP_SPEC_PRIORITY=1 gives it the
        // highest stream priority: the tree top then ends 0.12 ms after config 3's searches instead of 0.3 -- but the mere existence of a
        // prioritised stream in the process slows calls that alternate two ordinary streams (config 4: 1.41 -> 1.9-2.1 ms), so it is off.
        int lo_p = 0, hi_p = 0;
        const char* e = getenv("PCU_HIP_SPEC_PRIORITY");
        if (e && atoi(e) != 0 && hipDeviceGetStreamPriorityRange(&lo_p, &hi_p) == hipSuccess && hi_p != lo_p)
            HIP_TRY(hipStreamCreateWithPriority(&c->spec_stream, hipStreamNonBlocking, hi_p));
        else
            HIP_TRY(hipStreamCreateWithFlags(&c->spec_stream, hipStreamNonBlocking));
    }
    for (auto& e : c->jev) HIP_TRY(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->kev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc((void**)&c->h_pinned, 128 * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));   // kernels write the result block into the first 64 words; [64, 128): pair_finish's read-backs
    memset(c->h_pinned, 0, 128 * sizeof(int));
    HIP_TRY(hipMalloc((void**)&c->tickets, 64 * sizeof(unsigned)));
    HIP_TRY(hipMemset(c->tickets, 0, 64 * sizeof(unsigned)));
    HIP_TRY(hipMalloc((void**)&c->geo.dev, 512));
    HIP_TRY(hipMalloc((void**)&c->fill2, 2 * (size_t)kFillWords * sizeof(unsigned long long)));
    HIP_TRY(hipMemset(c->fill2, 0, 2 * (size_t)kFillWords * sizeof(unsigned long long)));
    *out_ctx = c;
    return 0;
}
void pcu_hip_ctx_destroy(pcu_hip_ctx* c) {
    if (!c) return;
    DeviceGuard dg(c->device);
    (void)hipDeviceSynchronize();
    ctx_end(c);
    for (pcu_hip_ctx* l : c->lanes) pcu_hip_ctx_destroy(l);
    c->lanes.clear();
    if (c->batch_ev) (void)hipEventDestroy(c->batch_ev);
    if (c->aux) (void)hipFree(c->aux);
    if (c->tickets) (void)hipFree(c->tickets);
    if (c->fill2) (void)hipFree(c->fill2);
    if (c->geo.dev) (void)hipFree(c->geo.dev);
    if (c->arena) (void)hipFree(c->arena);
    for (hipEvent_t e : {c->kd_spec.ev_fork, c->kd_spec.ev_init, c->kd_spec.ev_done}) if (e) (void)hipEventDestroy(e);
    kd_graph_drop(c);
    if (c->kd_ws) (void)hipFree(c->kd_ws);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->kev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->cev) if (e) (void)hipEventDestroy(e);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    if (c->aux_stream) (void)hipStreamDestroy(c->aux_stream);
    if (c->spec_stream) (void)hipStreamDestroy(c->spec_stream);
    for (auto& e : c->jev) if (e) (void)hipEventDestroy(e);
    delete c;
}
int pcu_hip_ctx_set_cell_occupancy(pcu_hip_ctx* c, double ppc) { if (!c) return fail(PCU_HIP_ERR_INVALID, "null context"); c->occupancy = ppc; return 0; }
int64_t pcu_hip_ctx_workspace_bytes(pcu_hip_ctx* c) { return c ? (int64_t)c->arena_cap : 0; }

int pcu_hip_knn_f32(pcu_hip_ctx* c, const float* q, int64_t nq, const float* r, int64_t nr, int k, int max_leaf, float* od, int64_t* oi,
                    unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(knn_impl<float>(c, q, nq, r, nr, k, max_leaf, od, oi, flags, stream, st)); }
int pcu_hip_knn_f64(pcu_hip_ctx* c, const double* q, int64_t nq, const double* r, int64_t nr, int k, int max_leaf, double* od, int64_t* oi,
                    unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(knn_impl<double>(c, q, nq, r, nr, k, max_leaf, od, oi, flags, stream, st)); }

int pcu_hip_one_sided_hausdorff_f32(pcu_hip_ctx* c, const float* a, int64_t na, const float* b, int64_t nb, int max_leaf, float* od, int64_t* oi, int64_t* oj,
                                    unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(hausdorff_impl<float>(c, a, na, b, nb, false, max_leaf, od, oi, oj, flags, stream, st)); }
int pcu_hip_one_sided_hausdorff_f64(pcu_hip_ctx* c, const double* a, int64_t na, const double* b, int64_t nb, int max_leaf, double* od, int64_t* oi, int64_t* oj,
                                    unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(hausdorff_impl<double>(c, a, na, b, nb, false, max_leaf, od, oi, oj, flags, stream, st)); }
int pcu_hip_hausdorff_f32(pcu_hip_ctx* c, const float* a, int64_t na, const float* b, int64_t nb, int max_leaf, float* od, int64_t* oi, int64_t* oj,
                          unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(hausdorff_impl<float>(c, a, na, b, nb, true, max_leaf, od, oi, oj, flags, stream, st)); }
int pcu_hip_hausdorff_f64(pcu_hip_ctx* c, const double* a, int64_t na, const double* b, int64_t nb, int max_leaf, double* od, int64_t* oi, int64_t* oj,
                          unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(hausdorff_impl<double>(c, a, na, b, nb, true, max_leaf, od, oi, oj, flags, stream, st)); }

int pcu_hip_chamfer_f32(pcu_hip_ctx* c, const float* x, int64_t nx, const float* y, int64_t ny, double p, int max_leaf, double* om, int64_t* cxy, int64_t* cyx,
                        unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(chamfer_impl<float>(c, x, nx, y, ny, p, max_leaf, om, cxy, cyx, flags, stream, st)); }
int pcu_hip_chamfer_f64(pcu_hip_ctx* c, const double* x, int64_t nx, const double* y, int64_t ny, double p, int max_leaf, double* om, int64_t* cxy, int64_t* cyx,
                        unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(chamfer_impl<double>(c, x, nx, y, ny, p, max_leaf, om, cxy, cyx, flags, stream, st)); }

#define PCU_BATCH_HAUSDORFF(SUF, T)                                                                                                         \
int pcu_hip_hausdorff_batch_##SUF(pcu_hip_ctx* c, int n_pairs, const T* const* xs, const int64_t* nxs, const T* const* ys, const int64_t* nys,   \
                                  int max_leaf, T* out_d2, int64_t* out_i2, int64_t* out_j2, unsigned flags, void* stream, pcu_hip_stats* st) {  \
    CallGuard dg(c);                                                                                                     \
    return abi_rc(batch_run<T>(c, n_pairs, flags, stream, st,                                                                                      \
        [&](pcu_hip_ctx* l, PendingPair<T>& pp, int p, unsigned lf, pcu_hip_stats* ls) { return hausdorff_begin<T>(l, xs[p], nxs[p], ys[p], nys[p], true, max_leaf, lf, nullptr, ls, pp); }, \
        [&](pcu_hip_ctx* l, PendingPair<T>& pp, int p) { return hausdorff_end<T>(l, pp, out_d2 + 2 * (size_t)p, out_i2 + 2 * (size_t)p, out_j2 + 2 * (size_t)p); }));       \
}
PCU_BATCH_HAUSDORFF(f32, float)
PCU_BATCH_HAUSDORFF(f64, double)
#undef PCU_BATCH_HAUSDORFF
#define PCU_BATCH_CHAMFER(SUF, T)                                                                                                           \
int pcu_hip_chamfer_batch_##SUF(pcu_hip_ctx* c, int n_pairs, const T* const* xs, const int64_t* nxs, const T* const* ys, const int64_t* nys,     \
                                double p_norm, int max_leaf, double* out_mean2, unsigned flags, void* stream, pcu_hip_stats* st) {          \
    CallGuard dg(c);                                                                                                     \
    return abi_rc(batch_run<T>(c, n_pairs, flags, stream, st,                                                                                      \
        [&](pcu_hip_ctx* l, PendingPair<T>& pp, int p, unsigned lf, pcu_hip_stats* ls) { return chamfer_begin<T>(l, xs[p], nxs[p], ys[p], nys[p], p_norm, max_leaf, nullptr, nullptr, lf, nullptr, ls, pp); }, \
        [&](pcu_hip_ctx* l, PendingPair<T>& pp, int p) { return chamfer_end<T>(l, pp, out_mean2 + 2 * (size_t)p); }));                            \
}
PCU_BATCH_CHAMFER(f32, float)
PCU_BATCH_CHAMFER(f64, double)
#undef PCU_BATCH_CHAMFER
int pcu_hip_normals_knn_f32(pcu_hip_ctx* c, const float* p, int64_t n, const float* dirs, int k, int max_leaf, double drop, float* out_n, uint8_t* keep,
                            unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(normals_knn_impl<float>(c, p, n, dirs, k, max_leaf, drop, out_n, keep, flags, stream, st)); }
int pcu_hip_normals_knn_f64(pcu_hip_ctx* c, const double* p, int64_t n, const double* dirs, int k, int max_leaf, double drop, double* out_n, uint8_t* keep,
                            unsigned flags, void* stream, pcu_hip_stats* st) { CallGuard dg(c); return abi_rc(normals_knn_impl<double>(c, p, n, dirs, k, max_leaf, drop, out_n, keep, flags, stream, st)); }
int pcu_hip_normals_ball_f32(pcu_hip_ctx* c, const float* p, int64_t n, const float* dirs, double radius, int min_pts, int max_pts, int weight_rbf, double drop,
                             float* out_n, uint8_t* keep, unsigned flags, void* stream, pcu_hip_stats* st) {
    CallGuard dg(c); return abi_rc(normals_ball_impl<float>(c, p, n, dirs, radius, min_pts, max_pts, weight_rbf, drop, out_n, keep, flags, stream, st)); }
int pcu_hip_normals_ball_f64(pcu_hip_ctx* c, const double* p, int64_t n, const double* dirs, double radius, int min_pts, int max_pts, int weight_rbf, double drop,
                             double* out_n, uint8_t* keep, unsigned flags, void* stream, pcu_hip_stats* st) {
    CallGuard dg(c); return abi_rc(normals_ball_impl<double>(c, p, n, dirs, radius, min_pts, max_pts, weight_rbf, drop, out_n, keep, flags, stream, st)); }

int pcu_hip_morton_encode(pcu_hip_ctx* c, const int32_t* pts, int64_t n, uint64_t* codes, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(morton_map_impl<int32_t, int32_t>(c, 0, pts, nullptr, n, codes, flags, stream)); }
int pcu_hip_morton_decode(pcu_hip_ctx* c, const uint64_t* codes, int64_t n, int32_t* pts, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(morton_map_impl<uint64_t, uint64_t>(c, 1, codes, nullptr, n, pts, flags, stream)); }
int pcu_hip_morton_addsub(pcu_hip_ctx* c, const uint64_t* c1, const uint64_t* c2, int64_t n, int subtract, uint64_t* out, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(morton_map_impl<uint64_t, uint64_t>(c, subtract ? 3 : 2, c1, c2, n, out, flags, stream)); }
int pcu_hip_morton_knn(pcu_hip_ctx* c, const uint64_t* codes, int64_t n, const uint64_t* qcodes, int64_t m, int k, int sort_dist, int64_t* out_nn, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(morton_knn_impl<uint64_t>(c, codes, n, qcodes, m, k, sort_dist, out_nn, flags, stream)); }
#define PCU_VOXEL(SUF, T, A)                                                                                                                        \
int pcu_hip_voxel_downsample_##SUF(pcu_hip_ctx* c, const T* pts, int64_t n, const A* attrib, int64_t attrib_rows, int attrib_cols, const double* voxel_size3,  \
                                   const double* min_bound3, const double* max_bound3, int min_points_per_voxel, T* out_v, A* out_attrib, int64_t* out_count,  \
                                   unsigned flags, void* stream) {                                                                                  \
    CallGuard dg(c);                                                                                                             \
    return abi_rc(voxel_downsample_impl<T, A>(c, pts, n, attrib, attrib_rows, attrib_cols, voxel_size3, min_bound3, max_bound3, min_points_per_voxel, out_v, out_attrib, out_count, flags, stream)); }
PCU_VOXEL(f32_f32, float, float) PCU_VOXEL(f32_f64, float, double) PCU_VOXEL(f64_f32, double, float) PCU_VOXEL(f64_f64, double, double)
#undef PCU_VOXEL
int pcu_hip_dedup_f32(pcu_hip_ctx* c, const float* pts, int64_t n, double epsilon, float* out_pts, int32_t* out_svi, int32_t* out_svj, int64_t* out_count, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(dedup_impl<float>(c, pts, n, epsilon, out_pts, out_svi, out_svj, out_count, flags, stream)); }
int pcu_hip_dedup_f64(pcu_hip_ctx* c, const double* pts, int64_t n, double epsilon, double* out_pts, int32_t* out_svi, int32_t* out_svj, int64_t* out_count, unsigned flags, void* stream) {
    CallGuard dg(c); return abi_rc(dedup_impl<double>(c, pts, n, epsilon, out_pts, out_svi, out_svj, out_count, flags, stream)); }

#define PCU_SINK(SUF, T)                                                                                                                             \
int pcu_hip_pairwise_##SUF(pcu_hip_ctx* c, const T* a, const T* b, int64_t nb, int64_t m, int64_t n, int64_t d, double p_norm, T* out, unsigned flags, void* stream) {  \
    CallGuard dg(c); return abi_rc(pairwise_impl<T>(c, a, b, nb, m, n, d, p_norm, out, flags, stream)); }                                 \
int pcu_hip_sinkhorn_##SUF(pcu_hip_ctx* c, const T* a, const T* b, const T* M, int64_t nb, int64_t m, int64_t n, double eps, int max_iters, double stop_thresh,  \
                           T* out_P, int* out_iters, unsigned flags, void* stream) {                                                                 \
    CallGuard dg(c); return abi_rc(sinkhorn_impl<T>(c, a, b, M, nb, m, n, eps, max_iters, stop_thresh, out_P, out_iters, flags, stream)); } \
int pcu_hip_dot_##SUF(pcu_hip_ctx* c, const T* x, const T* y, int64_t count, double* out, unsigned flags, void* stream) {                            \
    CallGuard dg(c); return abi_rc(dot_impl<T>(c, x, y, count, out, flags, stream)); }
PCU_SINK(f32, float) PCU_SINK(f64, double)
#undef PCU_SINK

int pcu_hip_ctx_set_batch_lanes(pcu_hip_ctx* c, int lanes) { if (!c) return fail(PCU_HIP_ERR_INVALID, "null context"); c->n_lanes_wanted = lanes > 0 ? lanes : 4; return 0; }

int pcu_hip_debug_kd_tree_f32(pcu_hip_ctx* c, const float* pts, int64_t n, int leaf_max, int64_t* out_vacc, int64_t* out_nnodes) { CallGuard dg(c); return abi_rc(debug_kd<float>(c, pts, n, leaf_max, out_vacc, out_nnodes)); }
int pcu_hip_debug_kd_tree_f64(pcu_hip_ctx* c, const double* pts, int64_t n, int leaf_max, int64_t* out_vacc, int64_t* out_nnodes) { CallGuard dg(c); return abi_rc(debug_kd<double>(c, pts, n, leaf_max, out_vacc, out_nnodes)); }

int pcu_hip_index_create_f32(pcu_hip_ctx* c, const float* r, int64_t nr, int k_hint, unsigned flags, void* stream, pcu_hip_index** out) {
    CallGuard dg(c);
    return abi_rc(index_create_impl<float>(c, r, nr, k_hint, flags, stream, out));
}
int pcu_hip_index_create_f64(pcu_hip_ctx* c, const double* r, int64_t nr, int k_hint, unsigned flags, void* stream, pcu_hip_index** out) {
    CallGuard dg(c);
    return abi_rc(index_create_impl<double>(c, r, nr, k_hint, flags, stream, out));
}
int pcu_hip_index_knn_f32(pcu_hip_ctx* c, const pcu_hip_index* ix, const float* q, int64_t nq, int k, int max_leaf, float* od, int64_t* oi,
                          unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!ix) return fail(PCU_HIP_ERR_INVALID, "null index");
    CallGuard dg(c);
    return abi_rc(knn_impl<float>(c, q, nq, nullptr, ix->n, k, max_leaf, od, oi, flags, stream, st, ix));
}
int pcu_hip_index_knn_f64(pcu_hip_ctx* c, const pcu_hip_index* ix, const double* q, int64_t nq, int k, int max_leaf, double* od, int64_t* oi,
                          unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!ix) return fail(PCU_HIP_ERR_INVALID, "null index");
    CallGuard dg(c);
    return abi_rc(knn_impl<double>(c, q, nq, nullptr, ix->n, k, max_leaf, od, oi, flags, stream, st, ix));
}
int64_t pcu_hip_index_size(const pcu_hip_index* ix) { return ix ? ix->n : 0; }
void pcu_hip_index_destroy(pcu_hip_index* ix) {
    if (!ix) return;
    DeviceGuard dg(ix->device);
    (void)hipDeviceSynchronize();
    index_free(ix);
}

}  // extern "C"
