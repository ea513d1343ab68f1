// csrc/pcu_hip.hip -- C ABI (include/pcu_hip.h) + host orchestration of the gfx950 kernels.
//
// One translation unit, built with:  hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -fPIC -shared
// (-ffp-contract=off is part of the numerical contract: see search.h). No PyTorch, no Python, no fallback:
// every entry point either runs the HIP path or returns an error.
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>
#include <algorithm>
#include <string>
#include <vector>

#include "../../include/pcu_hip.h"
#include "grid.h"
#include "reduce.h"
#include "search.h"

using namespace pcu;

// ------------------------------------------------------------------------------------------------ errors
static thread_local std::string g_err;
static int fail(int code, const char* fmt, ...) {
    char buf[1024];
    va_list ap; va_start(ap, fmt); vsnprintf(buf, sizeof buf, fmt, ap); va_end(ap);
    g_err = buf;
    return code;
}
#define HIP_TRY(x)                                                                              \
    do { hipError_t e_ = (x); if (e_ != hipSuccess)                                             \
        return fail(PCU_HIP_ERR_RUNTIME, "%s failed: %s (%s:%d)", #x, hipGetErrorString(e_), __FILE__, __LINE__); } while (0)

// ------------------------------------------------------------------------------------------------ context
struct pcu_hip_ctx {
    int device = 0;
    hipStream_t own_stream = nullptr;
    char* arena = nullptr; size_t arena_cap = 0, arena_off = 0;
    std::vector<void*> extra;                 // overflow allocations of the current call
    size_t extra_bytes = 0;
    hipEvent_t ev[8] = {};
    hipEvent_t kev[8] = {};                    // brackets of the main (pass-0) search launches of a call
    int n_kev = 0;
    double occupancy = 0;                      // <=0: default
    int* h_pinned = nullptr;                   // small pinned readback buffer
};

static size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

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

static void ctx_end(pcu_hip_ctx* c);
static int ctx_begin(pcu_hip_ctx* c, size_t want_bytes) {
    HIP_TRY(hipSetDevice(c->device));
    ctx_end(c);                                 // drop overflow blocks left by a call that failed midway
    if (want_bytes > c->arena_cap) {
        if (c->arena) { HIP_TRY(hipDeviceSynchronize()); HIP_TRY(hipFree(c->arena)); c->arena = nullptr; c->arena_cap = 0; }
        size_t cap = align_up(want_bytes + (want_bytes >> 3), 1 << 20);
        HIP_TRY(hipMalloc((void**)&c->arena, cap));
        c->arena_cap = cap;
    }
    c->arena_off = 0;
    c->n_kev = 0;
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
    int n = 0, max_cells = 0, scan_blocks = 0;
};

static int max_cells_for(int64_t n, double occ) {
    double c = (double)n / (occ > 0 ? occ : 1.0) * 1.25 + 64.0;
    if (c > 64.0 * 1024 * 1024) c = 64.0 * 1024 * 1024;
    return (int)c;
}
template <typename T>
static size_t index_bytes(int64_t n, double occ) {
    int mc = max_cells_for(n, occ);
    return align_up(sizeof(GridParams<T>), 256) + align_up((size_t)(mc + 1) * 4, 256) + align_up((size_t)n * sizeof(Pt4<T>), 256) +
           2 * align_up((size_t)n * 4, 256) + align_up((size_t)(mc / kScanChunk + 2) * 4, 256);
}
template <typename T>
static int index_alloc(Arena& a, GridIndex<T>& g, int64_t n, double occ) {
    g.n = (int)n; g.max_cells = max_cells_for(n, occ); g.scan_blocks = g.max_cells / kScanChunk + 1;
    if (aalloc(a, &g.gp, 1)) return -1;
    if (aalloc(a, &g.cell_start, (size_t)g.max_cells + 1)) return -1;
    if (aalloc(a, &g.sorted, (size_t)n)) return -1;
    if (aalloc(a, &g.cell_of, (size_t)n)) return -1;
    if (aalloc(a, &g.rank, (size_t)n)) return -1;
    if (aalloc(a, &g.block_sums, (size_t)g.scan_blocks + 1)) return -1;
    return 0;
}
// Enqueue the whole build on `s` (no host synchronisation).
template <typename T>
static int index_build(GridIndex<T>& g, const T* d_pts, double occ, hipStream_t s) {
    const int n = g.n;
    const int nb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL(k_grid_init<T>, dim3(1), dim3(64), 0, s, g.gp);
    hipLaunchKernelGGL(k_bbox<T>, dim3(std::min(nb, 2048)), dim3(kBlock), 0, s, d_pts, n, g.gp);
    hipLaunchKernelGGL(k_make_grid<T>, dim3(1), dim3(64), 0, s, g.gp, n, occ, g.max_cells);
    HIP_TRY(hipMemsetAsync(g.cell_start, 0, ((size_t)g.max_cells + 1) * 4, s));
    hipLaunchKernelGGL(k_count<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.gp, g.cell_of, g.rank, g.cell_start);
    hipLaunchKernelGGL(k_scan_reduce<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums);
    hipLaunchKernelGGL(k_scan_spine, dim3(1), dim3(kBlock), 0, s, g.block_sums, g.scan_blocks);
    hipLaunchKernelGGL(k_scan_apply<T>, dim3(g.scan_blocks), dim3(kBlock), 0, s, g.cell_start, g.gp, g.block_sums, (unsigned)n);
    hipLaunchKernelGGL(k_scatter<T>, dim3(nb), dim3(kBlock), 0, s, d_pts, n, g.cell_of, g.rank, g.cell_start, g.sorted);
    HIP_TRY(hipGetLastError());
    return 0;
}

// ------------------------------------------------------------------------------------------------ search driver
static double default_occupancy(int k) {
    // points per cell so that the k-th neighbour lies within one cell edge of the query with high probability
    // for locally uniform data: the ball of radius 1.12 h holds ~5.9*occ points.
    if (k <= 1) return 1.5;
    return std::max(2.0, (k + 3.0 * sqrt((double)k)) / 5.9);
}

template <typename T, int MODE>
static int launch_search(int K, const SearchArgs<T>& a, int nwork_upper, hipStream_t s) {
    if (nwork_upper <= 0) return 0;
    dim3 grid((nwork_upper + kBlock - 1) / kBlock), block(kBlock);
#define PCU_CASE(KK) case KK: hipLaunchKernelGGL((k_search<T, KK, MODE>), grid, block, 0, s, a); break;
    switch (K) {
        PCU_CASE(1) PCU_CASE(2) PCU_CASE(4) PCU_CASE(8) PCU_CASE(16) PCU_CASE(32) PCU_CASE(64) PCU_CASE(128)
        default: return fail(PCU_HIP_ERR_INVALID, "internal: unsupported K=%d", K);
    }
#undef PCU_CASE
    HIP_TRY(hipGetLastError());
    return 0;
}
static int pow2_at_least(int k) { int p = 1; while (p < k) p <<= 1; return p; }

constexpr int kMaxK = 64;       // FAST slots; LEX uses up to 128 (k+1 rounded up)

template <typename T>
struct SearchScratch {
    int* list_a = nullptr; int* list_b = nullptr; int* ties = nullptr; int* true_ties = nullptr;
    int* counters = nullptr;     // [0]=unresolved [1]=ties [2]=true ties [3]=spare
};
template <typename T>
static size_t scratch_bytes(int64_t nq) { return 4 * align_up((size_t)nq * 4, 256) + 256; }
template <typename T>
static int scratch_alloc(Arena& a, SearchScratch<T>& sc, int64_t nq) {
    if (aalloc(a, &sc.list_a, (size_t)nq)) return -1;
    if (aalloc(a, &sc.list_b, (size_t)nq)) return -1;
    if (aalloc(a, &sc.ties, (size_t)nq)) return -1;
    if (aalloc(a, &sc.true_ties, (size_t)nq)) return -1;
    if (aalloc(a, &sc.counters, 16)) return -1;
    return 0;
}

// All-queries exact KNN of `qidx`'s cloud against `ridx`'s cloud. d_ref_pts is needed only if a coarser
// dataset grid has to be built for far-away queries. Results land in d_out_d / d_out_i in original query order.
template <typename T>
static int knn_device(pcu_hip_ctx* c, Arena& ar, hipStream_t s, const GridIndex<T>& qidx, const GridIndex<T>& ridx0,
                      const T* d_ref_pts, double occ0, int k, bool squared, T* d_out_d, long long* d_out_i,
                      SearchScratch<T>& sc, pcu_hip_stats* st) {
    const int nq = qidx.n;
    const int KF = pow2_at_least(k), KL = pow2_at_least(k + 1);
    GridIndex<T> ridx = ridx0;
    double occ = occ0;
    int R = 1;
    int* cur_list = nullptr; int cur_count = nq;
    int* next_list = sc.list_a;
    GridParams<T> h_gp;
    bool have_gp = false;
    for (int pass = 0; pass < 64; ++pass) {
        HIP_TRY(hipMemsetAsync(sc.counters, 0, 16 * sizeof(int), s));
        SearchArgs<T> a;
        a.gp = ridx.gp; a.ref = ridx.sorted; a.cell_start = ridx.cell_start; a.qsorted = qidx.sorted;
        a.qlist = cur_list; a.qcount_dev = nullptr; a.nq = cur_count; a.R = R; a.kreq = k; a.squared = squared ? 1 : 0;
        a.out_d = d_out_d; a.out_i = d_out_i;
        a.unresolved = next_list; a.n_unresolved = sc.counters + 0; a.ties = sc.ties; a.n_ties = sc.counters + 1;
        const bool time_it = st && pass == 0 && c->n_kev + 2 <= 8;
        if (time_it) (void)hipEventRecord(c->kev[c->n_kev], s);
        if (launch_search<T, MODE_FAST>(KF, a, cur_count, s)) return -1;
        if (time_it) { (void)hipEventRecord(c->kev[c->n_kev + 1], s); c->n_kev += 2; }
        if (!have_gp) { HIP_TRY(hipMemcpyAsync(&h_gp, ridx.gp, sizeof h_gp, hipMemcpyDeviceToHost, s)); }
        HIP_TRY(hipMemcpyAsync(c->h_pinned, sc.counters, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        have_gp = true;
        const int n_unres = c->h_pinned[0], n_ties = c->h_pinned[1];
        if (st) { st->n_passes++; st->n_tie_flagged += n_ties; if (pass == 0) st->n_escalated += n_unres; }
        if (n_ties > 0) {
            // same grid, same radius, total order (d2,row); certification is identical so nothing is lost
            SearchArgs<T> b = a;
            b.qlist = sc.ties; b.nq = n_ties;
            b.unresolved = sc.true_ties /*unused sink*/; b.n_unresolved = sc.counters + 3;
            b.ties = sc.true_ties; b.n_ties = sc.counters + 2;
            if (launch_search<T, MODE_LEX>(KL, b, n_ties, s)) return -1;
            HIP_TRY(hipMemcpyAsync(c->h_pinned, sc.counters, 4 * sizeof(int), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipStreamSynchronize(s));
            if (st) { st->n_passes++; st->n_tie_true += c->h_pinned[2]; }
        }
        if (n_unres == 0) return 0;
        // escalate: wider radius on the same grid, then a coarser grid
        cur_list = next_list; cur_count = n_unres;
        next_list = (cur_list == sc.list_a) ? sc.list_b : sc.list_a;
        const int gmax = std::max(h_gp.G[0], std::max(h_gp.G[1], h_gp.G[2]));
        if (R >= gmax) return fail(PCU_HIP_ERR_RUNTIME, "internal: search did not certify with the whole grid scanned");
        if (R < 4 || gmax <= 8) {
            R *= 2;
        } else {
            occ *= 512.0;                                   // cell edge x8
            GridIndex<T> coarse;
            if (index_alloc(ar, coarse, ridx.n, occ)) return -1;
            if (index_build(coarse, d_ref_pts, occ, s)) return -1;
            if (st) st->n_grid_builds++;
            ridx = coarse; R = 1; have_gp = false;
        }
    }
    return fail(PCU_HIP_ERR_RUNTIME, "internal: too many search passes");
}

// ------------------------------------------------------------------------------------------------ validation
static int validate_sizes(int64_t nq, int64_t nr, const char* qname, const char* rname) {
    if (nq <= 0 || nr <= 0)
        return fail(PCU_HIP_ERR_INVALID,
                    "Invalid input set with zero elements: %s and %s must have shape (n, 3) and (m, 3). "
                    "Got %s.shape = (%lld, 3), %s.shape = (%lld, 3).", qname, rname, qname, (long long)nq, rname, (long long)nr);
    if (nq > 0x7ffffff0ll || nr > 0x7ffffff0ll)
        return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^31-16 rows are not supported");
    return 0;
}

struct Timer {
    pcu_hip_ctx* c; hipStream_t s; pcu_hip_stats* st;
    void mark(int i) { if (st) (void)hipEventRecord(c->ev[i], s); }
    float span(int i, int j) { float ms = 0; if (st) (void)hipEventElapsedTime(&ms, c->ev[i], c->ev[j]); return ms; }
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
template <typename T>
static int knn_impl(pcu_hip_ctx* c, const T* query, int64_t nq, const T* dataset, int64_t nr, int k,
                    T* out_d, int64_t* out_i, unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (k <= 0) return fail(PCU_HIP_ERR_INVALID, "Invalid value for k (%d) must be greater than 0.", k);
    if (validate_sizes(nq, nr, "query_points", "dataset_points")) return PCU_HIP_ERR_INVALID;
    if (k > kMaxK) return fail(PCU_HIP_ERR_INVALID, "k = %d > %d is not supported by the gfx950 path yet", k, kMaxK);
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE, squared = flags & PCU_HIP_SQUARED;
    hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    const double occ = c->occupancy > 0 ? c->occupancy : default_occupancy(k);
    const double occ_q = 2.0;
    size_t need = index_bytes<T>(nr, occ) + index_bytes<T>(nq, occ_q) + scratch_bytes<T>(nq) + 4096;
    if (!on_dev) need += align_up((size_t)nq * 3 * sizeof(T), 256) + align_up((size_t)nr * 3 * sizeof(T), 256) +
                         align_up((size_t)nq * k * sizeof(T), 256) + align_up((size_t)nq * k * 8, 256);
    if (ctx_begin(c, need)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    Timer tm{c, s, st};
    int rc = 0;
    do {
        const T *dq, *dr;
        if ((rc = stage_in(ar, query, nq, on_dev, s, &dq))) break;
        if ((rc = stage_in(ar, dataset, nr, on_dev, s, &dr))) break;
        T* dd = out_d; long long* di = (long long*)out_i;
        if (!on_dev) { if ((rc = aalloc(ar, &dd, (size_t)nq * k))) break; if ((rc = aalloc(ar, &di, (size_t)nq * k))) break; }
        GridIndex<T> ri, qi; SearchScratch<T> sc;
        if ((rc = index_alloc(ar, ri, nr, occ))) break;
        if ((rc = index_alloc(ar, qi, nq, occ_q))) break;
        if ((rc = scratch_alloc(ar, sc, nq))) break;
        tm.mark(0);
        if ((rc = index_build(ri, dr, occ, s))) break;
        if ((rc = index_build(qi, dq, occ_q, s))) break;
        if (st) st->n_grid_builds += 2;
        tm.mark(1);
        if ((rc = knn_device(c, ar, s, qi, ri, dr, occ, k, squared, dd, di, sc, st))) break;
        tm.mark(2);
        if (!on_dev) {
            HIP_TRY(hipMemcpyAsync(out_d, dd, (size_t)nq * k * sizeof(T), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_i, di, (size_t)nq * k * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        if (st) { st->n_queries = nq; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 2); collect_kernel_times(c, st); }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ------------------------------------------------------------------------------------------------ two-sided ops
// Shared front end of hausdorff / chamfer: both clouds indexed once; x->y and y->x searches with k = 1.
template <typename T>
struct PairState {
    const T *dx = nullptr, *dy = nullptr;
    GridIndex<T> ix, iy;
    T *d_xy = nullptr, *d_yx = nullptr;                 // nn distance of each x row in y / each y row in x
    long long *c_xy = nullptr, *c_yx = nullptr;
    SearchScratch<T> sc;
    T* pv = nullptr; long long* pi = nullptr; double* pd = nullptr;   // reduction partials
    T* res_v = nullptr; long long* res_ij = nullptr; double* res_s = nullptr;
};
template <typename T>
static size_t pair_bytes(int64_t nx, int64_t ny, double occ, bool on_dev, bool need_corr_out) {
    size_t b = index_bytes<T>(nx, occ) + index_bytes<T>(ny, occ) + scratch_bytes<T>(std::max(nx, ny)) +
               align_up((size_t)nx * sizeof(T), 256) + align_up((size_t)ny * sizeof(T), 256) +
               align_up((size_t)nx * 8, 256) + align_up((size_t)ny * 8, 256) +
               3 * align_up((size_t)kRedBlocks * 8, 256) + 4096;
    if (!on_dev) b += align_up((size_t)nx * 3 * sizeof(T), 256) + align_up((size_t)ny * 3 * sizeof(T), 256);
    (void)need_corr_out;
    return b;
}
template <typename T>
static int pair_run(pcu_hip_ctx* c, Arena& ar, hipStream_t s, const T* x, int64_t nx, const T* y, int64_t ny, bool on_dev,
                    bool squared, double occ, long long* ext_cxy, long long* ext_cyx, PairState<T>& P, Timer& tm, pcu_hip_stats* st,
                    bool do_xy, bool do_yx) {
    if (stage_in(ar, x, nx, on_dev, s, &P.dx)) return -1;
    if (stage_in(ar, y, ny, on_dev, s, &P.dy)) return -1;
    if (index_alloc(ar, P.ix, nx, occ) || index_alloc(ar, P.iy, ny, occ)) return -1;
    if (scratch_alloc(ar, P.sc, std::max(nx, ny))) return -1;
    if (aalloc(ar, &P.d_xy, (size_t)nx) || aalloc(ar, &P.d_yx, (size_t)ny)) return -1;
    P.c_xy = ext_cxy; P.c_yx = ext_cyx;
    if (!P.c_xy && aalloc(ar, &P.c_xy, (size_t)nx)) return -1;
    if (!P.c_yx && aalloc(ar, &P.c_yx, (size_t)ny)) return -1;
    if (aalloc(ar, &P.pv, (size_t)kRedBlocks) || aalloc(ar, &P.pi, (size_t)kRedBlocks) || aalloc(ar, &P.pd, (size_t)kRedBlocks)) return -1;
    if (aalloc(ar, &P.res_v, 4) || aalloc(ar, &P.res_ij, 8) || aalloc(ar, &P.res_s, 4)) return -1;
    tm.mark(0);
    if (index_build(P.ix, P.dx, occ, s) || index_build(P.iy, P.dy, occ, s)) return -1;
    if (st) st->n_grid_builds += 2;
    tm.mark(1);
    if (do_xy && knn_device(c, ar, s, P.ix, P.iy, P.dy, occ, 1, squared, P.d_xy, P.c_xy, P.sc, st)) return -1;
    if (do_yx && knn_device(c, ar, s, P.iy, P.ix, P.dx, occ, 1, squared, P.d_yx, P.c_yx, P.sc, st)) return -1;
    tm.mark(2);
    return 0;
}

template <typename T>
static int argmax_enqueue(hipStream_t s, const T* d, int n, const long long* corr, PairState<T>& P, int slot) {
    const int nb = std::min((n + kBlock - 1) / kBlock, kRedBlocks);
    hipLaunchKernelGGL(k_argmax_partial<T>, dim3(nb), dim3(kBlock), 0, s, d, n, P.pv, P.pi);
    hipLaunchKernelGGL(k_argmax_final<T>, dim3(1), dim3(kBlock), 0, s, P.pv, P.pi, nb, corr, P.res_v + slot, P.res_ij + 2 * slot);
    HIP_TRY(hipGetLastError());
    return 0;
}

template <typename T>
static int hausdorff_impl(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, bool two_sided,
                          T* out_d, int64_t* out_i, int64_t* out_j, unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (validate_sizes(nx, ny, "source", "targets")) return PCU_HIP_ERR_INVALID;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE, squared = flags & PCU_HIP_SQUARED;
    hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    const double occ = c->occupancy > 0 ? c->occupancy : default_occupancy(1);
    if (ctx_begin(c, pair_bytes<T>(nx, ny, occ, on_dev, false))) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c}; Timer tm{c, s, st};
    int rc = 0;
    do {
        PairState<T> P;
        if ((rc = pair_run(c, ar, s, x, nx, y, ny, on_dev, squared, occ, (long long*)nullptr, (long long*)nullptr, P, tm, st, true, two_sided))) break;
        if ((rc = argmax_enqueue(s, P.d_xy, (int)nx, P.c_xy, P, 0))) break;
        if (two_sided && (rc = argmax_enqueue(s, P.d_yx, (int)ny, P.c_yx, P, 1))) break;
        tm.mark(3);
        T hv[2]; long long hij[4];
        HIP_TRY(hipMemcpyAsync(hv, P.res_v, 2 * sizeof(T), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipMemcpyAsync(hij, P.res_ij, 4 * sizeof(long long), hipMemcpyDeviceToHost, s));
        HIP_TRY(hipStreamSynchronize(s));
        const int nres = two_sided ? 2 : 1;
        for (int r = 0; r < nres; ++r) { out_d[r] = hv[r]; out_i[r] = hij[2 * r]; out_j[r] = hij[2 * r + 1]; }
        if (st) { st->n_queries = two_sided ? nx + ny : nx; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 3); collect_kernel_times(c, st); }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

static int pcode_of(double p) {
    if (p == 2.0) return P_TWO;
    if (p == 1.0) return P_ONE;
    if (isinf(p)) return p > 0 ? P_INF : P_NINF;
    if (p == 0.0) return P_ZERO;
    return P_GEN;
}

template <typename T>
static int chamfer_impl(pcu_hip_ctx* c, const T* x, int64_t nx, const T* y, int64_t ny, double p_norm, double* out_mean2,
                        int64_t* out_cxy, int64_t* out_cyx, unsigned flags, void* stream, pcu_hip_stats* st) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (validate_sizes(nx, ny, "query_points", "dataset_points")) return PCU_HIP_ERR_INVALID;
    if (isnan(p_norm)) return fail(PCU_HIP_ERR_INVALID, "p_norm is NaN");
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = stream ? (hipStream_t)stream : c->own_stream;
    if (st) memset(st, 0, sizeof *st);
    const double occ = c->occupancy > 0 ? c->occupancy : default_occupancy(1);
    if (ctx_begin(c, pair_bytes<T>(nx, ny, occ, on_dev, true))) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c}; Timer tm{c, s, st};
    int rc = 0;
    do {
        PairState<T> P;
        long long* ext_xy = (on_dev && out_cxy) ? (long long*)out_cxy : nullptr;
        long long* ext_yx = (on_dev && out_cyx) ? (long long*)out_cyx : nullptr;
        if ((rc = pair_run(c, ar, s, x, nx, y, ny, on_dev, /*squared=*/false, occ, ext_xy, ext_yx, P, tm, st, true, true))) break;
        const int pc = pcode_of(p_norm);
        // __init__.py:112: norm(x[corrs_y_to_x] - y).mean() -> queries y, targets x ; :113 the other way round
        const int nbx = std::min((int)((nx + kBlock - 1) / kBlock), kRedBlocks), nby = std::min((int)((ny + kBlock - 1) / kBlock), kRedBlocks);
        hipLaunchKernelGGL(k_pnorm_partial<T>, dim3(nbx), dim3(kBlock), 0, s, P.dx, P.dy, P.c_xy, P.d_xy, (int)nx, pc, p_norm, P.pd);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kBlock), 0, s, P.pd, nbx, P.res_s + 0);
        hipLaunchKernelGGL(k_pnorm_partial<T>, dim3(nby), dim3(kBlock), 0, s, P.dy, P.dx, P.c_yx, P.d_yx, (int)ny, pc, p_norm, P.pd);
        hipLaunchKernelGGL(k_sum_final, dim3(1), dim3(kBlock), 0, s, P.pd, nby, P.res_s + 1);
        HIP_TRY(hipGetLastError());
        tm.mark(3);
        double hs[2];
        HIP_TRY(hipMemcpyAsync(hs, P.res_s, 2 * sizeof(double), hipMemcpyDeviceToHost, s));
        if (!on_dev) {
            if (out_cxy) HIP_TRY(hipMemcpyAsync(out_cxy, P.c_xy, (size_t)nx * 8, hipMemcpyDeviceToHost, s));
            if (out_cyx) HIP_TRY(hipMemcpyAsync(out_cyx, P.c_yx, (size_t)ny * 8, hipMemcpyDeviceToHost, s));
        }
        HIP_TRY(hipStreamSynchronize(s));
        out_mean2[0] = hs[0] / (double)nx;
        out_mean2[1] = hs[1] / (double)ny;
        if (st) { st->n_queries = nx + ny; st->ms_index = tm.span(0, 1); st->ms_search = tm.span(1, 2); st->ms_total = tm.span(0, 3); collect_kernel_times(c, st); }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ------------------------------------------------------------------------------------------------ C ABI
extern "C" {

const char* pcu_hip_last_error(void) { return g_err.c_str(); }
const char* pcu_hip_version(void) { return "pcu_hip 0.1 (gfx950)"; }

int pcu_hip_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int pcu_hip_ctx_create(int device, pcu_hip_ctx** out_ctx) {
    if (!out_ctx) return fail(PCU_HIP_ERR_INVALID, "null out_ctx");
    int n = pcu_hip_device_count();
    if (n <= 0) return fail(PCU_HIP_ERR_NO_DEVICE, "no HIP device visible: the gfx950 path has no CPU fallback");
    if (device < 0 || device >= n) return fail(PCU_HIP_ERR_INVALID, "device %d out of range [0,%d)", device, n);
    HIP_TRY(hipSetDevice(device));
    pcu_hip_ctx* c = new pcu_hip_ctx();
    c->device = device;
    HIP_TRY(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    for (auto& e : c->ev) HIP_TRY(hipEventCreate(&e));
    for (auto& e : c->kev) HIP_TRY(hipEventCreate(&e));
    HIP_TRY(hipHostMalloc((void**)&c->h_pinned, 64 * sizeof(int), hipHostMallocDefault));
    *out_ctx = c;
    return 0;
}
void pcu_hip_ctx_destroy(pcu_hip_ctx* c) {
    if (!c) return;
    (void)hipSetDevice(c->device);
    (void)hipDeviceSynchronize();
    ctx_end(c);
    if (c->arena) (void)hipFree(c->arena);
    for (auto& e : c->ev) if (e) (void)hipEventDestroy(e);
    for (auto& e : c->kev) if (e) (void)hipEventDestroy(e);
    if (c->h_pinned) (void)hipHostFree(c->h_pinned);
    if (c->own_stream) (void)hipStreamDestroy(c->own_stream);
    delete c;
}
int pcu_hip_ctx_set_cell_occupancy(pcu_hip_ctx* c, double ppc) { if (!c) return fail(PCU_HIP_ERR_INVALID, "null context"); c->occupancy = ppc; return 0; }
int64_t pcu_hip_ctx_workspace_bytes(pcu_hip_ctx* c) { return c ? (int64_t)c->arena_cap : 0; }

int pcu_hip_knn_f32(pcu_hip_ctx* c, const float* q, int64_t nq, const float* r, int64_t nr, int k, float* od, int64_t* oi,
                    unsigned flags, void* stream, pcu_hip_stats* st) { return knn_impl<float>(c, q, nq, r, nr, k, od, oi, flags, stream, st); }
int pcu_hip_knn_f64(pcu_hip_ctx* c, const double* q, int64_t nq, const double* r, int64_t nr, int k, double* od, int64_t* oi,
                    unsigned flags, void* stream, pcu_hip_stats* st) { return knn_impl<double>(c, q, nq, r, nr, k, od, oi, flags, stream, st); }

int pcu_hip_one_sided_hausdorff_f32(pcu_hip_ctx* c, const float* a, int64_t na, const float* b, int64_t nb, float* od, int64_t* oi, int64_t* oj,
                                    unsigned flags, void* stream, pcu_hip_stats* st) { return hausdorff_impl<float>(c, a, na, b, nb, false, od, oi, oj, flags, stream, st); }
int pcu_hip_one_sided_hausdorff_f64(pcu_hip_ctx* c, const double* a, int64_t na, const double* b, int64_t nb, double* od, int64_t* oi, int64_t* oj,
                                    unsigned flags, void* stream, pcu_hip_stats* st) { return hausdorff_impl<double>(c, a, na, b, nb, false, od, oi, oj, flags, stream, st); }
int pcu_hip_hausdorff_f32(pcu_hip_ctx* c, const float* a, int64_t na, const float* b, int64_t nb, float* od, int64_t* oi, int64_t* oj,
                          unsigned flags, void* stream, pcu_hip_stats* st) { return hausdorff_impl<float>(c, a, na, b, nb, true, od, oi, oj, flags, stream, st); }
int pcu_hip_hausdorff_f64(pcu_hip_ctx* c, const double* a, int64_t na, const double* b, int64_t nb, double* od, int64_t* oi, int64_t* oj,
                          unsigned flags, void* stream, pcu_hip_stats* st) { return hausdorff_impl<double>(c, a, na, b, nb, true, od, oi, oj, flags, stream, st); }

int pcu_hip_chamfer_f32(pcu_hip_ctx* c, const float* x, int64_t nx, const float* y, int64_t ny, double p, double* om, int64_t* cxy, int64_t* cyx,
                        unsigned flags, void* stream, pcu_hip_stats* st) { return chamfer_impl<float>(c, x, nx, y, ny, p, om, cxy, cyx, flags, stream, st); }
int pcu_hip_chamfer_f64(pcu_hip_ctx* c, const double* x, int64_t nx, const double* y, int64_t ny, double p, double* om, int64_t* cxy, int64_t* cyx,
                        unsigned flags, void* stream, pcu_hip_stats* st) { return chamfer_impl<double>(c, x, nx, y, ny, p, om, cxy, cyx, flags, stream, st); }

}  // extern "C"
