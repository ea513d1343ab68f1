// csrc/grid.h -- uniform-grid index build (HBM-bound integer/byte passes; no MFMA, no LDS tiling needed).
//
// Replaces the role of the reference's kd-tree *build* (nanoflann.hpp:1363-1375 buildIndex / :1001-1059
// divideTree, which pcu runs three times per call: src/point_cloud_distance.cpp:41-42) with a counting sort of
// the cloud into snake (boustrophedon) grid-cell order. The grid only decides which candidates a query evaluates; it has
// no influence on results, which depend only on the distance arithmetic in search.h.
//
// Pipeline (all on one stream, no host round trip; grid shape lives in device memory; both clouds of a call share the
// launches of every pass except the bucket sort):
//   k_bbox_partial   256 blocks per cloud   bbox partials (shuffle reduce, no atomics) + zero-fill of the cell counters
//   k_make_grid      1 block per cloud      fold partials; bbox -> cell edge h, cell counts G (`occupancy` points per cell)
//   bucketed build (default, "bucketed build" below): k_bucket_count, k_bucket_scatter, k_bucket_sort, k_bucket_large --
//       a two-level counting sort on the cell id whose per-point atomics are LDS atomics
//   atomic build (tiny clouds, refitted / very coarse grids, PCU_HIP_INDEX=atomic):
//       k_count (cell id + rank-in-cell by one returning device-scope atomicAdd per point), k_scan_reduce/_apply, k_scatter
#pragma once
#include "pcu_types.h"

#ifndef PCU_SORT_THREADS
#define PCU_SORT_THREADS 1024     // workgroup size of k_bucket_sort. A/B on MI355X (threads/bucket/stage -> headline, C5 ms):
                                  // 512/2048/2240 0.188 0.375 | 1024/2048/2240 0.194 0.317 | 1024/4096/4352 0.184 0.340 | 256/1024/1152 0.207
#endif

namespace pcu {

constexpr int kBlock = 256;

template <typename T>
__device__ __forceinline__ T wave_min(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w < v ? w : v; }
    return v;
}
template <typename T>
__device__ __forceinline__ T wave_max(T v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    return v;
}

// Wave reductions whose result is needed in ONE lane (lane 63): six DPP steps in the vector ALU (row_shr 1 / 2 / 4 / 8, row_bcast 15 / 31)
// instead of six ds_bpermute round trips through the LDS crossbar per value (__shfl_xor). Used where a block folds a dozen values
// (bbox, moments, flags): k_bbox_partial, make_grid_body.
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ int dpp_i(int old, int v) { return __builtin_amdgcn_update_dpp(old, v, CTRL, ROW_MASK, BANK_MASK, BOUND); }
template <int CTRL, int ROW_MASK, int BANK_MASK, bool BOUND>
__device__ __forceinline__ float dpp_f(float old, float v) { return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(v), CTRL, ROW_MASK, BANK_MASK, BOUND)); }
#define PCU_DPP_STEPS(OP, ID, DPP)                                  \
    v = OP(v, DPP<0x111, 0xf, 0xf, ID>(ID ? 0 : v, v));             \
    v = OP(v, DPP<0x112, 0xf, 0xf, ID>(ID ? 0 : v, v));             \
    v = OP(v, DPP<0x114, 0xf, 0xe, ID>(ID ? 0 : v, v));             \
    v = OP(v, DPP<0x118, 0xf, 0xc, ID>(ID ? 0 : v, v));             \
    v = OP(v, DPP<0x142, 0xa, 0xf, ID>(ID ? 0 : v, v));             \
    v = OP(v, DPP<0x143, 0xc, 0xf, ID>(ID ? 0 : v, v));
// (sum / or: lanes without a source add the identity 0 -- old = 0, bound_ctrl; min / max: they keep their own value -- old = v)
__device__ __forceinline__ float red_add(float a, float b) { return a + b; }
__device__ __forceinline__ float red_min(float a, float b) { return b < a ? b : a; }
__device__ __forceinline__ float red_max(float a, float b) { return b > a ? b : a; }
__device__ __forceinline__ int red_or(int a, int b) { return a | b; }
__device__ __forceinline__ float wave_sum63(float v) { PCU_DPP_STEPS(red_add, true, dpp_f) return v; }
__device__ __forceinline__ float wave_min63(float v) { PCU_DPP_STEPS(red_min, false, dpp_f) return v; }
__device__ __forceinline__ float wave_max63(float v) { PCU_DPP_STEPS(red_max, false, dpp_f) return v; }
__device__ __forceinline__ unsigned wave_or63(unsigned u) { int v = (int)u; PCU_DPP_STEPS(red_or, true, dpp_i) return (unsigned)v; }
#undef PCU_DPP_STEPS
// double: the shuffle forms (every lane gets the result; the float64 entry points are not the ones a launch chain is tuned for)
__device__ __forceinline__ double wave_sum63(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_min63(double v) { return wave_min(v); }
__device__ __forceinline__ double wave_max63(double v) { return wave_max(v); }

// "Last block finishes the job" without __threadfence(): on gfx950 a device-scope fence is an L2 write-back +
// invalidate (buffer_wbl2 / buffer_inv) per block, which costs more than the launch it saves. Instead the few values
// that cross blocks are written with agent-scope atomic stores (write-through, `sc1`), the writer waits for them
// (s_waitcnt) before taking its ticket with a relaxed atomic, and the last block reads them back with agent-scope
// atomic loads (`sc1`, served coherently).
template <typename V>
__device__ __forceinline__ void publish(V* p, V v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
template <typename V>
__device__ __forceinline__ V peek(const V* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void wait_stores() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
// to be called by ONE thread after a __syncthreads() that follows the publishing threads' wait_stores()
__device__ __forceinline__ bool take_ticket(unsigned* ticket, unsigned nblocks) {
    return __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == nblocks - 1;
}

#ifndef PCU_BBOX_BLOCKS
#define PCU_BBOX_BLOCKS 256
#endif
constexpr int kBboxBlocks = PCU_BBOX_BLOCKS;   // one per CU; partial[b][0..2] = min xyz, [3..5] = max xyz (finite values only), [6] = non-finite flags
constexpr int kBboxStride = 16;                // values per partial: [0..2] min, [3..5] max, [6] non-finite flags, [7] finite points, [8..10] sum (v - pivot), [11..13] sum (v - pivot)^2
// Robust grid range. The grid is laid over [max(min, mean - kCoreSigmas sigma), min(max, mean + kCoreSigmas sigma)] per axis instead of
// the exact bounding box: a few stray points far from the cloud (real scans have them) would otherwise inflate the box until the
// whole cloud sits in a handful of cells -- the first build then died in per-cell atomics (1.2 ms for 1M points) and the call went
// through the refit machinery (2.3 ms against 0.17 ms). Points beyond the range are held by the border cells (cell_coord clamps; the
// searches know: GridParams::org, face_lower_bound), the exact box stays in gmin / gmax for the certification. Clouds that fill
// their box (uniform, surfaces, two far clusters: box within mean +- 1.8 sigma) keep exactly the box. The moments are taken
// relative to the cloud's first point (cancellation: a cloud at 1000 +- 0.001) and are a heuristic only: the grid decides which
// candidates a query looks at, never a result.
constexpr double kCoreSigmas = 3.0;
// Non-finite coordinates (k_bbox_partial -> GridParams::nonfinite). The bounding box -- hence the grid -- is laid over the FINITE values;
// points with a non-finite coordinate sit in border cells (cell_coord clamps) and their d2 is +inf or NaN, which never beats a
// neighbour (strict '<' against a k-th best that starts at FLT_MAX): exactly what the reference's result set does with them
// (nanoflann.hpp:1563 `dist < worst_dist`, :182). What the reference does NOT survive is a kd-tree built over NaN bounds: a NaN in
// the dataset, or +inf and -inf along one axis ((low + high) / 2 = NaN, nanoflann.hpp:1090), make its traversal prune arbitrarily
// and its rows depend on the tree. Those inputs are rejected (kNfNaN / kNfBothInf -> ValueError), see pcu_hip.hip: search_finish.
constexpr int kNfNaN = 1, kNfBothInf = 2, kNfAnyInf = 4;

// pts: row-major (n,3). Lane i reads 3 consecutive scalars at 3*i: a wave covers one contiguous 768 B
// (f32) span with three strided dword loads, all of whose sectors are consumed. No atomics: every block
// writes one partial; k_make_grid folds the kBboxBlocks partials. The same launch zero-fills the cell
// counters (so the build needs no separate memset launch).
template <typename T>
__device__ __forceinline__ void bbox_body(const T* __restrict__ pts, int n, T* partial, unsigned* __restrict__ counts, int n_counts,
                                          unsigned* __restrict__ zero2, int n_zero2, const int bid, const int nblk) {
    T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
    T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
    unsigned nf = 0;          // bit 0: NaN seen; bits 1..3: +inf on axis j; bits 4..6: -inf on axis j
    auto classify = [&](T v, int j) { nf |= v != v ? 1u : (v > (T)0 ? (2u << j) : (16u << j)); };
    T piv[3], s1[3] = {(T)0, (T)0, (T)0}, s2[3] = {(T)0, (T)0, (T)0}, cnt = (T)0;       // moments of the finite points about the cloud's first point
#pragma unroll
    for (int j = 0; j < 3; ++j) { const T v = pts[j]; piv[j] = ((v < (T)0 ? -v : v) <= Limits<T>::max_v) ? v : (T)0; }
    // four points = 12 consecutive scalars = three 16-byte (f32) loads per trip, all in flight together; a plain
    // point-per-trip loop waits for memory 15 times per thread at n = 1M
    struct __attribute__((packed, aligned(4))) Vec4 { T v[4]; };
    const int n4 = n >> 2;
    const int gtid = bid * kBlock + (int)threadIdx.x, gstride = nblk * kBlock;
    for (int gi = gtid; gi < n4; gi += gstride) {
        const Vec4* p = reinterpret_cast<const Vec4*>(pts + 12 * (size_t)gi);
        const Vec4 a = p[0], b = p[1], c = p[2];
        const T v[12] = {a.v[0], a.v[1], a.v[2], a.v[3], b.v[0], b.v[1], b.v[2], b.v[3], c.v[0], c.v[1], c.v[2], c.v[3]};
        bool all_fin = true;
#pragma unroll
        for (int k = 0; k < 12; ++k) {
            const bool fin = (v[k] < (T)0 ? -v[k] : v[k]) <= Limits<T>::max_v;      // false for NaN and +-inf
            all_fin = all_fin && fin;
            lo[k % 3] = (fin && v[k] < lo[k % 3]) ? v[k] : lo[k % 3];
            hi[k % 3] = (fin && v[k] > hi[k % 3]) ? v[k] : hi[k % 3];
            const T dv = fin ? v[k] - piv[k % 3] : (T)0;
            s1[k % 3] += dv; s2[k % 3] += dv * dv;
        }
        cnt += (T)4;
        if (!all_fin) {
#pragma unroll
            for (int k = 0; k < 12; ++k) if (!((v[k] < (T)0 ? -v[k] : v[k]) <= Limits<T>::max_v)) classify(v[k], k % 3);
        }
    }
    if (gtid < (n & 3)) {
        const int i = (n4 << 2) + gtid;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            T v = pts[3 * (size_t)i + j];
            const bool fin = (v < (T)0 ? -v : v) <= Limits<T>::max_v;
            if (!fin) classify(v, j);
            lo[j] = (fin && v < lo[j]) ? v : lo[j];
            hi[j] = (fin && v > hi[j]) ? v : hi[j];
            const T dv = fin ? v - piv[j] : (T)0;
            s1[j] += dv; s2[j] += dv * dv;
        }
        cnt += (T)1;
    }
    {   // zero-fill (16-byte stores; `counts` is 256-byte aligned arena memory)
        uint4* c4 = reinterpret_cast<uint4*>(counts);
        const int m4 = n_counts >> 2;
        for (int i = gtid; i < m4; i += gstride) c4[i] = make_uint4(0u, 0u, 0u, 0u);
        if (gtid < (n_counts & 3)) counts[(m4 << 2) + gtid] = 0u;
        for (int i = gtid; i < n_zero2; i += gstride) zero2[i] = 0u;
    }
    __shared__ T s_lo[kBlock / 64][3], s_hi[kBlock / 64][3], s_mom[kBlock / 64][7];
    __shared__ unsigned s_nf[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T a = wave_min63(lo[j]), b = wave_max63(hi[j]);
        if (lane == 63) { s_lo[wave][j] = a; s_hi[wave][j] = b; }
    }
    {
        T m[7] = {cnt, s1[0], s1[1], s1[2], s2[0], s2[1], s2[2]};
#pragma unroll
        for (int q = 0; q < 7; ++q) { const T r = wave_sum63(m[q]); if (lane == 63) s_mom[wave][q] = r; }
    }
    {
        const unsigned m = wave_or63(nf);
        if (lane == 63) s_nf[wave] = m;
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        T a = s_lo[0][j], b = s_hi[0][j];
        for (int w = 1; w < kBlock / 64; ++w) { a = s_lo[w][j] < a ? s_lo[w][j] : a; b = s_hi[w][j] > b ? s_hi[w][j] : b; }
        publish(&partial[bid * kBboxStride + j], a);
        publish(&partial[bid * kBboxStride + 3 + j], b);
    } else if (threadIdx.x == 3) {
        unsigned m = 0;
        for (int w = 0; w < kBlock / 64; ++w) m |= s_nf[w];
        publish(&partial[bid * kBboxStride + 6], (T)m);       // (<= 127: exact in T)
    } else if (threadIdx.x >= 4 && threadIdx.x < 11) {
        const int q = threadIdx.x - 4;
        T a = s_mom[0][q];
        for (int w = 1; w < kBlock / 64; ++w) a += s_mom[w][q];
        publish(&partial[bid * kBboxStride + 7 + q], a);
    }
}

// Both clouds of a call are indexed by the SAME launches: blocks [0, nb0) work on side 0, the rest on side 1 (a side with
// n = 0 gets no blocks). The build passes are latency-bound (1-3 TB/s), so two clouds per launch cost far less than two
// launches, and the launch count of a two-sided call halves.
template <typename T>
struct BboxSide { const T* pts; int n; T* partial; unsigned* counts; int n_counts; unsigned* zero2; int n_zero2; GridParams<T>* gp; };
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bbox_partial(const BboxSide<T> a0, const BboxSide<T> a1, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const BboxSide<T>& a = second ? a1 : a0;
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    if (bid == 0 && threadIdx.x == 0) { a.gp->sumsq = 0ull; a.gp->has_large = 0; }        // (the accumulators: every later kernel of the build adds to / raises them)
    bbox_body<T>(a.pts, a.n, a.partial, a.counts, a.n_counts, a.zero2, a.n_zero2, bid, second ? (int)gridDim.x - nb0 : nb0);
}

// The range [rlo, rhi] per axis -> a grid: cubic cells of edge h with about `occupancy` points per cell if the cloud filled the range
// uniformly, capped at max_cells; axes with (near-)zero extent get one cell. Serial (one thread). Writes h, inv_h, G, ncells, slack, org, closed.
template <typename T>
__device__ __forceinline__ void grid_layout(GridParams<T>* gp, const T (&rlo)[3], const T (&rhi)[3], int n, double occupancy, int max_cells, double h_want) {
    double ext[3];
    for (int j = 0; j < 3; ++j) ext[j] = (double)rhi[j] - (double)rlo[j];
    double emax = ext[0] > ext[1] ? (ext[0] > ext[2] ? ext[0] : ext[2]) : (ext[1] > ext[2] ? ext[1] : ext[2]);
    double want = (double)n / (occupancy > 0 ? occupancy : 1.0);
    if (want > (double)max_cells) want = (double)max_cells;
    if (want < 1.0) want = 1.0;
    int G[3] = {1, 1, 1};
    double h = 1.0;
    if (emax > 0 && isfinite(emax)) {
        // active axes: extent not negligible against the largest one
        bool act[3]; int nd = 0; double vol = 1.0;
        for (int j = 0; j < 3; ++j) { act[j] = ext[j] > emax * 1e-6; if (act[j]) { ++nd; vol *= ext[j]; } }
        // (single precision for the root: this runs serially on one thread and any h near the target will do -- the grid only decides which
        // candidates a query looks at; every block of a one-pass build computes the same value from the same inputs)
        const float ratio = (float)vol / (float)want;
        h = (double)(nd == 3 ? cbrtf(ratio) : (nd == 2 ? sqrtf(ratio) : ratio));
        if (!(h > 0.0) || !isfinite(h)) h = pow(vol / want, 1.0 / nd);          // (ratio outside the float range)
        if (h_want > 0 && h_want > h) h = h_want;             // fixed-radius searches (normals.h): cells no smaller than asked for
        for (int it = 0; it < 400; ++it) {
            double cells = 1.0;
            const double inv_hd = 1.0 / h;                   // (a cell count one off at an exact multiple is harmless: cell_coord clamps)
            for (int j = 0; j < 3; ++j) {
                double g = act[j] ? floor(ext[j] * inv_hd) + 1.0 : 1.0;
                if (g > 2048.0) g = 2048.0;                    // keeps row tables and int math small
                G[j] = (int)g; cells *= g;
            }
            if (cells <= (double)max_cells) break;
            h *= 1.05;
        }
        // a capped axis (2048) must still span its extent
        for (int j = 0; j < 3; ++j) if (act[j] && h * G[j] < ext[j]) h = ext[j] / G[j] * 1.0000001;
    }
    gp->h = (T)h;
    gp->inv_h = (T)1 / gp->h;
    for (int j = 0; j < 3; ++j) {
        gp->G[j] = G[j];
        double scale = fabs((double)rlo[j]) + fabs((double)rhi[j]) + (double)G[j] * h;
        gp->slack[j] = (T)(8.0 * (double)Limits<T>::eps * scale);
    }
    gp->ncells = G[0] * G[1] * G[2];
    for (int j = 0; j < 3; ++j) gp->org[j] = rlo[j];
    gp->closed = 0;
}

// One block folds the bbox partials; one thread then turns the bbox into a grid: cubic cells of edge h with about `occupancy` points per cell if the
// cloud filled its bbox uniformly, capped at max_cells. Axes with (near-)zero extent get one cell.
// NT = threads of the calling block. `gp` may be a block's private copy in LDS (k_bucket_onepass: every block lays out the grid itself,
// which saves the k_make_grid launch): with accumulators = false the fields other kernels accumulate into or raise -- sumsq, has_large,
// zeroed by k_bbox_partial -- and the sentinel records are left alone.
// (The partials come from an earlier launch: plain loads. With agent-scope loads -- which bypass the L1 -- the 490 blocks of a
// one-pass build hammered the 14 cache lines of the partials in L2 and the layout cost 7.6 us per block instead of ~2.)
template <typename T, int NT>
__device__ __forceinline__ void make_grid_body(GridParams<T>* gp, const T* __restrict__ partial, int nparts, int n, double occupancy, int max_cells, Pt4<T>* sentinel, double h_want = 0.0,
                                               bool accumulators = true, const T* __restrict__ pivot = nullptr) {
    __shared__ T s_lo[NT / 64][3], s_hi[NT / 64][3];
    __shared__ double s_mom[NT / 64][7];
    __shared__ unsigned s_nf[NT / 64];
    // (the serial part below is one thread's dependent chain on the critical path of the build: its three global loads are requested
    // here, and it avoids double-precision divisions and roots where a float or a reciprocal will do)
    T piv[3] = {(T)0, (T)0, (T)0};
    if (pivot) { piv[0] = pivot[0]; piv[1] = pivot[1]; piv[2] = pivot[2]; }
    {
        T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
        T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
        double mom[7] = {0, 0, 0, 0, 0, 0, 0};
        unsigned nf = 0;
        for (int b = threadIdx.x; b < nparts; b += NT) {
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T a = partial[b * kBboxStride + j], c = partial[b * kBboxStride + 3 + j];
                lo[j] = a < lo[j] ? a : lo[j]; hi[j] = c > hi[j] ? c : hi[j];
            }
            nf |= (unsigned)partial[b * kBboxStride + 6];
#pragma unroll
            for (int q = 0; q < 7; ++q) mom[q] += (double)partial[b * kBboxStride + 7 + q];
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (wave < (nparts + 63) / 64) {            // (the other waves of a large block hold no partials: their slots are not read)
#pragma unroll
            for (int j = 0; j < 3; ++j) {
                T a = wave_min63(lo[j]), c = wave_max63(hi[j]);
                if (lane == 63) { s_lo[wave][j] = a; s_hi[wave][j] = c; }
            }
            // (the moments are a heuristic -- see kCoreSigmas --: a wave's 64 block sums are added in T, the waves' sums in double)
#pragma unroll
            for (int q = 0; q < 7; ++q) { const T r = wave_sum63((T)mom[q]); if (lane == 63) s_mom[wave][q] = (double)r; }
            const unsigned m = wave_or63(nf);
            if (lane == 63) s_nf[wave] = m;
        }
    }
    __syncthreads();
    if (threadIdx.x != 0) return;
    const int nw = min((nparts + 63) / 64, NT / 64);      // waves that hold partials
    {
        unsigned m = 0;
        for (int w = 0; w < nw; ++w) m |= s_nf[w];
        const unsigned pinf = (m >> 1) & 7u, ninf = (m >> 4) & 7u;
        gp->nonfinite = ((m & 1u) ? kNfNaN : 0) | ((pinf & ninf) ? kNfBothInf : 0) | ((pinf | ninf) ? kNfAnyInf : 0) | (int)(m << 8);      // (bits 8..14: the raw mask, for the kd-tree's root box)
    }
    if (sentinel) put_sentinels(sentinel - n, n);
    double mom[7] = {0, 0, 0, 0, 0, 0, 0};
    T rlo[3], rhi[3];                               // the range the grid is laid over (see kCoreSigmas)
    for (int w = 0; w < nw; ++w) for (int q = 0; q < 7; ++q) mom[q] += s_mom[w][q];
    const double inv0 = mom[0] > 0 ? 1.0 / mom[0] : 0.0;
    for (int j = 0; j < 3; ++j) {
        T lo = s_lo[0][j], hi = s_hi[0][j];
        for (int w = 1; w < nw; ++w) { lo = s_lo[w][j] < lo ? s_lo[w][j] : lo; hi = s_hi[w][j] > hi ? s_hi[w][j] : hi; }
        if (!(lo <= hi)) { lo = 0; hi = 0; }      // no finite value in this column
        gp->gmin[j] = lo; gp->gmax[j] = hi;
        rlo[j] = lo; rhi[j] = hi;
        if (pivot && mom[0] > 0) {
            T pv = piv[j]; if (!((pv < (T)0 ? -pv : pv) <= Limits<T>::max_v)) pv = (T)0;
            const double m1 = mom[1 + j] * inv0, var = mom[4 + j] * inv0 - m1 * m1, sd = var > 0 ? (double)sqrtf((float)var) : 0.0, mu = (double)pv + m1;
            const T a = (T)(mu - kCoreSigmas * sd), b = (T)(mu + kCoreSigmas * sd);
            if (a <= b) {                         // (moments that overflowed give NaN: the exact box stands)
                if (a > rlo[j] && a < hi) rlo[j] = a;
                if (b < rhi[j] && b > rlo[j]) rhi[j] = b;
            }
        }
    }
    grid_layout<T>(gp, rlo, rhi, n, occupancy, max_cells, h_want);
    if (accumulators) { gp->sumsq = 0ull; gp->has_large = 0; }
}
template <typename T>
struct GridSide { GridParams<T>* gp; const T* partial; int nparts; int n; double occupancy; int max_cells; Pt4<T>* sentinel; double h_want; const T* pts; };
template <typename T>
__global__ __launch_bounds__(kBlock) void k_make_grid(const GridSide<T> a0, const GridSide<T> a1) {      // one block per side
    const GridSide<T>& a = blockIdx.x ? a1 : a0;
    make_grid_body<T, kBlock>(a.gp, a.partial, a.nparts, a.n, a.occupancy, a.max_cells, a.sentinel, a.h_want, true, a.pts);
}

// Cell id + rank-in-cell of every point. The ranks come from a returning atomicAdd on the cell's counter, and those
// are memory-side on MI355X: thousands of points of a dense cluster in one cell serialise on one address (a refit build
// over a tight cluster took ~180 us, a coarse build over a Gaussian blob ~1 ms). Each block therefore first aggregates its
// 256 points by cell in a small LDS hash table (LDS atomics), issues ONE returning global atomic per distinct cell, and
// hands out the ranks locally. On spread-out data (every point its own cell) this is the same number of global atomics.
constexpr int kCountSlots = 512;           // >= 2 x kBlock: open addressing always finds a slot
template <typename T>
__global__ __launch_bounds__(kBlock) void k_count(const T* __restrict__ pts, int n, const GridParams<T>* __restrict__ gp,
                                                  unsigned* __restrict__ cell_of, unsigned* __restrict__ rank,
                                                  unsigned* counts) {
    __shared__ unsigned s_key[kCountSlots], s_cnt[kCountSlots], s_base[kCountSlots];
    for (int i = threadIdx.x; i < kCountSlots; i += kBlock) { s_key[i] = 0xffffffffu; s_cnt[i] = 0u; }
    __syncthreads();
    const int i = blockIdx.x * kBlock + threadIdx.x;
    unsigned c = 0xffffffffu, slot = 0, lr = 0;
    if (i < n) {
        const T x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
        bool in = true;
        if (gp->closed) {      // sub-box level: points outside the box are simply not part of this index
            const T tx = (x - gp->org[0]) * gp->inv_h, ty = (y - gp->org[1]) * gp->inv_h, tz = (z - gp->org[2]) * gp->inv_h;
            in = tx >= 0 && tx < (T)gp->G[0] && ty >= 0 && ty < (T)gp->G[1] && tz >= 0 && tz < (T)gp->G[2];
        }
        if (in) {
            const int cx = grid_cell(*gp, 0, x), cy = grid_cell(*gp, 1, y), cz = grid_cell(*gp, 2, z);
            c = (unsigned)row_run_lo(gp->G[0], grid_row(gp->G[1], cy, cz), cx, cx);
            slot = (c * 2654435761u) >> 23;                      // 9 bits
            for (;;) {
                const unsigned prev = atomicCAS(&s_key[slot], 0xffffffffu, c);
                if (prev == 0xffffffffu || prev == c) break;
                slot = (slot + 1) & (kCountSlots - 1);
            }
            lr = atomicAdd(&s_cnt[slot], 1u);
        }
        cell_of[i] = c;
    }
    __syncthreads();
    for (int k = threadIdx.x; k < kCountSlots; k += kBlock)
        if (s_key[k] != 0xffffffffu) s_base[k] = atomicAdd(&counts[s_key[k]], s_cnt[k]);
    __syncthreads();
    if (i < n) rank[i] = c != 0xffffffffu ? s_base[slot] + lr : 0u;
}

// ---- exclusive scan over `counts[0..m)` in place; counts[m] receives the total -------------------------
constexpr int kScanItems = 8;                       // per thread
constexpr int kScanChunk = kBlock * kScanItems;     // per block

__device__ __forceinline__ unsigned block_exclusive_scan(unsigned v, unsigned* total) {
    __shared__ unsigned s_w[kBlock / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned run = 0;
        for (int w = 0; w < kBlock / 64; ++w) { unsigned t = s_w[w]; s_w[w] = run; run += t; }
        s_w[kBlock / 64] = run;
    }
    __syncthreads();
    unsigned ex = inc - v + s_w[wave];
    *total = s_w[kBlock / 64];
    __syncthreads();
    return ex;
}

// Same for a block of NT threads (NT a multiple of 64).
template <int NT>
__device__ __forceinline__ unsigned block_exclusive_scan_nt(unsigned v, unsigned* total) {
    __shared__ unsigned s_w[NT / 64 + 1];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = v;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    if (threadIdx.x < 64) {
        unsigned t = threadIdx.x < NT / 64 ? s_w[threadIdx.x] : 0u, ti = t;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { unsigned u = __shfl_up(ti, o, 64); if (lane >= o) ti += u; }
        if (threadIdx.x < NT / 64) s_w[threadIdx.x] = ti - t;
        if (threadIdx.x == NT / 64 - 1) s_w[NT / 64] = ti;
    }
    __syncthreads();
    unsigned ex = inc - v + s_w[wave];
    *total = s_w[NT / 64];
    __syncthreads();
    return ex;
}

// m is read from device memory (gp->ncells) so no host round trip is needed between build stages.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan_reduce(const unsigned* __restrict__ counts, GridParams<T>* gp,
                                                        unsigned* __restrict__ block_sums) {
    const int m = gp->ncells;
    const int base = blockIdx.x * kScanChunk;
    if (base >= m) { if (threadIdx.x == 0) block_sums[blockIdx.x] = 0; return; }
    unsigned s = 0; unsigned long long s2 = 0;       // s2: balance metric sum(count^2), see GridParams::sumsq
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) {
        int i = base + j * kBlock + threadIdx.x;
        if (i < m) { const unsigned c = counts[i]; s += c; s2 += (unsigned long long)c * c; }
    }
    unsigned total; block_exclusive_scan(s, &total);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s2 += __shfl_xor(s2, o, 64);
    if ((threadIdx.x & 63) == 0 && s2) atomicAdd(&gp->sumsq, s2);
    if (threadIdx.x == 0) block_sums[blockIdx.x] = total;
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_scan_apply(unsigned* counts, const GridParams<T>* __restrict__ gp,
                                                       const unsigned* __restrict__ block_sums, unsigned n_total) {
    const int m = gp->ncells;
    const int base = blockIdx.x * kScanChunk;
    if (base >= m) return;
    // thread-contiguous items so the running sum within a thread is the scan order
    unsigned v[kScanItems], s = 0;
    const int i0 = base + threadIdx.x * kScanItems;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) { v[j] = (i0 + j < m) ? counts[i0 + j] : 0; s += v[j]; }
    // offset of this chunk = sum of the totals of all earlier chunks (a few hundred values: folded here
    // rather than by a separate single-block "spine" launch)
    unsigned pre = 0;
    for (int b = threadIdx.x; b < (int)blockIdx.x; b += kBlock) pre += block_sums[b];
    unsigned total, ptotal;
    block_exclusive_scan(pre, &ptotal);
    unsigned ex = block_exclusive_scan(s, &total) + ptotal;
#pragma unroll
    for (int j = 0; j < kScanItems; ++j) { if (i0 + j < m) counts[i0 + j] = ex; ex += v[j]; }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        if (n_total == 0xffffffffu) { n_total = 0; for (int b = 0; b * kScanChunk < m; ++b) n_total += block_sums[b]; }   // closed index: count what is in it
        counts[m] = n_total;
    }
}

// `rank` is turned into the row's slot in place (rank[i] := cell_start[cell_of[i]] + rank[i]): the row -> slot map
// k_unpermute needs.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_scatter(const T* __restrict__ pts, int n, const unsigned* __restrict__ cell_of,
                                                    unsigned* rank, const unsigned* __restrict__ cell_start,
                                                    Pt4<T>* __restrict__ sorted) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    if (cell_of[i] == 0xffffffffu) return;            // not part of a closed (sub-box) index
    Pt4<T> p;
    p.x = pts[3 * (size_t)i]; p.y = pts[3 * (size_t)i + 1]; p.z = pts[3 * (size_t)i + 2]; p.idx = i;
    const unsigned pos = cell_start[cell_of[i]] + rank[i];
    sorted[pos] = p;
    put_xyz(sorted, n, pos, p);
    rank[i] = pos;
}

// ---- bucketed build (the default for the whole-cloud indexes) -------------------------------------------------------
// The count/scan/scatter pipeline above spends most of its time in one returning *device-scope* atomic per point (1M
// atomics = 41 us on MI355X: they are executed memory-side, 39 MB of write traffic for 4 MB of counters). The bucketed
// build is a two-level MSD counting sort on the linear (snake) cell id that keeps all per-point atomics in LDS:
//   k_bucket_count    blocks of 8192 points: LDS histogram over *buckets* (runs of 2^shift consecutive cells, ~2k
//                     points each); one returning global atomic per (block, non-empty bucket) reserves the block's
//                     slice of the bucket                                     [~60k global atomics instead of 1M]
//   k_bucket_scatter  same blocks: prefix of the bucket totals, LDS rank inside (block, bucket), records written into
//                     `tmp` grouped by bucket (runs of ~8 records)
//   k_bucket_sort     one 512-thread block per bucket: LDS histogram over the bucket's cells, scan -> cell_start; the
//                     bucket's records, parked in LDS, are permuted in place and leave as one coalesced copy
//   k_bucket_large    buckets holding more than kLargeBucket points (clusters, surfaces tangent to a row of cells, a far
//                     outlier) are not sorted by one block: k_bucket_scatter takes their per-cell ranks with the
//                     returning global atomic of the old scheme (wave-aggregated when a whole wave hits one cell),
//                     k_bucket_sort only scans their cell counters, and this grid-strided kernel places the records.
//                     It exits at once when there is no such bucket.
// Any valid cell order gives the same search results; the order inside a cell is arbitrary in both builds.
constexpr int kBkThreads = 1024;                    // 16 waves per block: the passes are latency-bound
#ifndef PCU_BK_PTS
#define PCU_BK_PTS 8
#endif
constexpr int kBkPts = PCU_BK_PTS;                  // points per thread of the bucket passes (tuning knob)
constexpr int kBkBlockPts = kBkThreads * kBkPts;    // 8192 points per block (4096: +4 % step time, 16384: +2 %)
constexpr int kBkMaxBuckets = 8192;                 // LDS: 32 KB (count) / 64 KB (scatter) of bucket counters per 1024-thread block
constexpr int kBkMaxCellsPerBucket = 4096;
constexpr int kSortThreads = PCU_SORT_THREADS;
constexpr int kSortIters = 8;                       // (16 costs 8 more VGPRs: 3 instead of 4 resident blocks per CU)
constexpr unsigned kLargeBucket = kSortThreads * kSortIters;      // 8192 points = twice the mean bucket
#ifndef PCU_BUCKET_PTS
#define PCU_BUCKET_PTS 4096
#endif
#ifndef PCU_STAGE_RECS
#define PCU_STAGE_RECS 4352
#endif
constexpr int kBucketPts = PCU_BUCKET_PTS;          // expected points per bucket (tuning knob, with PCU_SORT_THREADS and PCU_STAGE_RECS)
constexpr int kStageRecs = PCU_STAGE_RECS;          // records of a bucket staged in LDS for the coalesced copy-out (mean bucket 4096, sigma 64: the stage holds +4 sigma)

template <typename T> struct RawRec;
template <> struct RawRec<float>  { typedef unsigned __attribute__((ext_vector_type(4))) type; };
template <> struct RawRec<double> { typedef unsigned long long __attribute__((ext_vector_type(4))) type; };

template <typename T>
__device__ __forceinline__ unsigned cell_linear(const GridParams<T>& g, T x, T y, T z) {
    const int cx = grid_cell(g, 0, x), cy = grid_cell(g, 1, y), cz = grid_cell(g, 2, z);
    return (unsigned)row_run_lo(g.G[0], grid_row(g.G[1], cy, cz), cx, cx);
}

// counter[key] += 1 for the lanes with `valid`, returning the lane's rank. A wave whose valid lanes all carry the same
// key issues one atomic. Must be called from wave-uniform control flow.
__device__ __forceinline__ unsigned count_rank(unsigned* counter, unsigned key, bool valid) {
    const unsigned long long act = __ballot(valid);
    if (act == 0) return 0;
    const int lane = threadIdx.x & 63, first = __ffsll((long long)act) - 1;
    const unsigned k0 = (unsigned)__shfl((int)key, first, 64);
    unsigned r = 0;
    if (__all(!valid || key == k0)) {
        unsigned base = 0;
        if (lane == first) base = atomicAdd(&counter[k0], (unsigned)__popcll(act));
        base = (unsigned)__shfl((int)base, first, 64);
        r = base + (unsigned)__popcll(act & ((1ull << lane) - 1ull));
    } else if (valid) {
        r = atomicAdd(&counter[key], 1u);
    }
    return r;
}

// The same for a GLOBAL counter array when many lanes of a wave may share few keys (records of over-full buckets: a cluster is a
// handful of cells): up to kAggRounds distinct keys are served by one atomic each (ballot of the lanes with the leader's key), the rest
// lane by lane. 100k points of a tight cluster then cost thousands of atomics on their few cell counters instead of 100k
// (k_bucket_scatter on the 10 % cluster cloud: 490 -> 40 us). Must be called from wave-uniform control flow.
__device__ __forceinline__ unsigned count_rank_agg(unsigned* counter, unsigned key, bool valid) {
    constexpr int kAggRounds = 8;
    const int lane = threadIdx.x & 63;
    unsigned r = 0;
    unsigned long long todo = __ballot(valid);
    for (int round = 0; round < kAggRounds && todo; ++round) {
        const int first = __ffsll((long long)todo) - 1;
        const unsigned k0 = (unsigned)__shfl((int)key, first, 64);
        const unsigned long long same = __ballot(valid && key == k0) & todo;
        unsigned base = 0;
        if (lane == first) base = atomicAdd(&counter[k0], (unsigned)__popcll(same));
        base = (unsigned)__shfl((int)base, first, 64);
        if ((same >> lane) & 1ull) r = base + (unsigned)__popcll(same & ((1ull << lane) - 1ull));
        todo &= ~same;
    }
    if ((todo >> lane) & 1ull) r = atomicAdd(&counter[key], 1u);
    return r;
}

template <typename T>
__device__ __forceinline__ void bucket_count_body(const int bid, const T* __restrict__ pts, int n, const GridParams<T>* __restrict__ gp, int shift,
                                                             unsigned* bucket_total, unsigned* __restrict__ block_base, int nb_stride) {
    __shared__ unsigned s_cnt[kBkMaxBuckets];
    const GridParams<T>& g = *gp;
    const int NB = (g.ncells + (1 << shift) - 1) >> shift;
    for (int i = threadIdx.x; i < NB; i += kBkThreads) s_cnt[i] = 0;
    __syncthreads();
    const int base = bid * kBkBlockPts;
    T px[kBkPts], py[kBkPts], pz[kBkPts];            // all loads first (clamped index, no branch): one wait instead of kBkPts
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const int i = min(base + j * kBkThreads + (int)threadIdx.x, n - 1);
        px[j] = pts[3 * (size_t)i]; py[j] = pts[3 * (size_t)i + 1]; pz[j] = pts[3 * (size_t)i + 2];
    }
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const bool valid = base + j * kBkThreads + (int)threadIdx.x < n;
        const unsigned b = cell_linear(g, px[j], py[j], pz[j]) >> shift;
        (void)count_rank(s_cnt, b, valid);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < NB; i += kBkThreads) {
        const unsigned c = s_cnt[i];
        block_base[(size_t)bid * nb_stride + i] = c ? atomicAdd(&bucket_total[i], c) : 0u;
    }
}

template <typename T>
struct BucketSide {        // one cloud's view of the bucket passes
    const T* pts; int n; GridParams<T>* gp; int shift; int nb_stride;
    unsigned *bucket_total, *block_base, *bucket_start; Pt4<T>* tmp; unsigned *cell_start, *rank_tmp;
    Pt4<T>* sorted; unsigned *pos_of, *large_list, *n_large;
    unsigned cap;          // > 0: one-pass build -- bucket b owns the slot tmp[b * cap, (b + 1) * cap), bucket_total[b] = its fill (k_bucket_onepass)
};
template <typename T>
__global__ __launch_bounds__(kBkThreads) void k_bucket_count(const BucketSide<T> a0, const BucketSide<T> a1, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const BucketSide<T>& a = second ? a1 : a0;
    bucket_count_body<T>(second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, a.pts, a.n, a.gp, a.shift, a.bucket_total, a.block_base, a.nb_stride);
}


// One-pass variant of count + scatter (the default for whole-call builds): every bucket owns a fixed slot of `cap` records in
// `tmp` (cap = kLargeBucket = twice the expected fill), so a block can reserve its share of a bucket with the one returning
// atomic per (block, non-empty bucket) that k_bucket_count spends anyway and write its records at once -- the points are read
// once instead of twice and one launch goes away. No bucket start is known yet: k_bucket_sort derives its output offset from
// the fills. A bucket that would overflow its slot (clusters, surfaces: the data the two-pass build has k_bucket_large for)
// raises GridParams::has_large = 2: the index is then incomplete, every search pass gives up at once, and the host rebuilds
// with the two-pass pipeline and keeps to it for this context (pcu_hip.hip: search_finish).
template <typename T>
__device__ __forceinline__ void bucket_onepass_body(const int bid, const T* __restrict__ pts, int n, GridParams<T>* gp, int shift, unsigned* fill,
                                                   Pt4<T>* __restrict__ tmp, const unsigned cap, long long* prof, const GridSide<T>& gs) {
    // diagnostics (PCU_HIP_PROF_BUILD): per-stage time of every block's thread 0, summed; 100 MHz ticks
    long long t_prev = prof ? wall_clock64() : 0;
#define OP_PROF(slot) do { if (prof && threadIdx.x == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&prof[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
    __shared__ unsigned s_cnt[kBkMaxBuckets];      // the block's count per bucket, then the slot position of its first record
    __shared__ GridParams<T> s_gp;
    const int base = bid * kBkBlockPts;
    T px[kBkPts], py[kBkPts], pz[kBkPts];          // (the point loads do not need the grid: requested first, in flight during the layout)
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const int i = min(base + j * kBkThreads + (int)threadIdx.x, n - 1);
        px[j] = pts[3 * (size_t)i]; py[j] = pts[3 * (size_t)i + 1]; pz[j] = pts[3 * (size_t)i + 2];
    }
    // Every block folds the bbox partials and lays out the grid for itself (identical arithmetic on identical inputs: identical grids);
    // the side's first block also publishes it for the kernels that follow. One launch (k_make_grid, ~5 us of launch boundary + a
    // serial layout on an otherwise idle chip) less per build.
    make_grid_body<T, kBkThreads>(&s_gp, gs.partial, gs.nparts, gs.n, gs.occupancy, gs.max_cells, nullptr, gs.h_want, /*accumulators=*/false, gs.pts);
    if (threadIdx.x == 0 && bid == 0) {
        GridParams<T>& o = *gp;
        for (int j = 0; j < 3; ++j) { o.gmin[j] = s_gp.gmin[j]; o.gmax[j] = s_gp.gmax[j]; o.slack[j] = s_gp.slack[j]; o.G[j] = s_gp.G[j]; o.org[j] = s_gp.org[j]; }
        o.h = s_gp.h; o.inv_h = s_gp.inv_h; o.ncells = s_gp.ncells; o.closed = 0; o.nonfinite = s_gp.nonfinite;
        put_sentinels(gs.sentinel - gs.n, gs.n);
    }
    __syncthreads();
    const GridParams<T>& g = s_gp;
    const int NB = (g.ncells + (1 << shift) - 1) >> shift;
    for (int i = threadIdx.x; i < NB; i += kBkThreads) s_cnt[i] = 0;
    __syncthreads();
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); OP_PROF(0); }          // head: point loads + grid layout + zero
    unsigned bk[kBkPts], rk[kBkPts];
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const bool valid = base + j * kBkThreads + (int)threadIdx.x < n;
        bk[j] = cell_linear(g, px[j], py[j], pz[j]) >> shift;
        rk[j] = count_rank(s_cnt, bk[j], valid);
    }
    __syncthreads();
    OP_PROF(1);                                                                          // keys + LDS ranks
    for (int i = threadIdx.x; i < NB; i += kBkThreads) {
        const unsigned c = s_cnt[i];
        if (c) {
            const unsigned at = atomicAdd(&fill[i], c);
            if (at + c > cap) gp->has_large = 2;
            s_cnt[i] = (unsigned)i * cap + at;
        }
    }
    __syncthreads();
    OP_PROF(2);                                                                          // slot reservations (global atomics)
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const int i = base + j * kBkThreads + (int)threadIdx.x;
        const unsigned pos = s_cnt[bk[j]] + rk[j];
        if (i < n && pos < (bk[j] + 1u) * cap) { Pt4<T> p; p.x = px[j]; p.y = py[j]; p.z = pz[j]; p.idx = i; tmp[pos] = p; }
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); OP_PROF(3); if (threadIdx.x == 0) atomicAdd((unsigned long long*)&prof[7], 1ull); }   // stores
#undef OP_PROF
}
template <typename T>
__global__ __launch_bounds__(kBkThreads) void k_bucket_onepass(const BucketSide<T> a0, const BucketSide<T> a1, int nb0, long long* prof, const GridSide<T> g0, const GridSide<T> g1) {
    const bool second = (int)blockIdx.x >= nb0;
    const BucketSide<T>& a = second ? a1 : a0;
    bucket_onepass_body<T>(second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, a.pts, a.n, a.gp, a.shift, a.bucket_total, a.tmp, a.cap, prof, second ? g1 : g0);
}

template <typename T>
__device__ __forceinline__ void bucket_scatter_body(const int bid, const T* __restrict__ pts, int n, const GridParams<T>* __restrict__ gp, int shift,
                                                               const unsigned* __restrict__ bucket_total, const unsigned* __restrict__ block_base,
                                                               int nb_stride, unsigned* __restrict__ bucket_start, Pt4<T>* __restrict__ tmp,
                                                               unsigned* cell_counts, unsigned* __restrict__ rank_tmp) {
    __shared__ unsigned s_cnt[kBkMaxBuckets];      // running count of this block per bucket
    __shared__ unsigned s_off[kBkMaxBuckets];      // slot of this block's first record in the bucket; bit 31: large bucket
    const GridParams<T>& g = *gp;
    const int NB = (g.ncells + (1 << shift) - 1) >> shift;
    const int base = bid * kBkBlockPts;
    // the point loads do not depend on the prefix below: issued first, all together (clamped index instead of a branch)
    T px[kBkPts], py[kBkPts], pz[kBkPts];
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const int i = min(base + j * kBkThreads + (int)threadIdx.x, n - 1);
        px[j] = pts[3 * (size_t)i]; py[j] = pts[3 * (size_t)i + 1]; pz[j] = pts[3 * (size_t)i + 2];
    }
    {   // exclusive prefix of the bucket totals (every block computes it: NB <= 4096 values)
        const int per = (NB + kBkThreads - 1) / kBkThreads;
        const int i0 = (int)threadIdx.x * per;
        unsigned loc = 0;
        for (int q = 0; q < per; ++q) if (i0 + q < NB) loc += bucket_total[i0 + q];
        unsigned total;
        unsigned ex = block_exclusive_scan_nt<kBkThreads>(loc, &total);
        for (int q = 0; q < per; ++q) {
            const int i = i0 + q;
            if (i < NB) {
                const unsigned t = bucket_total[i];
                s_off[i] = (ex + block_base[(size_t)bid * nb_stride + i]) | (t > kLargeBucket ? 0x80000000u : 0u);
                s_cnt[i] = 0;
                if (bid == 0) bucket_start[i] = ex;
                ex += t;
            }
        }
        if (bid == 0 && threadIdx.x == 0) bucket_start[NB] = total;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < kBkPts; ++j) {
        const int i = base + j * kBkThreads + (int)threadIdx.x;
        const bool valid = i < n;
        Pt4<T> p; p.x = px[j]; p.y = py[j]; p.z = pz[j]; p.idx = i;
        const unsigned c = cell_linear(g, p.x, p.y, p.z);
        const unsigned b = c >> shift;
        const unsigned r = count_rank(s_cnt, b, valid);
        unsigned so = 0;
        if (valid) so = s_off[b];
        const unsigned pos = (so & 0x7fffffffu) + r;
        if (valid) tmp[pos] = p;
        const bool lg = valid && (so >> 31);
        if (__any(lg)) {               // large bucket: per-cell rank by the returning global atomic
            const unsigned rk = count_rank_agg(cell_counts, c, lg);
            if (lg) rank_tmp[pos] = rk;
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kBkThreads) void k_bucket_scatter(const BucketSide<T> a0, const BucketSide<T> a1, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const BucketSide<T>& a = second ? a1 : a0;
    bucket_scatter_body<T>(second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, a.pts, a.n, a.gp, a.shift, a.bucket_total, a.block_base, a.nb_stride,
                           a.bucket_start, a.tmp, a.cell_start, a.rank_tmp);
}


template <typename T>
__device__ __forceinline__ void bucket_sort_body(const int bid, GridParams<T>* gp, int shift, const unsigned* __restrict__ bucket_start,
                                                              const Pt4<T>* __restrict__ tmp, unsigned* cell_start, Pt4<T>* __restrict__ sorted,
                                                              unsigned* __restrict__ pos_of, unsigned* __restrict__ large_list, unsigned* n_large,
                                                              long long* prof, const int cnt_cap, const unsigned cap, const unsigned* __restrict__ fill, const int n_pts) {
    // diagnostics (PCU_HIP_PROF_BUILD): per-stage time of every block's thread 0, summed; 100 MHz ticks
    long long t_prev = prof ? wall_clock64() : 0;
#define BK_PROF(slot) do { if (prof && threadIdx.x == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&prof[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
    // dynamic LDS: [cnt_cap counters][kStageRecs records] (cnt_cap = the largest 2^shift of the launch's sides)
    extern __shared__ __attribute__((aligned(32))) unsigned char s_dyn[];
    unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_dyn);
    Pt4<T>* const s_stage = reinterpret_cast<Pt4<T>*>(s_dyn + (size_t)cnt_cap * 4);
    __shared__ unsigned s_w[kSortThreads / 64 + 1];
    __shared__ unsigned long long s_q[kSortThreads / 64];
    const GridParams<T>& g = *gp;
    const int CB = 1 << shift;
    const int NB = (g.ncells + CB - 1) >> shift;
    const int b = bid;
    if (b >= NB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned c0 = (unsigned)b << shift;
    const int ncl = min(CB, g.ncells - (int)c0);
    // records of the bucket: tmp[s_in, s_in + (e - s)); its slice of `sorted`: [s, e)
    unsigned s, e, s_in;
    if (cap) {            // one-pass build: the bucket's slot; its output offset = the fills of the buckets before it
        unsigned part = 0;
        for (int i = tid; i < b; i += kSortThreads) part += min(fill[i], cap);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) s_w[wave] = part;
        __syncthreads();
        s = 0;
        for (int w = 0; w < kSortThreads / 64; ++w) s += s_w[w];
        __syncthreads();
        e = s + min(fill[b], cap); s_in = (unsigned)b * cap;
    } else { s = bucket_start[b]; e = bucket_start[b + 1]; s_in = s; }
    const bool large = e - s > kLargeBucket;
    BK_PROF(0);
    for (int i = tid; i < CB; i += kSortThreads) s_cnt[i] = (large && i < ncl) ? cell_start[c0 + i] : 0u;
    __syncthreads();
    BK_PROF(1);
    unsigned rr[kSortIters];           // (cell in bucket) << 16 | rank in cell      (small buckets: <= kLargeBucket records)
    // Records are fetched kSortBatch trips at a time, all loads of a batch in flight together (clamped index instead of a
    // branch around the load); the batch loop ends, wave-uniformly, with the bucket.
    constexpr int kSortBatch = 4;
    const unsigned in_off = s_in - s;           // tmp index = sorted index + in_off (mod 2^32)
    const unsigned last = e > s ? e - 1u : s;
    const bool staged = e - s <= (unsigned)kStageRecs;      // the bucket fits the LDS stage (the normal case)
    if (!large) {
#pragma unroll
        for (int it0 = 0; it0 < kSortIters; it0 += kSortBatch) {
            Pt4<T> rec[kSortBatch];
            const bool batch_on = s + (unsigned)(it0 * kSortThreads) < e;       // uniform in the block
            if (batch_on) {
#pragma unroll
                for (int u = 0; u < kSortBatch; ++u) rec[u] = tmp[min(s + (unsigned)((it0 + u) * kSortThreads + tid), last) + in_off];
            }
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) {
                const unsigned p = s + (unsigned)((it0 + u) * kSortThreads + tid);
                rr[it0 + u] = 0;
                if (batch_on && p < e) {
                    const unsigned c = cell_linear(g, rec[u].x, rec[u].y, rec[u].z) - c0;
                    rr[it0 + u] = (c << 16) | atomicAdd(&s_cnt[c], 1u);
                    if (staged) s_stage[p - s] = rec[u];          // parked in arrival order; permuted in place below
                }
            }
        }
        __syncthreads();
    }
    BK_PROF(2);
    // exclusive scan of the CB counters (thread-contiguous items), cell_start, balance metric
    const int per = CB >= kSortThreads ? CB / kSortThreads : 1;
    const int i0 = tid * per;
    unsigned v[kBkMaxCellsPerBucket / kSortThreads], sum = 0; unsigned long long sq = 0;
#pragma unroll
    for (int q = 0; q < kBkMaxCellsPerBucket / kSortThreads; ++q) {
        v[q] = (q < per && i0 + q < CB) ? s_cnt[i0 + q] : 0u;
        sum += v[q]; sq += (unsigned long long)v[q] * v[q];
    }
    unsigned inc = sum;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
    if (lane == 63) s_w[wave] = inc;
    if (lane == 0) s_q[wave] = sq;
    __syncthreads();
    if (tid == 0) {
        unsigned run = 0; unsigned long long Q = 0;
        for (int w = 0; w < kSortThreads / 64; ++w) { const unsigned t = s_w[w]; s_w[w] = run; run += t; Q += s_q[w]; }
        if (Q) atomicAdd(&gp->sumsq, Q);
        if (large) { large_list[atomicAdd(n_large, 1u)] = (unsigned)b; gp->has_large = 1; }
        if (b == NB - 1) cell_start[g.ncells] = e;
    }
    __syncthreads();
    unsigned ex = inc - sum + s_w[wave];
#pragma unroll
    for (int q = 0; q < kBkMaxCellsPerBucket / kSortThreads; ++q) {
        if (q < per && i0 + q < CB) {
            s_cnt[i0 + q] = ex;
            if (i0 + q < ncl) cell_start[c0 + i0 + q] = s + ex;
            ex += v[q];
        }
    }
    if (large) return;
    __syncthreads();
    BK_PROF(3);
    // Placement. A bucket that fits the LDS stage is permuted there, in place (every thread takes its records out, barrier,
    // puts them into their sorted slots), and leaves as one contiguous, fully coalesced copy: no second pass over `tmp`
    // and whole 128-byte lines instead of one scattered 16-byte store per record. Larger buckets re-read and store directly.
    if (staged) {
        constexpr int kStageIters = (kStageRecs + kSortThreads - 1) / kSortThreads;       // trips that cover the stage
        static_assert(kStageIters <= kSortIters, "rr[] holds one rank per trip");
        typedef typename RawRec<T>::type Raw;                 // a record as one vector register tuple (a struct copy goes through scratch)
        Raw* const raw = reinterpret_cast<Raw*>(s_stage);
        const unsigned cnt = e - s;
        const unsigned lastl = cnt ? cnt - 1u : 0u;
        Raw m[kStageIters];
#pragma unroll
        for (int u = 0; u < kStageIters; ++u) m[u] = raw[min((unsigned)(tid + u * kSortThreads), lastl)];
        __syncthreads();
#pragma unroll
        for (int u = 0; u < kStageIters; ++u)
            if ((unsigned)(tid + u * kSortThreads) < cnt) raw[s_cnt[rr[u] >> 16] + (rr[u] & 0xffffu)] = m[u];
        __syncthreads();
        for (unsigned i = tid; i < e - s; i += kSortThreads) {
            const Pt4<T> r = s_stage[i];
            sorted[s + i] = r;
            put_xyz(sorted, n_pts, s + i, r);
            if (pos_of) pos_of[r.idx] = s + i;
        }
    } else {
#pragma unroll
        for (int it0 = 0; it0 < kSortIters; it0 += kSortBatch) {
            if (!(s + (unsigned)(it0 * kSortThreads) < e)) break;                  // uniform in the block
            Pt4<T> rec[kSortBatch];
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) rec[u] = tmp[min(s + (unsigned)((it0 + u) * kSortThreads + tid), last) + in_off];
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) {
                const unsigned p = s + (unsigned)((it0 + u) * kSortThreads + tid);
                if (p < e) {
                    const unsigned pos = s + s_cnt[rr[it0 + u] >> 16] + (rr[it0 + u] & 0xffffu);
                    sorted[pos] = rec[u];
                    put_xyz(sorted, n_pts, pos, rec[u]);
                    if (pos_of) pos_of[rec[u].idx] = pos;
                }
            }
        }
    }
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); BK_PROF(4); if (threadIdx.x == 0) atomicAdd((unsigned long long*)&prof[7], 1ull); }
#undef BK_PROF
}

template <typename T>
__global__ __launch_bounds__(kSortThreads) void k_bucket_sort(const BucketSide<T> a0, const BucketSide<T> a1, int nb0, long long* prof, int cnt_cap) {
    const bool second = (int)blockIdx.x >= nb0;
    const BucketSide<T>& a = second ? a1 : a0;
    bucket_sort_body<T>(second ? (int)blockIdx.x - nb0 : (int)blockIdx.x, a.gp, a.shift, a.bucket_start, a.tmp, a.cell_start, a.sorted, a.pos_of,
                        a.large_list, a.n_large, prof, cnt_cap, a.cap, a.bucket_total, a.n);
}
template <typename T>
static size_t bucket_sort_lds_bytes(int cnt_cap) { return (size_t)cnt_cap * 4 + (size_t)kStageRecs * sizeof(Pt4<T>) + 64; }      // (+ 64: grid2.h's packed stage starts up to 16 bytes in)


template <typename T>
struct LargeJob {
    const GridParams<T>* gp; const unsigned* bucket_start; const unsigned* large_list; const unsigned* n_large;
    const Pt4<T>* tmp; const unsigned* rank_tmp; const unsigned* cell_start; Pt4<T>* sorted; unsigned* pos_of; int n_pts;
};
// One launch serves the indexes built back to back (both clouds of a two-sided call): njobs <= 2.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_bucket_large(const LargeJob<T> j0, const LargeJob<T> j1, int njobs) {
    // (bucket, slice) pairs dealt to the blocks: every over-full bucket is placed by kLargeSlices blocks at once and many buckets are in
    // flight together. (A loop over the buckets with the whole grid striding each was a chain of nl dependent round trips: 111 us for
    // the ~100 over-full buckets of a Gaussian cloud.)
    constexpr unsigned kLargeSlices = 16;
    for (int jj = 0; jj < njobs; ++jj) {
        const LargeJob<T>& J = jj ? j1 : j0;
        const unsigned nl = *J.n_large;
        if (nl == 0) continue;
        const GridParams<T>& g = *J.gp;
        for (unsigned v = blockIdx.x; v < nl * kLargeSlices; v += gridDim.x) {
            const unsigned b = J.large_list[v / kLargeSlices], sub = v % kLargeSlices;
            const unsigned s = J.bucket_start[b], e = J.bucket_start[b + 1];
            for (unsigned p = s + sub * kBlock + threadIdx.x; p < e; p += kLargeSlices * kBlock) {
                const Pt4<T> rec = J.tmp[p];
                const unsigned pos = J.cell_start[cell_linear(g, rec.x, rec.y, rec.z)] + J.rank_tmp[p];
                J.sorted[pos] = rec;
                put_xyz(J.sorted, J.n_pts, pos, rec);
                if (J.pos_of) J.pos_of[rec.idx] = pos;
            }
        }
    }
}

// ---- refitting the grid of an unbalanced cloud -----------------------------------------------------------------------
// When the first (bbox-filling, mean-density) grid turns out badly unbalanced -- clusters, blobs, a far outlier that
// inflates the bbox -- finer grids are fitted to the *core* of the cloud: three rounds of per-axis 1024-bin
// histograms, each over the range that held all but ~2/4096 of the mass in the previous round (1024^3 dynamic
// range), give the core range; k_make_grid_refit lays `target_cells` cubic cells over it. Points outside the core
// fall into the border cells. The grid only decides which candidates are looked at, never the result.
constexpr int kHistBins = 1024;            // + underflow bin 0 and overflow bin kHistBins + 1
constexpr int kHistBlocks = 128;

template <typename T>
struct QuantState { T lo[3], hi[3]; };     // histogram range of the current round

template <typename T>
__global__ void k_quant_init(const GridParams<T>* gp, QuantState<T>* qs) {
    if (threadIdx.x < 3) { qs->lo[threadIdx.x] = gp->gmin[threadIdx.x]; qs->hi[threadIdx.x] = gp->gmax[threadIdx.x]; }
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_hist_axis(const T* __restrict__ pts, int n, const QuantState<T>* __restrict__ qs,
                                                      unsigned* __restrict__ partial) {
    __shared__ unsigned h[3][kHistBins + 2];
    for (int i = threadIdx.x; i < 3 * (kHistBins + 2); i += kBlock) (&h[0][0])[i] = 0;
    __syncthreads();
    T lo[3], sc[3];
    for (int a = 0; a < 3; ++a) { lo[a] = qs->lo[a]; const T w = qs->hi[a] - qs->lo[a]; sc[a] = w > 0 ? (T)kHistBins / w : (T)0; }
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            const T t = (pts[3 * (size_t)i + a] - lo[a]) * sc[a];
            const int b = (t >= 0) ? ((t < (T)kHistBins) ? 1 + (int)t : kHistBins + 1) : 0;
            atomicAdd(&h[a][b], 1u);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < 3 * (kHistBins + 2); i += kBlock) partial[(size_t)blockIdx.x * 3 * (kHistBins + 2) + i] = (&h[0][0])[i];
}

static __global__ __launch_bounds__(kBlock) void k_hist_merge(const unsigned* __restrict__ partial, int nblocks, unsigned* __restrict__ hist) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= 3 * (kHistBins + 2)) return;
    unsigned s = 0;
    for (int b = 0; b < nblocks; ++b) s += partial[(size_t)b * 3 * (kHistBins + 2) + i];
    hist[i] = s;
}

// Per axis (one wave each: launch 3 blocks of 64): zoom the range to the bins that hold all but n/4096 of the mass at either
// end. b0 = number of leading bins whose cumulative count (with the underflow bin) stays <= tail, b1 likewise from the top;
// the cumulative counts are monotone, so both are plain counts over a wave-wide prefix / suffix scan (16 bins per lane).
template <typename T>
__global__ __launch_bounds__(64) void k_quant_zoom(QuantState<T>* qs, const unsigned* __restrict__ hist, int n) {
    const int a = blockIdx.x, lane = threadIdx.x;
    if (a >= 3) return;
    const unsigned* h = hist + a * (kHistBins + 2);
    const T lo = qs->lo[a], hi = qs->hi[a];
    const double w = ((double)hi - (double)lo) / kHistBins;
    if (!(w > 0)) return;
    const double tail = (double)n / 4096.0;
    constexpr int kPer = kHistBins / 64;                   // 16 bins per lane
    unsigned v[kPer]; unsigned long long loc = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) { v[i] = h[1 + lane * kPer + i]; loc += v[i]; }
    // prefix: P[i] = underflow + sum of bins 0..i
    unsigned long long inc = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    unsigned long long run = inc - loc + h[0];
    int c0 = 0;
#pragma unroll
    for (int i = 0; i < kPer; ++i) { run += v[i]; c0 += ((double)run <= tail) ? 1 : 0; }
    // suffix: S[i] = overflow + sum of bins i..last
    unsigned long long dec = loc;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned long long t = __shfl_down(dec, o, 64); if (lane + o < 64) dec += t; }
    unsigned long long runs = dec - loc + h[kHistBins + 1];
    int c1 = 0;
#pragma unroll
    for (int i = kPer - 1; i >= 0; --i) { runs += v[i]; c1 += ((double)runs <= tail) ? 1 : 0; }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { c0 += __shfl_xor(c0, o, 64); c1 += __shfl_xor(c1, o, 64); }
    if (lane == 0) {
        const int b0 = c0 < kHistBins - 1 ? c0 : kHistBins - 1;
        int b1 = kHistBins - 1 - c1;
        if (b1 < b0) b1 = b0;
        qs->lo[a] = (T)((double)lo + b0 * w); qs->hi[a] = (T)((double)lo + (b1 + 1) * w);
    }
}

// Cubic cells of about `target_cells` over the core range qs (exact bbox kept in gmin/gmax for certification).
template <typename T>
__global__ void k_make_grid_refit(GridParams<T>* gp, const GridParams<T>* base, const QuantState<T>* qs, double target_cells,
                                  int max_cells, Pt4<T>* sentinel, int n_sorted, int closed, const double* target_dev) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    if (target_dev) target_cells = *target_dev;
    if (sentinel) put_sentinels(sentinel - n_sorted, n_sorted);
    double ext[3];
    for (int j = 0; j < 3; ++j) { gp->gmin[j] = base->gmin[j]; gp->gmax[j] = base->gmax[j]; gp->org[j] = qs->lo[j]; ext[j] = (double)qs->hi[j] - (double)qs->lo[j]; }
    double emax = ext[0] > ext[1] ? (ext[0] > ext[2] ? ext[0] : ext[2]) : (ext[1] > ext[2] ? ext[1] : ext[2]);
    double want = target_cells;
    if (want > (double)max_cells) want = (double)max_cells;
    if (want < 1.0) want = 1.0;
    int G[3] = {1, 1, 1};
    double h = 1.0;
    if (emax > 0 && isfinite(emax)) {
        bool act[3]; int nd = 0; double vol = 1.0;
        for (int j = 0; j < 3; ++j) { act[j] = ext[j] > emax * 1e-6; if (act[j]) { ++nd; vol *= ext[j]; } }
        h = pow(vol / want, 1.0 / nd);
        for (int it = 0; it < 400; ++it) {
            double cells = 1.0;
            for (int j = 0; j < 3; ++j) {
                double g = act[j] ? floor(ext[j] / h) + 1.0 : 1.0;
                if (g > 2048.0) g = 2048.0;
                G[j] = (int)g; cells *= g;
            }
            if (cells <= (double)max_cells) break;
            h *= 1.05;
        }
    }
    gp->h = (T)h;
    gp->inv_h = (T)1 / gp->h;
    for (int j = 0; j < 3; ++j) {
        gp->G[j] = G[j];
        double scale = fabs((double)gp->org[j]) + fabs((double)qs->hi[j]) + (double)G[j] * h;
        gp->slack[j] = (T)(8.0 * (double)Limits<T>::eps * scale);
    }
    gp->ncells = G[0] * G[1] * G[2];
    gp->sumsq = 0ull; gp->closed = closed; gp->has_large = 0; gp->nonfinite = base->nonfinite;
}

// Heavy part of an indexed cloud: bounding box (+1 cell) and number of the points that sit in cells holding more than
// `thresh` points, and the cell count a sub-box grid over it should get: the parent's cells inside the box times how
// overfull the heavy cells are. Two stages: per-block partials {lo[3], hi[3], count, sum of cell counts}, one block folds.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_heavy_partial(const T* __restrict__ pts, int n, const GridParams<T>* __restrict__ gp, const unsigned* __restrict__ cell_of,
                                                          const unsigned* __restrict__ cell_start, unsigned thresh,
                                                          T* __restrict__ pbox, double* __restrict__ pcnt) {
    T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v}, hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
    double cnt = 0, sq = 0;
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        T v[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) v[j] = pts[3 * (size_t)i + j];
        // (cell_of: what k_count stored; the bucketed builds keep no such array -- an open grid holds every point: the cell is recomputed)
        const unsigned c = cell_of ? cell_of[i]
                                   : (unsigned)row_run_lo(gp->G[0], grid_row(gp->G[1], grid_cell(*gp, 1, v[1]), grid_cell(*gp, 2, v[2])), grid_cell(*gp, 0, v[0]), grid_cell(*gp, 0, v[0]));
        if (c == 0xffffffffu) continue;
        const unsigned k = cell_start[c + 1] - cell_start[c];
        if (k <= thresh) continue;
        cnt += 1; sq += k;
#pragma unroll
        for (int j = 0; j < 3; ++j) { lo[j] = v[j] < lo[j] ? v[j] : lo[j]; hi[j] = v[j] > hi[j] ? v[j] : hi[j]; }
    }
    __shared__ T s_lo[kBlock / 64][3], s_hi[kBlock / 64][3]; __shared__ double s_c[kBlock / 64], s_q[kBlock / 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) { const T a = wave_min(lo[j]), b = wave_max(hi[j]); if (lane == 0) { s_lo[wave][j] = a; s_hi[wave][j] = b; } }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { cnt += __shfl_xor(cnt, o, 64); sq += __shfl_xor(sq, o, 64); }
    if (lane == 0) { s_c[wave] = cnt; s_q[wave] = sq; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double C = 0, Q = 0;
        for (int w = 0; w < kBlock / 64; ++w) { C += s_c[w]; Q += s_q[w]; }
        for (int j = 0; j < 3; ++j) {
            T a = s_lo[0][j], b = s_hi[0][j];
            for (int w = 1; w < kBlock / 64; ++w) { a = s_lo[w][j] < a ? s_lo[w][j] : a; b = s_hi[w][j] > b ? s_hi[w][j] : b; }
            pbox[blockIdx.x * 6 + j] = a; pbox[blockIdx.x * 6 + 3 + j] = b;
        }
        pcnt[blockIdx.x * 2] = C; pcnt[blockIdx.x * 2 + 1] = Q;
    }
}
template <typename T>
__global__ void k_heavy_finish(const GridParams<T>* __restrict__ gp, const T* __restrict__ pbox, const double* __restrict__ pcnt, int nparts,
                               double occ, double cap, QuantState<T>* out, double* out_target) {
    // one wave (launched with 64 threads): the lanes stride over the partials, then fold (counts are integers held in doubles: exact)
    if (blockIdx.x != 0 || threadIdx.x >= 64) return;
    double C = 0, Q = 0;
    T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v}, hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
    for (int b = threadIdx.x; b < nparts; b += 64) {
        C += pcnt[2 * b]; Q += pcnt[2 * b + 1];
        for (int j = 0; j < 3; ++j) { const T a = pbox[b * 6 + j], c = pbox[b * 6 + 3 + j]; lo[j] = a < lo[j] ? a : lo[j]; hi[j] = c > hi[j] ? c : hi[j]; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { C += __shfl_xor(C, o, 64); Q += __shfl_xor(Q, o, 64); }
#pragma unroll
    for (int j = 0; j < 3; ++j) { lo[j] = wave_min(lo[j]); hi[j] = wave_max(hi[j]); }
    if (threadIdx.x != 0) return;
    double cells_in_box = 1;
    for (int j = 0; j < 3; ++j) {
        T a = lo[j], b = hi[j];
        if (!(a <= b)) { a = 0; b = 0; }
        out->lo[j] = a - gp->h; out->hi[j] = b + gp->h;
        cells_in_box *= (gp->G[j] > 1) ? ((double)(b - a) / (double)gp->h + 2.0) : 1.0;
    }
    const double overfull = C > 0 ? (Q / C) / occ : 1.0;        // mean count of a heavy point's cell / wanted occupancy
    double t = cells_in_box * overfull;
    if (t < C / occ * 0.25) t = C / occ * 0.25;
    if (t > cap) t = cap;
    if (t < 1) t = 1;
    out_target[0] = t;
    out_target[1] = C;          // (for the host: how much of the parent sits in heavy cells)
}

}  // namespace pcu
