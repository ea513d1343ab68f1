// csrc/search.h -- exact k-nearest-neighbour search over a grid index (the hot loop).
//
// Replaces nanoflann's per-query tree descent (nanoflann.hpp:1393-1418 findNeighbors, :1544-1624 searchLevel,
// :194-227 KNNResultSet::addPoint) as driven by src/point_cloud_distance.cpp:49-95.
//
// One lane = one query, processed in the *query cloud's own cell order* so that the 64 lanes of a wave look at
// the same few rows of dataset cells (L1/L2-resident). A lane scans the (2R+1)^3 block of cells around its
// own cell, row by row: the cells [x0..x1] of one (y,z) row are one contiguous run of `sorted` records
// (snake cell order, pcu_types.h).
//
// Arithmetic contract (bit parity with the reference):
//   d2 = ((dx*dx) + (dy*dy)) + (dz*dz),  dx = q.x - r.x, ...   all in T, no FMA (TU built with
//   -ffp-contract=off), i.e. L2_Simple_Adaptor::evalMetric, nanoflann.hpp:496-507.
//
// Exactness: after the scan the lane knows its k best d2 inside the scanned box of cells. Every unscanned
// dataset point lies beyond one of the box faces that are interior to the grid; `face_lower_bound` computes,
// with the same rounding-monotone arithmetic, a value LB such that the *computed* d2 of any such point is
// >= LB. The result is final ("certified") iff kth_best < LB (strict: a tie with an unscanned point is not
// accepted). Uncertified queries are appended to `unresolved` and re-run by the host loop with a larger R or a
// coarser grid until the scanned box is the whole grid (LB = +inf).
//
// Ties: the lane-per-query passes only *detect* that an exact tie may matter (equal d2 met at the k-th boundary, or equal
// neighbours in the final list) and append the query to `ties`; the wave-per-query pass re-runs those queries under the
// total order (d2, dataset row) with K >= k+1 slots and reports which of them have a genuine tie inside the top-(k+1)
// ("true ties"), for which the order of the reference is defined by its kd-tree traversal (kd_order.h).
#pragma once
#include "pcu_types.h"
#include "grid.h"
#include "reduce.h"

namespace pcu {

template <typename T>
struct SearchArgs {
    const GridParams<T>* gp;        // dataset grid
    const Pt4<T>* ref;              // dataset in cell order
    const T* ref_xyz;               // the same records without the row id (3 T each; GridIndex::xyz), k_search1_flat's candidate stream
    const unsigned* cell_start;     // [ncells+1]
    const Pt4<T>* qsorted;          // queries in their own cell order
    const T* q_xyz; const int* q_idx;   // the same records as two streams (pcu_types.h: xyz_of / idx32_of): coordinates, 32-bit row ids ...
    const int* ref_idx;             // ... and the dataset's row ids. k_search1_flat reads only the streams; with `lean` set the Pt4 arrays `ref` /
    int lean;                       // `qsorted` are NOT written (grid2.h) and k_search_wave reads the streams too. No other kernel takes a lean index.
    const int* qlist;               // nullable: positions into qsorted to process (escalation / tie passes)
    const int* qcount_dev;          // nullable: device-side count for qlist passes
    int nq;                         // number of work items when qcount_dev is null
    int R;                          // search radius in cells
    const int* qlist2; const int* qcount2_dev; int R2;   // wave-per-query passes: a second device-side list with its own radius,
                                    // served by the same launch (work items = list 1 followed by list 2)
    unsigned n_ref;                 // number of dataset records; ref[n_ref] is the +inf sentinel record
    unsigned lane_max_cand;         // a lane whose 27 cells hold more candidates than this hands its query to the wave-per-query
                                    // pass (via the tie list) instead of scanning them serially: one heavy cell next to a query
                                    // must not turn a wave into a millisecond-long pole
    float skew_far;                 // the flag's value is 2 instead of 1 when sumsq > skew_far (far beyond a change of resolution: the host then skips a read-back)
    float skew_limit;               // > 0: a whole-cloud pass gives up at once when the uniform dataset grid is unbalanced
    int* skew_flag;                 //      beyond this (sumsq > limit) and raises the flag; the host then refits the dataset grid (pcu_hip.hip, search_finish)
    float skew_lo;                  //      ... or when it is MORE even than this (sumsq < skew_lo; 0 = off): a grid kept finer than the default for
                                    //      surface-like clouds met a cloud that fills its volume (pcu_hip.hip: rescale_wanted)
    const GridParams<T>* qgp;       // grid of the QUERY cloud: a pass gives up (flag word skew_flag[kLargeFlag] = the OR of both clouds'
                                    // GridParams::has_large) when either cloud's bucketed index is not ready: 1 = over-full buckets still
                                    // unplaced (the host runs k_bucket_large and repeats the pass -- the common case saves that launch),
                                    // 2 = a one-pass build overflowed a bucket slot (the host rebuilds with the two-pass pipeline)
    int kreq;                       // neighbours requested (<= K)
    int squared;                    // write d2 instead of sqrt(d2)
    int row_out;                    // 1: result row of a query goes to its ORIGINAL row (k >= 4: rows are >= 48 B, scattering whole
                                    // rows costs less than a row-order restore pass); 0: to its slot in the queries' cell order
    T* out_d;                       // (nq_total, kreq) in the queries' CELL order: row qpos belongs to qsorted[qpos]
    long long* out_i;               // (nq_total, kreq)  (k_unpermute restores the caller's row order when needed)
    int* unresolved; int* n_unresolved;
    T* ubound;                      // nullable, lane passes: ubound[i] = the k-th best d2 the lane found for unresolved[i] (max_v: fewer than k points) --
                                    // an upper bound of the true one, which lets the wave-per-query pass start with the ball round
    const T* qbound2;               // wave-per-query passes: those bounds for the entries of qlist2
    int* ties;       int* n_ties;         // lane passes: possible tie; wave pass: genuine tie ("true ties")
    // Fused epilogue (reduce.h, FUSE_*): instead of result rows the k = 1 lane pass writes ONE partial per block --
    // FUSE_SUM: fp64 sum of its certified lanes' distances -> f_sum[block]; FUSE_ARGMAX: their arg-max -> f_max_v / f_max_k[block]
    // (key = source row << 32 | tie bit 31 | dataset row). No rows are written and only deferred lanes enter the tie list.
    int fuse;
    double* f_sum; T* f_max_v; long long* f_max_k;
    // ... and the wave-per-query pass adds its few queries: FUSE_SUM exactly, into f_limbs / f_special (reduce.h: exact_add);
    // FUSE_ARGMAX one partial per wave in f_wave_v / f_wave_k[wave]. k_fuse_tail (reduce.h) folds everything.
    unsigned long long* f_limbs; double* f_special; T* f_wave_v; long long* f_wave_k;
    int f_accum;                    // FUSE_ARGMAX, later wave-per-query launches of the same call (pcu_hip.hip: fused_continue): combine with the
                                    // slots instead of overwriting them
    int escalate;                   // wave-per-query passes: finish every query inside the launch (box round, then the ball round: k_search_wave)
    int maxval;                     // fused arg-max: the lane pass runs its value-only variant (FUSE_MAXVAL; k_fuse_tail resolves the winner's neighbour)
    int brick;                      // 1: query cloud and dataset share ONE grid and the call takes search_brick.h's staged pass (fused sum, float)
    int* n_fallback;                // ... whose blocks that did not fit their stage count themselves here (pcu_hip.hip: brick feedback)
    const unsigned* cancel_word; unsigned cancel_gen;      // pcu_types.h: cancel_seen (the wave-per-query pass looks between its work items)
    int bad_r, bad_q;               // GridParams::nonfinite flags (grid.h: kNf*) of the dataset / of the query cloud that this operator rejects:
                                    // the passes give up at once and raise bit 2 of the large-bucket flag word (-> ValueError on the host)
};
// The "index not usable" word of a pass: over-full buckets (1), one-pass overflow (2), rejected non-finite input (4).
template <typename T>
__device__ __forceinline__ int index_not_ready(const SearchArgs<T>& a, const GridParams<T>& g) {
    const GridParams<T>& qg = *a.qgp;
    return g.has_large | qg.has_large | (((g.nonfinite & a.bad_r) | (qg.nonfinite & a.bad_q)) ? 4 : 0);
}

constexpr int kLargeFlag = 3;      // counters[C_LARGE] relative to counters[C_SKEW] (pcu_hip.hip)

// Append `value` for lanes with `flag` set; one atomic per wave. Returns the lane's slot (-1: not appended).
__device__ __forceinline__ int wave_append(bool flag, int value, int* list, int* counter) {
    const unsigned long long m = __ballot(flag);
    if (m == 0) return -1;
    const int lane = threadIdx.x & 63;
    const int leader = __ffsll((long long)m) - 1;
    int base = 0;
    if (lane == leader) base = atomicAdd(counter, __popcll(m));
    base = __shfl(base, leader, 64);
    const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
    if (flag) list[slot] = value;
    return flag ? slot : -1;
}

// The same for a whole workgroup of kBlock threads (EVERY thread of the block must call it): one atomic per block instead of one per
// wave. For lists that nearly every thread of a launch joins -- the queries a closed sub-box level hands on without a scan, 0.9M of 1M on
// the tight-cluster cloud: 14k returning atomics on one counter took ~200 us per level (same-address atomics serialise at one L2 channel).
__device__ __forceinline__ void block_append(bool flag, int value, int* list, int* counter) {
    __shared__ int s_cnt[kBlock / 64], s_base;
    const unsigned long long m = __ballot(flag);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) s_cnt[wave] = __popcll(m);
    __syncthreads();
    if (threadIdx.x == 0) {
        int tot = 0;
#pragma unroll
        for (int w = 0; w < kBlock / 64; ++w) tot += s_cnt[w];
        s_base = tot ? atomicAdd(counter, tot) : 0;
    }
    __syncthreads();
    int off = s_base;
    for (int w = 0; w < wave; ++w) off += s_cnt[w];
    if (flag) list[off + __popcll(m & ((1ull << lane) - 1ull))] = value;
}

// Splits a query list by a closed sub-box level's box (the test of k_search's head): queries inside go to `inside` (the level's lane pass
// runs over that list), the others straight to `outside` (the next level's input). 8192 queries per block, one atomic per block and list:
// on the tight-cluster cloud 0.9M of 1M queries skip each level, and handing them on from inside the lane kernel cost one returning
// atomic per wave (14k on one counter: ~200 us per level), then one per block (3.9k: ~100 us). Now 123 per list and ~10 us.
constexpr int kSplitThreads = 1024, kSplitPer = 8;
template <typename T>
__global__ __launch_bounds__(kSplitThreads) void k_box_split(const Pt4<T>* __restrict__ qsorted, const int* __restrict__ in_list, const int* __restrict__ in_count,
                                                             int nq_all, const GridParams<T>* __restrict__ gp, int* __restrict__ inside, int* n_inside,
                                                             int* __restrict__ outside, int* n_outside) {
    const GridParams<T>& g = *gp;
    const int nq = in_count ? *in_count : nq_all;
    const int base = (int)blockIdx.x * kSplitThreads * kSplitPer;
    if (base >= nq) return;
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    int qp[kSplitPer]; unsigned in_m = 0, out_m = 0;
#pragma unroll
    for (int u = 0; u < kSplitPer; ++u) {
        const int i = base + (int)threadIdx.x * kSplitPer + u;      // (a thread's queries are consecutive: both lists keep the input's -- spatial -- order)
        qp[u] = 0;
        if (i < nq) {
            qp[u] = in_list ? in_list[i] : i;
            const Pt4<T> q = qsorted[qp[u]];
            const T tx = (q.x - g.org[0]) * g.inv_h, ty = (q.y - g.org[1]) * g.inv_h, tz = (q.z - g.org[2]) * g.inv_h;      // (k_search's test, same arithmetic)
            const bool in = tx >= (T)0 && tx < (T)Gx && ty >= (T)0 && ty < (T)Gy && tz >= (T)0 && tz < (T)Gz;
            if (in) in_m |= 1u << u; else out_m |= 1u << u;
        }
    }
    // exclusive prefix over the block of (inside count, outside count), packed 16 + 16 bits (a block holds 8192 queries)
    __shared__ unsigned s_w[kSplitThreads / 64]; __shared__ int s_base[2];
    const unsigned mine = (unsigned)__popc(in_m) | ((unsigned)__popc(out_m) << 16);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned inc = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned before = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kSplitThreads / 64; ++w) { const unsigned v = s_w[w]; if (w < wave) before += v; total += v; }
    if (threadIdx.x == 0) {
        const int ti = (int)(total & 0xffffu), to = (int)(total >> 16);
        s_base[0] = ti ? atomicAdd(n_inside, ti) : 0;
        s_base[1] = to ? atomicAdd(n_outside, to) : 0;
    }
    __syncthreads();
    const unsigned ex = before + inc - mine;
    int pi = s_base[0] + (int)(ex & 0xffffu), po = s_base[1] + (int)(ex >> 16);
#pragma unroll
    for (int u = 0; u < kSplitPer; ++u) {
        if (in_m >> u & 1u) inside[pi++] = qp[u];
        if (out_m >> u & 1u) outside[po++] = qp[u];
    }
}

// Lower bound on the computed d2 of every dataset point whose cell lies outside [c0..c1] (per axis) of the grid.
template <typename T>
__device__ __forceinline__ T face_lower_bound(const GridParams<T>& g, T qx, T qy, T qz,
                                              int x0, int x1, int y0, int y1, int z0, int z1) {
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    const T q[3] = {qx, qy, qz};
    const int c0[3] = {x0, y0, z0}, c1[3] = {x1, y1, z1};
    // distance from q to the data bbox along each axis (0 inside): valid for *every* dataset point
    T o[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T a = g.gmin[j] - q[j], b = q[j] - g.gmax[j];
        T m = a > b ? a : b;
        o[j] = m > (T)0 ? m * shrink : (T)0;
    }
    T lb = INFINITY;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (c0[j] > 0 || g.closed) {  // points with cell_j <= c0-1 (closed sub-box grids: also everything beyond the border): coordinate < gmin + c0*h (+slack)
            T B = face_below(g, j, c0[j]);
            T m = q[j] - B;
            m = m > (T)0 ? m * shrink : (T)0;
            m = m > o[j] ? m : o[j];
            T t0 = j == 0 ? m : o[0], t1 = j == 1 ? m : o[1], t2 = j == 2 ? m : o[2];
            T f = ((t0 * t0) + (t1 * t1)) + (t2 * t2);
            lb = f < lb ? f : lb;
        }
        if (c1[j] < g.G[j] - 1 || g.closed) {     // points with cell_j >= c1+1: coordinate >= gmin + (c1+1)*h (-slack)
            T B = face_above(g, j, c1[j]);
            T m = B - q[j];
            m = m > (T)0 ? m * shrink : (T)0;
            m = m > o[j] ? m : o[j];
            T t0 = j == 0 ? m : o[0], t1 = j == 1 ? m : o[1], t2 = j == 2 ? m : o[2];
            T f = ((t0 * t0) + (t1 * t1)) + (t2 * t2);
            lb = f < lb ? f : lb;
        }
    }
    return lb;
}

// The same bound without the distance to the data's bounding box (the o[] terms above): a weaker lower bound for queries OUTSIDE the
// box, identical inside. The k = 1 lane pass uses it -- six face distances from values it already holds instead of six more grid
// parameters fetched at the end of every wave; a query outside the box that it fails to certify goes to the wave-per-query pass, which
// uses the full bound.
template <typename T>
__device__ __forceinline__ T face_lower_bound_inner(const GridParams<T>& g, T qx, T qy, T qz, int x0, int x1, int y0, int y1, int z0, int z1) {
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    const T q[3] = {qx, qy, qz};
    const int c0[3] = {x0, y0, z0}, c1[3] = {x1, y1, z1};
    T lb = INFINITY;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        if (c0[j] > 0 || g.closed) {
            T m = q[j] - face_below(g, j, c0[j]);
            m = m > (T)0 ? m * shrink : (T)0;
            const T f = m * m;
            lb = f < lb ? f : lb;
        }
        if (c1[j] < g.G[j] - 1 || g.closed) {
            T m = face_above(g, j, c1[j]) - q[j];
            m = m > (T)0 ? m * shrink : (T)0;
            const T f = m * m;
            lb = f < lb ? f : lb;
        }
    }
    return lb;
}

template <typename T>
__device__ __forceinline__ T dist2(const Pt4<T>& q, const Pt4<T>& r) {
    const T dx = q.x - r.x, dy = q.y - r.y, dz = q.z - r.z;
    return ((dx * dx) + (dy * dy)) + (dz * dz);
}

// Row pruning. The 3x3 rows (oy, oz) around the query's cell are visited centre first; before a row is scanned its
// lower bound is compared with the current k-th best: LB(oy, oz) = ((0) + (my*my)) + (mz*mz), where my / mz is the
// (rounding-safe, see face_lower_bound) distance from the query to the slab of cells with cell_y < ccy (oy = -1) or
// cell_y > ccy (oy = +1), 0 for oy = 0. Every computed d2 of a point in that row is >= LB, so a row with kth < LB
// (strict: no tie can hide there either) is skipped by the lane. Typically 4-5 of the 9 rows survive. The lanes of a
// wave skip different rows, so this saves few instructions -- it saves the *gathered bytes* (a skipped lane issues no
// loads), and the L1 data path (64 B/clk/CU, ~1.7 KB gathered per query) is what bounds the lane-per-query kernels.
__device__ constexpr int kRowOy[9] = {0, -1, 1, 0, 0, -1, 1, -1, 1};
__device__ constexpr int kRowOz[9] = {0, 0, 0, -1, 1, -1, -1, 1, 1};
template <typename T>
__device__ __forceinline__ void row_lower_bounds(const GridParams<T>& g, const Pt4<T>& q, int ccy, int ccz, T (&lb)[9]) {
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    T my[3], mz[3];      // index 0: same row, 1: below, 2: above
    my[0] = (T)0; mz[0] = (T)0;
    { T m = q.y - face_below(g, 1, ccy); my[1] = m > (T)0 ? m * shrink : (T)0; }
    { T m = face_above(g, 1, ccy) - q.y; my[2] = m > (T)0 ? m * shrink : (T)0; }
    { T m = q.z - face_below(g, 2, ccz); mz[1] = m > (T)0 ? m * shrink : (T)0; }
    { T m = face_above(g, 2, ccz) - q.z; mz[2] = m > (T)0 ? m * shrink : (T)0; }
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const T a = my[kRowOy[j] == 0 ? 0 : (kRowOy[j] < 0 ? 1 : 2)], b = mz[kRowOz[j] == 0 ? 0 : (kRowOz[j] < 0 ? 1 : 2)];
        lb[j] = (a * a) + (b * b);
    }
}

// x clamped to [lo, hi] (lo <= hi): the median of the three
__device__ __forceinline__ float clamp3(float lo, float hi, float x) { return __builtin_amdgcn_fmed3f(lo, hi, x); }
__device__ __forceinline__ double clamp3(double lo, double hi, double x) { return x < lo ? lo : (x > hi ? hi : x); }
// Offer one candidate to a lane's K best (ascending d2, registers). Ties are only *detected* here: an equal d2 that
// is rejected at the k-th slot, or an evicted element equal to the new k-th, raises `tie`.
template <typename T, int K>
__device__ __forceinline__ void offer(const T d, const int id, T (&bd)[K], int (&bi)[K], bool& tie) {
    if (K == 1) {                          // branch-free: two compares, two selects
        tie = tie || (d == bd[0]);
        const bool lt = d < bd[0];
        bd[0] = lt ? d : bd[0];
        bi[0] = lt ? id : bi[0];
        return;
    }
    tie = tie || (d == bd[K - 1]);         // rejected (or about to tie with) the k-th
    if (d < bd[K - 1]) {
        const T ev = bd[K - 1];
        // sorted insertion, the largest drops out: new bd[i] = d clamped to [bd[i-1], bd[i]] -- one v_med3 per slot for the distances;
        // the row ids follow with the two compares that slot i shares with slot i-1
        bool gi = true;                     // bd[K-1] > d
#pragma unroll
        for (int i = K - 1; i > 0; --i) {
            // Early exit, four slots at a time: once no lane of the wave has a slot above that still moves (gi: old bd[i] > d), every
            // lower slot keeps its value (the list is sorted). Candidates met late in a scan beat the k-th best only narrowly and land
            // in the top few slots: the K-wide chain (four 4-cycle-class instructions per slot, profiles/r03_valu_rate.txt) shrinks
            // to its first chunk for most insertions.
            if (K >= 8 && (i & 3) == 3 && i != K - 1) { if (!__any(gi)) break; }
            const bool gm = bd[i - 1] > d;
            bd[i] = clamp3(bd[i - 1], bd[i], d);
            bi[i] = gm ? bi[i - 1] : (gi ? id : bi[i]);
            gi = gm;
        }
        bi[0] = gi ? id : bi[0];
        bd[0] = gi ? d : bd[0];
        if (ev == bd[K - 1] && ev != Limits<T>::max_v) tie = true;      // evicted one equals the new k-th
    }
}

// Candidates are consumed four at a time; slots past the end of the range get all exponent bits set (+inf or NaN),
// which compares false with both `<` and `==`, so they can never be selected or flagged. Done with integer OR so
// that the group stays straight-line code (a `?:` on the distance makes the compiler branch around the loads).
__device__ __forceinline__ float kill_if(float d, bool dead) { return __uint_as_float(__float_as_uint(d) | (dead ? 0x7f800000u : 0u)); }
__device__ __forceinline__ double kill_if(double d, bool dead) {
    return __longlong_as_double(__double_as_longlong(d) | (dead ? 0x7ff0000000000000ll : 0ll));
}
// Certification, output and list appends of one lane (shared by the gather and the LDS-tile main passes).
// `valid` is false for padding lanes of a partial wave (they only take part in the wave-wide list appends).
// POS: the ids in bi[] are record positions in the dataset's cell order (k_search_runs reads the coordinates-only stream, which has no row ids);
// the rows are looked up here, once, for the kreq best.
template <typename T, int K, bool POS = false>
__device__ __forceinline__ void finish_lane(const SearchArgs<T>& a, const GridParams<T>& g, const Pt4<T>& q, int qpos,
                                            int x0, int x1, int y0, int y1, int z0, int z1, T (&bd)[K], int (&bi)[K], bool tie, bool valid,
                                            bool defer = false) {
    if (defer) {                 // nothing was scanned: the wave-per-query pass at the same radius takes over
        wave_append(false, qpos, a.unresolved, a.n_unresolved);
        wave_append(valid, qpos, a.ties, a.n_ties);
        return;
    }
    const T lb = face_lower_bound_inner(g, q.x, q.y, q.z, x0, x1, y0, y1, z0, z1);
    const int kreq = a.kreq;
    T kth = bd[0];
#pragma unroll
    for (int i = 1; i < K; ++i) if (i == kreq - 1) kth = bd[i];
    const bool certified = valid && kth < lb;

    if (certified) {
        const size_t o = (size_t)(a.row_out ? (int)q.idx : qpos) * (size_t)kreq;       // cell order (coalesced rows; see k_unpermute) unless row_out
#pragma unroll
        for (int i = 0; i < K; ++i) {
            if (i < kreq) {
                const bool found = bi[i] != 0x7fffffff;
                a.out_i[o + i] = found ? (long long)(POS ? a.ref_idx[bi[i]] : bi[i]) : -1ll;
                a.out_d[o + i] = found ? (a.squared ? bd[i] : sqrt(bd[i])) : (T)-1;
            }
        }
        // equal neighbours inside the first kreq(+1) slots
        bool adj = false;
#pragma unroll
        for (int i = 1; i < K; ++i)
            if (i <= kreq && bd[i] == bd[i - 1] && bi[i] != 0x7fffffff) adj = true;
        tie = tie || adj;
    }
    const int us = wave_append(valid && !certified, qpos, a.unresolved, a.n_unresolved);
    if (us >= 0 && a.ubound) a.ubound[us] = kth;
    wave_append(certified && tie, qpos, a.ties, a.n_ties);
}

// Main pass: radius R = 1 (the 27 cells around the query's cell). The 9 row bounds are fetched up front (18
// independent loads in flight), then each row is consumed in groups of 4 candidates whose 4 loads are issued
// together, so a lane exposes ~20 dependent memory latencies instead of ~70.
#ifndef PCU_KBUF
#define PCU_KBUF 12
#endif
#ifndef PCU_KSEARCH_PIPE
#define PCU_KSEARCH_PIPE 1
#endif

// Measured on config 3 (k = 16, 4M-vs-4M; profiles/r03_c3_*): 14.6k VALU + 4.9k SALU + 384 vector-memory instructions per wave at 44 % active
// lanes, 49 % of the wave-cycles waiting on memory at 4 waves per SIMD (125 VGPRs); 40 KB of code (the burst insertion is inlined at every
// row). Tried in round 3 and not kept: forcing 5 / 6 waves per SIMD (96 VGPRs: equal; 80: spills, 1.45x slower), a larger parking buffer
// (equal), and a collection radius that cuts the insertions while a lane's slots fill (from ~67 to ~27 per lane: kernel time unchanged,
// 1.7x the stragglers) -- the K-wide insertion is not what bounds it.
template <typename T, int K>
__global__ __launch_bounds__(kBlock) void k_search(const SearchArgs<T> a) {
    // XCD-aware block order: workgroup b is dispatched to XCD b % 8 (observed placement; speed only, never
    // correctness). Queries are in spatial (cell) order, so giving each XCD one CONTIGUOUS eighth of the blocks
    // makes every XCD's private L2 hold one eighth of the dataset (+halo) instead of all of it.
    // (the launcher rounds the grid up to a multiple of 8, so the map below is a bijection of the block ids)
    const int per = (int)(gridDim.x >> 3);
    const int vb = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);
    const int t = vb * kBlock + threadIdx.x;
    const int nq = a.qcount_dev ? *a.qcount_dev : a.nq;
    const GridParams<T>& g = *a.gp;
    if (!g.closed && t >= nq) return;              // (closed levels: the whole block stays for block_append below)
    const bool active = t < nq;
    const int qpos = active ? (a.qlist ? a.qlist[t] : t) : 0;
    const Pt4<T> q = a.qsorted[qpos];
    if (const int hl = index_not_ready(a, g)) { if (t == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (t == 0) *a.skew_flag = (float)g.sumsq > a.skew_far ? 2 : 1; return; }
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    // A closed sub-box level (dense part of an unbalanced cloud) can never certify a query outside its box [org, org + G h) --
    // face_lower_bound is then at most the query's distance to the level's points, which no candidate beats -- so such a query goes to
    // the next level without a scan (most queries of the call, when the box is a small cluster).
    if (g.closed) {
        const T tx = (q.x - g.org[0]) * g.inv_h, ty = (q.y - g.org[1]) * g.inv_h, tz = (q.z - g.org[2]) * g.inv_h;
        const bool outside = !(tx >= (T)0 && tx < (T)Gx && ty >= (T)0 && ty < (T)Gy && tz >= (T)0 && tz < (T)Gz);      // (skipping a level is always safe)
        block_append(active && outside, qpos, a.unresolved, a.n_unresolved);
        if (!active || outside) return;
    }

    const int ccx = grid_cell(g, 0, q.x), ccy = grid_cell(g, 1, q.y), ccz = grid_cell(g, 2, q.z);
    const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, Gx - 1);
    const int y0 = max(ccy - 1, 0), y1 = min(ccy + 1, Gy - 1);
    const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, Gz - 1);

    T bd[K];
    int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = Limits<T>::max_v; bi[i] = 0x7fffffff; }
    bool tie = false;

    unsigned rs[9], re[9];                 // rows centre-out: near rows first so the k-th best shrinks early
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        const int cy = ccy + kRowOy[j], cz = ccz + kRowOz[j];
        const bool ok = cy >= 0 && cy < Gy && cz >= 0 && cz < Gz;
        const int lo = row_run_lo(Gx, grid_row(Gy, ok ? cy : ccy, ok ? cz : ccz), x0, x1);
        rs[j] = a.cell_start[lo];
        const unsigned e = a.cell_start[lo + (x1 - x0 + 1)];
        re[j] = ok ? e : rs[j];
    }
    // Candidates are consumed kGroup at a time. Slots past the end of a row read the sentinel record sorted[n]
    // (all coordinates +inf -> d2 = +inf, never < nor == anything), so a group is straight-line code with no
    // per-slot masking, and record addresses are 32-bit byte offsets from a uniform base (one shift per load).
    constexpr int kGroup = 4;
    const char* const base = reinterpret_cast<const char*>(a.ref);
    const unsigned sentinel = a.n_ref;
    unsigned total = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) total += re[j] - rs[j];
    const bool defer = total > a.lane_max_cand;
    T rlb[9];
    row_lower_bounds(g, q, ccy, ccz, rlb);
    // K > 1: accepted candidates are first parked in a small per-lane buffer in LDS and inserted in bursts. The sorted
    // insertion is K-wide predicated code that the whole wave executes whenever ANY lane accepts a candidate -- early in a
    // scan that is every candidate (k = 16: 64 of the 78 VALU instructions per candidate step). Parked candidates go through
    // the very same offer() later, against a k-th best that can only have become smaller, so results and tie flags are
    // unchanged; a wave now runs the insertion max-over-lanes-of-the-buffer-fill times per burst instead of once per step.
    constexpr int kBuf = K > 1 ? PCU_KBUF : 1;
    __shared__ T s_bd[kBuf][kBlock];
    __shared__ int s_bi[kBuf][kBlock];
    const int tid = threadIdx.x;
    int cnt = 0;
    auto flush = [&]() {
        int mx = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        for (int i = 0; i < mx; ++i)
            if (i < cnt) offer<T, K>(s_bd[i][tid], s_bi[i][tid], bd, bi, tie);
        cnt = 0;
    };
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        if (K > 1 && __any(cnt >= kBuf / 2)) flush();        // (also gives the row pruning below a fresher k-th best)
        const unsigned e = (defer || bd[K - 1] < rlb[j]) ? rs[j] : re[j];
        // The groups of a row are software-pipelined: the next group's four loads are requested before the current group is evaluated
        // (straight-line, a lane past its row's end re-reads the sentinel), so a wait on memory covers the evaluation and the parking of
        // a whole group. Before (one group requested and waited for per trip, PCU_KSEARCH_PIPE=0) k_search<float,16> spent 49 % of its
        // wave-cycles waiting on memory at 4 waves per SIMD (profiles/r03_c3_*.txt).
        auto load_group = [&](unsigned p, Pt4<T> (&c)[kGroup]) {
#pragma unroll
            for (int u = 0; u < kGroup; ++u) {
                const unsigned idx = (p + u < e) ? p + u : sentinel;
                c[u] = *reinterpret_cast<const Pt4<T>*>(base + (size_t)(idx * (unsigned)sizeof(Pt4<T>)));
            }
        };
        auto eval_group = [&](const Pt4<T> (&c)[kGroup]) {
            if (K == 1) {
#pragma unroll
                for (int u = 0; u < kGroup; ++u) offer<T, K>(dist2(q, c[u]), (int)c[u].idx, bd, bi, tie);
            } else {
#pragma unroll
                for (int u = 0; u < kGroup; ++u) {
                    const T d = dist2(q, c[u]);
                    tie = tie || (d == bd[K - 1]);            // as offer() would flag it (the k-th best may be stale: conservative)
                    if (d < bd[K - 1]) { s_bd[cnt][tid] = d; s_bi[cnt][tid] = (int)c[u].idx; ++cnt; }
                }
                if (__any(cnt > kBuf - kGroup)) flush();
            }
        };
        if (PCU_KSEARCH_PIPE && K > 1) {
            unsigned p = rs[j];
            if (p < e) {
                Pt4<T> ca[kGroup], cb[kGroup];
                load_group(p, ca);
                for (;;) {
                    p += kGroup;
                    load_group(p < e ? p : sentinel, cb);      // (past the end: four sentinel records, never evaluated)
                    eval_group(ca);
                    if (!(p < e)) break;
                    p += kGroup;
                    load_group(p < e ? p : sentinel, ca);
                    eval_group(cb);
                    if (!(p < e)) break;
                }
            }
        } else {
            for (unsigned p = rs[j]; p < e; p += kGroup) {
                Pt4<T> c[kGroup];
                load_group(p, c);
                eval_group(c);
            }
        }
    }
    if (K > 1) flush();

    finish_lane<T, K>(a, g, q, qpos, x0, x1, y0, y1, z0, z1, bd, bi, tie, true, defer);
}

// -------------------------------------------------------------------------------------------------------
// Main pass for k = 1 (Chamfer / Hausdorff / k_nearest_neighbors(k=1)): k_search1_flat, the same lane-per-query scan as
// k_search<T,1> re-built around what bounded it (profiles/r01_pmc.txt: 2,365 VALU instructions per wave, VALU pipes 96 %
// busy, 108 gathered candidate slots per lane):
//   * a row is consumed in groups of 4 records from its first record on; the last group may run up to 3 records past
//     the row's end. Those are real dataset points of the cells that follow in snake order (or the +inf sentinels behind
//     the last record), so offering them is harmless and no slot needs masking or a select on its address. (At a grid
//     border the records that follow can belong to another row of the same 27 cells: a winner seen twice is recognised,
//     see the end of the kernel.)
//   * the (x,y) differences, squares of a record go through the packed-fp32 pipe (v_pk_add_f32 / v_pk_mul_f32: IEEE
//     add and mul, no FMA -- bit-identical to the scalar sequence);
//   * only the running minimum d2 and the *group* it came from are tracked (v_min3 + one compare + two selects per 4
//     candidates instead of a compare and two selects per candidate); the winning record is identified afterwards by
//     re-evaluating that one group. An equal minimum met in another group, or twice inside the winning group, flags a
//     possible tie (re-resolved by the wave-per-query pass under the total order);
//   * rows and the outer cells of a row's run are pruned per lane against the running minimum, with lower bounds computed
//     in the same rounding-monotone arithmetic as the certification (strict '<': no tie can hide in what is skipped):
//       rows   LB = (my*my) + (mz*mz)                 my / mz: distance to the slab of the row (0: own row)
//       cells  LB = ((mx*mx) + (my*my)) + (mz*mz)     for the cells ccx-1 / ccx+1 of a run; the cut run stays contiguous
//     A skipped lane issues no loads: this saves gathered bytes (the L1/texture path is the co-bottleneck);
//   * after the centre row has given a first estimate, the surviving cut runs of the other eight rows are written to a
//     per-lane list in LDS ({byte offset, record count, row bound as a round-down bf16}: 8 bytes each) and consumed by ONE
//     loop per lane, with the next group's four loads issued before the current group is evaluated. The lanes of a wave
//     walk their own lists in lock step (a wave runs max-over-lanes of the TOTAL group count instead of the sum over rows
//     of per-row maxima) and every wait on memory covers two groups. Row pruning stays adaptive (the bound is re-checked,
//     against a minimum that may be one group stale -- still a valid bound -- when a lane moves to its next run); the cell
//     cuts are those decided after the centre row.
// Only for open indexes (closed sub-box levels have no sentinel behind their last record).
// The query's cell along one axis, as cell_coord (pcu_types.h) gives it: for float the clamp is one v_med3 before the conversion instead of two
// compare + select pairs behind it (t in [0, G - 1] converts to itself truncated, everything above to G - 1, everything below -- and NaN -- to 0).
__device__ __forceinline__ int cell_of_query(const GridParams<float>& g, const int axis, const float v) {
    const float t = (v - g.org[axis]) * g.inv_h;
    return (int)__builtin_amdgcn_fmed3f(t, 0.f, (float)(g.G[axis] - 1));
}
__device__ __forceinline__ int cell_of_query(const GridParams<double>& g, const int axis, const double v) { return grid_cell(g, axis, v); }
template <typename T> struct K1Group { static constexpr int n = 4; };     // records per group (8 measured slower: more bytes gathered past the row ends)
struct __attribute__((packed, aligned(4))) CellStart4 { unsigned v[4]; };
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ float dist2_k1(const Pt4<float>& q, const Pt4<float>& c) {
    const f32x2 qxy = {q.x, q.y}, cxy = {c.x, c.y};
    const f32x2 d = qxy - cxy;
    const f32x2 dd = d * d;
    const float dz = q.z - c.z;
    return (dd.x + dd.y) + (dz * dz);
}
__device__ __forceinline__ double dist2_k1(const Pt4<double>& q, const Pt4<double>& c) { return dist2(q, c); }
__device__ __forceinline__ float min4(float a, float b, float c, float d) { return __builtin_fminf(__builtin_fminf(__builtin_fminf(a, b), c), d); }
__device__ __forceinline__ double min4(double a, double b, double c, double d) { return __builtin_fmin(__builtin_fmin(a, b), __builtin_fmin(c, d)); }

__device__ __forceinline__ unsigned lb_pack(float lb) { return __float_as_uint(lb) >> 16; }                  // truncation rounds a value >= 0 down
__device__ __forceinline__ unsigned lb_pack(double lb) { const double v = lb * (1.0 - 1e-6); return __float_as_uint((float)(v < 1e38 ? v : 1e38)) >> 16; }
template <typename T> __device__ __forceinline__ T lb_unpack(unsigned b) { return (T)__uint_as_float(b << 16); }

// Variants of the k = 1 main pass (compile-time, A/B-measured on the GPU; profiles/r03_*):
//   PCU_FLAT_XYZ   candidates are read from a coordinates-only copy of the cell-ordered cloud (GridIndex::xyz: 3 T per record, no
//                  row id): a group of 4 records is 12 consecutive scalars = THREE 16-byte loads instead of four, and the six
//                  (x,y) / (z,x) / (y,z) pairs of the group go through the packed-fp32 pipe. A per-lane gather instruction costs
//                  the CU's L1 / texture-address path ~17-21 cycles whatever its width (profiles/r02_ubench.txt), and that path is
//                  the kernel's tightest resource, so a quarter fewer instructions in the loops is a quarter less of it. The row id
//                  of the winner is fetched once, at the end, from the Pt4 records (not at all by the fused Chamfer sum).
// Measured and rejected in round 3 (profiles/r03_flat_ab.txt): re-dealing the block's 256 queries to its lanes by remaining work after
// the centre row (LDS counting sort + hand-over of the query state, wave 0 = the 64 heaviest ...): fewer loop trips, but four block
// barriers in a latency-bound kernel, 80 instead of 72 VGPRs and 25 KB of LDS: 85.9 vs 83.9 us alone, 81.0 vs 78.0 us on top of XYZ.
#ifndef PCU_FLAT_XYZ
#define PCU_FLAT_XYZ 1
#endif
constexpr bool kFlatXyz = PCU_FLAT_XYZ != 0;

template <typename T> struct __attribute__((packed, aligned(4))) Group12 { T v[12]; };      // 4 records of a coordinates-only stream
// record index -> byte offset in the candidate stream: shifts and one add (v_mul_lo_u32 is a quarter-rate instruction, and a 24-bit multiply
// does not reach the 2^27 - 16 records an index may hold)
template <unsigned REC> __device__ __forceinline__ unsigned rec_bytes(unsigned i) {
    static_assert(REC == 12u || REC == 24u || REC == 16u || REC == 32u, "record sizes of the candidate streams");
    if (REC == 16u) return i << 4;
    if (REC == 32u) return i << 5;
    unsigned r; const unsigned t = REC == 12u ? i << 2 : i << 3;      // (as inline assembly: the compiler folds the C expression back into the multiply)
    if (REC == 12u) asm("v_lshl_add_u32 %0, %1, 3, %2" : "=v"(r) : "v"(i), "v"(t));
    else asm("v_lshl_add_u32 %0, %1, 4, %2" : "=v"(r) : "v"(i), "v"(t));
    return r;
}
// a * b + c with 24-bit unsigned factors, one full-rate instruction (the compiler widens the C expression to a 64-bit multiply-add when it cannot prove the ranges)
__device__ __forceinline__ unsigned mad24(unsigned a, unsigned b, unsigned c) { unsigned r; asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c)); return r; }

// One group = 4 consecutive records of the candidate stream: `load` requests it (straight-line loads), `dists` gives the 4 squared
// distances (bit-identical to 4 x dist2: IEEE subtract, multiply, add in the reference's order).
template <typename T, bool XYZ> struct GroupEval;
template <typename T> struct GroupEval<T, false> {           // Pt4 records
    static constexpr unsigned kRec = (unsigned)sizeof(Pt4<T>);
    struct Raw { Pt4<T> c[4]; };
    static __device__ __forceinline__ Raw load(const char* base, unsigned off) {
        const Pt4<T>* c = reinterpret_cast<const Pt4<T>*>(base + (size_t)off);
        Raw r; r.c[0] = c[0]; r.c[1] = c[1]; r.c[2] = c[2]; r.c[3] = c[3]; return r;
    }
    static __device__ __forceinline__ void dists(const Raw& r, const Pt4<T>& q, T (&d)[4]) {
        d[0] = dist2_k1(q, r.c[0]); d[1] = dist2_k1(q, r.c[1]); d[2] = dist2_k1(q, r.c[2]); d[3] = dist2_k1(q, r.c[3]);
    }
};
template <> struct GroupEval<float, true> {
    static constexpr unsigned kRec = 12u;
    typedef Group12<float> Raw;
    static __device__ __forceinline__ Raw load(const char* base, unsigned off) { return *reinterpret_cast<const Raw*>(base + (size_t)off); }
    static __device__ __forceinline__ void dists(const Raw& g, const Pt4<float>& q, float (&d)[4]) {
        const f32x2 qxy = {q.x, q.y}, qzx = {q.z, q.x}, qyz = {q.y, q.z};
        f32x2 p0 = qxy - f32x2{g.v[0], g.v[1]}, p1 = qzx - f32x2{g.v[2], g.v[3]}, p2 = qyz - f32x2{g.v[4], g.v[5]};
        f32x2 p3 = qxy - f32x2{g.v[6], g.v[7]}, p4 = qzx - f32x2{g.v[8], g.v[9]}, p5 = qyz - f32x2{g.v[10], g.v[11]};
        p0 = p0 * p0; p1 = p1 * p1; p2 = p2 * p2; p3 = p3 * p3; p4 = p4 * p4; p5 = p5 * p5;
        d[0] = (p0.x + p0.y) + p1.x; d[1] = (p1.y + p2.x) + p2.y; d[2] = (p3.x + p3.y) + p4.x; d[3] = (p4.y + p5.x) + p5.y;
    }
};
template <> struct GroupEval<double, true> {
    static constexpr unsigned kRec = 24u;
    typedef Group12<double> Raw;
    static __device__ __forceinline__ Raw load(const char* base, unsigned off) { return *reinterpret_cast<const Raw*>(base + (size_t)off); }
    static __device__ __forceinline__ void dists(const Raw& g, const Pt4<double>& q, double (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const double dx = q.x - g.v[3 * u], dy = q.y - g.v[3 * u + 1], dz = q.z - g.v[3 * u + 2]; d[u] = ((dx * dx) + (dy * dy)) + (dz * dz); }
    }
};
// A value's bit pattern as an unsigned integer: for d2 >= +0 the patterns order like the values (LDS atomic min on distances, see "adoption")
template <typename T> struct BitsOf;
template <> struct BitsOf<float> { typedef unsigned type; static __device__ __forceinline__ unsigned of(float v) { return __float_as_uint(v); } static __device__ __forceinline__ float back(unsigned b) { return __uint_as_float(b); } };
template <> struct BitsOf<double> { typedef unsigned long long type; static __device__ __forceinline__ unsigned long long of(double v) { return (unsigned long long)__double_as_longlong(v); } static __device__ __forceinline__ double back(unsigned long long b) { return __longlong_as_double((long long)b); } };
// Register layout: the centre row's table is loaded and scanned first; only then are the other eight rows' tables fetched
// (EARLY = false: one more dependent wait per wave, but their 32 registers are not live during the centre scan: 68 VGPRs
// instead of 93, 7 waves per SIMD instead of 5). EARLY = true fetches them right away (79 VGPRs; measured equal).
template <typename T, bool EARLY, int FUSE>
__device__ __forceinline__ void search1_flat_body(const SearchArgs<T>& a, const int nq_arg, const int bid, const int nblk, bool& f_ok, T& f_v, long long& f_key) {
    constexpr bool XYZ = kFlatXyz;
    typedef GroupEval<T, XYZ> GE;
    __shared__ uint2 s_rng[8][kBlock];
    const int per = nblk >> 3;
    const int vb = (bid & 7) * per + (bid >> 3);       // XCD-aware block order, see k_search (nblk and the side's first block: multiples of 8)
    const int tid = threadIdx.x;
    const int nq = a.qcount_dev ? *a.qcount_dev : nq_arg;
    if (vb * kBlock >= nq) return;                     // (block-uniform)
    const GridParams<T>& g = *a.gp;
    if (const int hl = index_not_ready(a, g)) { if (vb == 0 && tid == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (vb == 0 && tid == 0) *a.skew_flag = (float)g.sumsq > a.skew_far ? 2 : 1; return; }
    const int t = vb * kBlock + tid;
    if (t >= nq) return;
    constexpr bool valid = true;
    const int qpos = a.qlist ? a.qlist[t] : t;
    Pt4<T> q;                                         // (from the coordinate stream; the row id only where an epilogue uses it)
    {
        struct __attribute__((packed, aligned(4))) Q3 { T v[3]; };
        const Q3 c = *reinterpret_cast<const Q3*>(a.q_xyz + 3 * (size_t)qpos);
        q.x = c.v[0]; q.y = c.v[1]; q.z = c.v[2];
        q.idx = FUSE == FUSE_SUM ? 0 : a.q_idx[qpos];
    }
    // (FUSE_MAXVAL, round 6: Hausdorff's lane pass on the fused sum's value-only program -- no winner, no tie flags, adoption -- whose partial names
    // the arg-max QUERY; k_fuse_tail resolves that one query's neighbour, reduce.h: FuseTail::maxval)
    constexpr bool VALUE_ONLY = FUSE == FUSE_SUM || FUSE == FUSE_MAXVAL;
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = cell_of_query(g, 0, q.x), ccy = cell_of_query(g, 1, q.y), ccz = cell_of_query(g, 2, q.z);
    constexpr unsigned kRec = GE::kRec;               // bytes per record of the candidate stream
    constexpr int kG = K1Group<T>::n;
    const char* const base = XYZ ? reinterpret_cast<const char*>(a.ref_xyz) : reinterpret_cast<const char*>(a.ref);
    const unsigned cand_cap = a.lane_max_cand < 65535u ? a.lane_max_cand : 65535u;     // a run's record count is packed into 16 bits
    T best = Limits<T>::max_v;
    unsigned boff = 0xffffffffu, toff = 0xffffffffu;
    bool tie = false, tie2 = false;
#define PCU_K1_EVAL(RAW, OFF)                                                                                \
    {                                                                                                        \
        T d_[4]; GE::dists((RAW), q, d_);                                                                    \
        const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);                                                       \
        const bool eq_ = m_ == best, lt_ = m_ < best;                                                        \
        tie2 = !lt_ && (tie2 || (tie && eq_));                                                               \
        tie = !lt_ && (tie || eq_);                                                                          \
        toff = eq_ ? (OFF) : toff;                                                                           \
        best = lt_ ? m_ : best;                                                                              \
        boff = lt_ ? (OFF) : boff;                                                                           \
    }
    // ---- the nine row tables. Round 5: the per-query part of this kernel was 680 of its 1180 vector instructions per wave (428 of them
    // here: nine table addresses with their clamps, snake-order selects and run-length cases, and the cut decisions made on the *results*
    // of per-row comparisons), the candidate loops ~500. What every row shares is now computed once per query:
    //   * the x direction of a row is (cy ^ cz) & 1 (pcu_types.h: grid_row / row_run_lo; the parity of the linear row number whatever Gy is),
    //     so a row either runs like the centre row (|oy| + |oz| even) or against it: two table offsets, and the two sides of the query's cell
    //     (A = the side met first in a row that runs like the centre row, B = the other) are swapped on the *inputs* of the cut tests;
    //   * the linear numbers of the nine rows are three bases and a per-query step of +-1 (the y order flips with cz);
    //   * a row's table is ALWAYS the four words {start of the cell before the query's, start of its own, of the next one, end of that}: at a
    //     grid border the missing cell's word belongs to the neighbouring row (or, before the first row, to the padding behind GridParams:
    //     index_alloc) and is never used -- a missing cell counts as cut -- so there are no run-length cases and no clamped cell range.
    // The arithmetic of the bounds (and with it what is scanned and what is certified) is unchanged.
    const bool hasxl = ccx > 0, hasxh = ccx < Gx - 1;
    const bool odd0 = ((ccy ^ ccz) & 1) != 0, zodd = (ccz & 1) != 0;
    const bool hasA = (odd0 && hasxh) || (!odd0 && hasxl), hasB = (odd0 && hasxl) || (!odd0 && hasxh);      // (mask logic: scalar unit)
    const unsigned Gx4 = (unsigned)Gx << 2;           // byte strides of the table (cells per row <= 2048, rows < 2^22: 24-bit multiplies)
    const unsigned xe4 = (unsigned)ccx << 2, xo4 = (unsigned)(Gx - 1 - ccx) << 2;     // (+1 cell: the base below is one word before the table)
    const unsigned xS4 = odd0 ? xo4 : xe4, xR4 = odd0 ? xe4 : xo4;                    // rows that run like the centre row / against it
    const int yS = zodd ? Gy - 1 - ccy : ccy, sS = zodd ? -1 : 1;                     // y position inside the query's z slab, and the slab's y step
    const int row0 = (int)mad24((unsigned)ccz, (unsigned)Gy, (unsigned)yS);
    const int flip = (Gy - 1 - yS) - yS;                                              // the neighbouring slabs count y the other way round
    const int rowM = row0 + flip - Gy, rowP = row0 + flip + Gy;
    const bool okyM = ccy > 0, okyP = ccy < Gy - 1, okzM = ccz > 0, okzP = ccz < Gz - 1;
    const char* const tbase = reinterpret_cast<const char*>(a.cell_start) - 4;
    auto row_table = [&](int j, bool& ok) {
        const int oy = kRowOy[j], oz = kRowOz[j];
        ok = (oy == 0 || (oy < 0 ? okyM : okyP)) && (oz == 0 || (oz < 0 ? okzM : okzP));
        const int rr = oz == 0 ? row0 + oy * sS : (oz < 0 ? rowM : rowP) - oy * sS;
        const unsigned row = (unsigned)(ok ? rr : row0);                              // (a row outside the grid: any valid address, the row is dropped below)
        return *reinterpret_cast<const CellStart4*>(tbase + (size_t)(__umul24(row, Gx4) + (((oy + oz) & 1) == 0 ? xS4 : xR4)));
    };
    bool okj[9];
    CellStart4 tb[9];
    tb[0] = row_table(0, okj[0]);
    if (EARLY) {
#pragma unroll
        for (int j = 1; j < 9; ++j) tb[j] = row_table(j, okj[j]);
    }
    // ---- centre row: whole run
    const unsigned c_s = hasA ? tb[0].v[0] : tb[0].v[1], c_e = hasB ? tb[0].v[3] : tb[0].v[2];
    const unsigned cnt0 = c_e - c_s;
    bool defer = cnt0 > cand_cap;
    {
        const unsigned o0 = rec_bytes<kRec>(c_s);
        const unsigned o1 = (defer || !valid) ? o0 : rec_bytes<kRec>(c_e);
        for (unsigned off = o0; off < o1; off += (unsigned)kG * kRec) { const typename GE::Raw raw = GE::load(base, off); PCU_K1_EVAL(raw, off) }
    }
    // ---- the other rows: cut runs that survive the centre row's minimum -> this lane's list
    if (!EARLY) {
#pragma unroll
        for (int j = 1; j < 9; ++j) tb[j] = row_table(j, okj[j]);
    }
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    T mxl = q.x - face_below(g, 0, ccx); mxl = mxl > (T)0 ? mxl * shrink : (T)0;
    T mxh = face_above(g, 0, ccx) - q.x; mxh = mxh > (T)0 ? mxh * shrink : (T)0;
    const T mxl2 = mxl * mxl, mxh2 = mxh * mxh;
    const T mA2 = odd0 ? mxh2 : mxl2, mB2 = odd0 ? mxl2 : mxh2;
    T my2[3], mz2[3];
    {
        T m;
        my2[0] = (T)0; mz2[0] = (T)0;
        m = q.y - face_below(g, 1, ccy); m = m > (T)0 ? m * shrink : (T)0; my2[1] = m * m;
        m = face_above(g, 1, ccy) - q.y; m = m > (T)0 ? m * shrink : (T)0; my2[2] = m * m;
        m = q.z - face_below(g, 2, ccz); m = m > (T)0 ? m * shrink : (T)0; mz2[1] = m * m;
        m = face_above(g, 2, ccz) - q.z; m = m > (T)0 ? m * shrink : (T)0; mz2[2] = m * m;
    }
    // (the cap on a lane's serial work counts what the lane would scan: the centre row and the cut runs it lists)
    unsigned total = cnt0;
    int n = 0;
#pragma unroll
    for (int j = 1; j < 9; ++j) {
        const int oy = kRowOy[j], oz = kRowOz[j];
        const T ry = my2[oy == 0 ? 0 : (oy < 0 ? 1 : 2)], rz = mz2[oz == 0 ? 0 : (oz < 0 ? 1 : 2)];
        const bool same = ((oy + oz) & 1) == 0;
        const T mF2 = same ? mA2 : mB2, mL2 = same ? mB2 : mA2;      // the squared x margins of the run's first / last cell
        const bool hasF = same ? hasA : hasB, hasL = same ? hasB : hasA;
        // lower bounds ((mx*mx) + (my*my)) + (mz*mz) in the distance's own operation order; a zero term is left out (x + 0 == x for x >= +0)
        const T rlb = oy == 0 ? rz : (oz == 0 ? ry : ry + rz);
        const T bF = oy == 0 ? mF2 + rz : (oz == 0 ? mF2 + ry : (mF2 + ry) + rz), bL = oy == 0 ? mL2 + rz : (oz == 0 ? mL2 + ry : (mL2 + ry) + rz);
        const bool cutF = !hasF || best < bF, cutL = !hasL || best < bL;
        const unsigned s_run = cutF ? tb[j].v[1] : tb[j].v[0], e_run = cutL ? tb[j].v[2] : tb[j].v[3];
        const bool take = okj[j] && !defer && !(best < rlb) && e_run > s_run;
        s_rng[n][tid] = make_uint2(rec_bytes<kRec>(s_run), ((e_run - s_run) << 16) | lb_pack(rlb));       // (slot n is overwritten until a run is taken)
        n += take ? 1 : 0;
        total += take ? e_run - s_run : 0u;
    }
    if (total > cand_cap) { defer = true; n = 0; }
    const int own = tid;
    int r = 0;
    unsigned off = 0, end = 0;
    bool live = false;
    auto next_run = [&]() {
        live = false;
        while (r < n) {
            const uint2 e = s_rng[r][own];
            ++r;
            if (!(best < lb_unpack<T>(e.y & 0xffffu))) { off = e.x; end = e.x + __umul24(e.y >> 16, kRec); live = true; break; }
        }
    };
    next_run();
    // Ping-pong: A is evaluated while B's loads are in flight, and vice versa. The loads are unconditional straight-line code (a lane
    // that has just run out of work fetches the +inf sentinel records once): loads issued under a branch would make the compiler wait
    // for them at the join, i.e. before the older group is evaluated.
    const unsigned sent_off = a.n_ref * kRec;
#ifndef PCU_ADOPT
#define PCU_ADOPT 1
#endif
#ifndef PCU_ADOPT_T
#define PCU_ADOPT_T 3
#endif
    // ---- adoption (round 5, fused sum only). The loop below runs until the wave's slowest lane is done: ~10 groups for a mean need of 3.5
    // (profiles/r05_flat_ab.txt), and every trip costs the CU's texture-address path three gather instructions whatever the number of lanes
    // still in it -- the resource that bounds this kernel. So every lane first evaluates at most TWO groups of its own list (which keeps the
    // adaptive row pruning where it pays: a lane's first groups tighten its minimum most); then the groups the wave's lanes still owe
    // -- runs not yet overtaken by their lane's minimum, ~130 per wave -- are listed in LDS (slots by a wave prefix sum) and evaluated by all 64
    // lanes in lock step, item w by lane w mod 64, against the query of the lane that listed it (coordinates through the LDS crossbar), a
    // group's minimum merged into its owner's best by an LDS atomic min on the value's bits (d2 >= +0: the bit patterns order like the values;
    // +inf sentinels and NaN sort behind every finite value, as with '<'). ~3 lock-step trips instead of ~8 more max-over-lanes trips; the same
    // candidates or a few more (what the owner's shrinking minimum would have skipped), all of them dataset points of the box -- the minimum is
    // the same. Only the fused sum needs no more than the minimum (no winner's row, no tie flags); a wave whose list would not fit, or with
    // exited lanes, goes on lane by lane.
    if (PCU_ADOPT && VALUE_ONLY) {
        // (list capacity: with the run list, the block's fold and 8 blocks per CU -- the occupancy the kernel is tuned for -- 160 four-byte items
        // per wave are what fits the 160 KB of LDS; an item = record index of the group (26 bits: clouds of 2^26 records or more go on lane by
        // lane) | owner lane << 26. After three own groups a wave owes ~90 groups on a uniform cloud, 140 at most in the replay.)
        constexpr int kAdoptCap = 160;
        constexpr int kOwn = PCU_ADOPT_T;
        typedef typename BitsOf<T>::type Bits;
        __shared__ unsigned s_aitem[kBlock / 64][kAdoptCap];
        __shared__ Bits s_abest[kBlock / 64][64];
        const int lane_ = tid & 63, wave_ = tid >> 6;
        constexpr unsigned kStep = (unsigned)kG * kRec;
        if (live) {                                      // the lane's own first groups, the next one requested before the current one is evaluated
            typename GE::Raw g0 = GE::load(base, off);
#pragma unroll
            for (int it = 0; it < kOwn; ++it) {
                const unsigned coff = off;
                off += kStep;
                if (off >= end) next_run();
                typename GE::Raw g1;
                if (it + 1 < kOwn) g1 = GE::load(base, live ? off : sent_off);
                PCU_K1_EVAL(g0, coff)
                if (!live) break;
                if (it + 1 < kOwn) g0 = g1;
            }
        }
        if (__ballot(true) == ~0ull && a.n_ref < (1u << 26)) {
            // groups this lane still owes (runs its minimum has overtaken in the meantime are dropped here as the loop would drop them)
            unsigned c = 0;
            if (live) {
                c = (end - off + kStep - 1u) / kStep;
                for (int rr = r; rr < n; ++rr) { const uint2 e = s_rng[rr][own]; if (!(best < lb_unpack<T>(e.y & 0xffffu))) c += ((e.y >> 16) + (unsigned)kG - 1u) / (unsigned)kG; }
            }
            unsigned inc = c;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane_ >= o) inc += u; }
            const unsigned W = (unsigned)__shfl((int)inc, 63, 64);
            if (W != 0u && W <= (unsigned)kAdoptCap) {
                unsigned slot = inc - c;
                const unsigned tag = (unsigned)lane_ << 26;
                if (c) {
                    for (unsigned o = off; o < end; o += kStep) { s_aitem[wave_][slot] = (o / kRec) | tag; ++slot; }
                    for (int rr = r; rr < n; ++rr) {
                        const uint2 e = s_rng[rr][own];
                        if (!(best < lb_unpack<T>(e.y & 0xffffu))) {
                            const unsigned r0 = e.x / kRec, r1 = r0 + (e.y >> 16);
                            for (unsigned o = r0; o < r1; o += (unsigned)kG) { s_aitem[wave_][slot] = o | tag; ++slot; }
                        }
                    }
                }
                s_abest[wave_][lane_] = BitsOf<T>::of(best);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                for (unsigned wb = 0; wb < W; wb += 64u) {          // (uniform trips; one register set: 8 waves per SIMD hide a trip's latency)
                    const unsigned w_ = wb + (unsigned)lane_;
                    const bool in_ = w_ < W;
                    const unsigned it_ = in_ ? s_aitem[wave_][in_ ? w_ : 0u] : 0u;
                    const int ql = in_ ? (int)(it_ >> 26) : lane_;
                    const typename GE::Raw g1 = GE::load(base, in_ ? rec_bytes<kRec>(it_ & 0x03ffffffu) : sent_off);
                    Pt4<T> qq; qq.x = __shfl(q.x, ql, 64); qq.y = __shfl(q.y, ql, 64); qq.z = __shfl(q.z, ql, 64); qq.idx = 0;
                    T d_[4]; GE::dists(g1, qq, d_);
                    const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);
                    if (in_) atomicMin(&s_abest[wave_][ql], BitsOf<T>::of(m_));
                }
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                best = BitsOf<T>::back(s_abest[wave_][lane_]);
                live = false;
            }
        }
    }
    if (live) {
        typename GE::Raw ga = GE::load(base, off), gb;
        for (;;) {
            unsigned coff = off;
            off += (unsigned)kG * kRec;
            if (off >= end) next_run();
            gb = GE::load(base, live ? off : sent_off);
            PCU_K1_EVAL(ga, coff)
            if (!live) break;
            coff = off;
            off += (unsigned)kG * kRec;
            if (off >= end) next_run();
            ga = GE::load(base, live ? off : sent_off);
            PCU_K1_EVAL(gb, coff)
            if (!live) break;
        }
    }
    // ---- radius 2, inside the launch (round 4). A query whose best is not certified by the 27 cells -- its nearest neighbour may lie
    // beyond them: 2 in 10,000 queries of a uniform cloud -- used to go to the wave-per-query pass: a launch of its own on the critical path of
    // every k = 1 call (10.5 of 164 us at 1M-vs-1M Chamfer) for a few hundred queries. Now the query's own wave serves it, the way that pass would: the
    // 5 x 5 rows x 5 cells around its cell are dealt to the lanes (a row each; one lane alone would walk ~125 dependent round trips and hold its
    // wave for longer than the rest of the launch takes -- measured: +19 us), every lane scans its row from scratch against the broadcast query,
    // a wave reduction picks the winner, and the query is certified against that box. The wave pass is launched only when a call's lists are not
    // empty afterwards (pcu_hip.hip: fused_wave_if_needed, knn_attempt). Waves with exited lanes (the last of a cloud) leave their stragglers to it.
    // Certification. Round 5: every face of the 27-cell box that exists lies at least one cell edge away from the query, less the face slacks and
    // the rounding of the face positions -- the query's own cell is bounded by face_above(c - 1) <= q < face_below(c + 1) (pcu_types.h), the box
    // by face_below(c - 1) / face_above(c + 1) -- so `quick` below, a grid-wide constant, is a lower bound of the exact bound of EVERY query
    // (same shrink, same squaring: monotone). A wave whose lanes all pass it skips the six face distances (88 vector instructions per wave; at two
    // points per cell 4 lanes in 10,000 fail it); the others compute the exact bound as before and may rescue their stragglers.
    T lb;                                                // the certification bound of the box the lane's result is checked against
    {
        const T smax = g.slack[0] > g.slack[1] ? (g.slack[0] > g.slack[2] ? g.slack[0] : g.slack[2]) : (g.slack[1] > g.slack[2] ? g.slack[1] : g.slack[2]);
        const T mag = ((fabs(g.org[0]) + fabs(g.org[1])) + fabs(g.org[2])) + (T)(Gx + Gy + Gz + 3) * g.h + smax;      // bounds every face position
        const T hq = (g.h - (T)2 * smax) - (T)16 * Limits<T>::eps * mag;
        const T hs = hq > (T)0 ? hq * shrink : (T)0;
        lb = hs * hs;
    }
#ifndef PCU_NO_RESCUE
#define PCU_NO_RESCUE 0
#endif
    if (__ballot(valid && !defer && !(best < lb)) != 0ull) {
        const int cx0 = max(ccx - 1, 0), cx1 = min(ccx + 1, Gx - 1);
        const int cy0 = max(ccy - 1, 0), cy1 = min(ccy + 1, Gy - 1), cz0 = max(ccz - 1, 0), cz1 = min(ccz + 1, Gz - 1);
        lb = face_lower_bound_inner(g, q.x, q.y, q.z, cx0, cx1, cy0, cy1, cz0, cz1);
        unsigned long long todo = __ballot(valid && !defer && !(best < lb));
        // (more than a few of them in one wave is not bad luck but the shape of the input -- a query cloud away from the dataset, every lane
        // uncertified: those go to the wave pass, whose rounds are built for it, without 64 futile box scans per wave first)
        if (!PCU_NO_RESCUE && todo && __popcll(todo) <= 4 && __ballot(true) == ~0ull) {
            const int lane_ = tid & 63;
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Pt4<T> sq;                                                   // the straggler's query, in scalar registers
                sq.x = __shfl(q.x, l, 64); sq.y = __shfl(q.y, l, 64); sq.z = __shfl(q.z, l, 64); sq.idx = 0;
                const int scx = grid_cell(g, 0, sq.x), scy = grid_cell(g, 1, sq.y), scz = grid_cell(g, 2, sq.z);
                const int bx0 = max(scx - 2, 0), bx1 = min(scx + 2, Gx - 1), by0 = max(scy - 2, 0), by1 = min(scy + 2, Gy - 1);
                const int bz0 = max(scz - 2, 0), bz1 = min(scz + 2, Gz - 1);
                const int ny_ = by1 - by0 + 1, nrows = ny_ * (bz1 - bz0 + 1);
                // two lanes per row of the box (<= 25 rows), each scans its half from scratch with its own running minimum
                T wbest = Limits<T>::max_v; unsigned wboff = 0xffffffffu, wtoff = 0xffffffffu; bool wtie = false, wtie2 = false;
                if ((lane_ >> 1) < nrows) {
                    const int r_ = lane_ >> 1;
                    const unsigned lo = (unsigned)row_run_lo(Gx, grid_row(Gy, by0 + r_ % ny_, bz0 + r_ / ny_), bx0, bx1);
                    const unsigned rs_ = a.cell_start[lo], re_ = a.cell_start[lo + (unsigned)(bx1 - bx0 + 1)];
                    const unsigned mid = rs_ + (((re_ - rs_ + 1u) >> 1) + 3u) / 4u * 4u;            // (whole groups in the first half: no record is seen twice)
                    const unsigned s_ = ((lane_ & 1) ? min(mid, re_) : rs_) * kRec, e_ = ((lane_ & 1) ? re_ : min(mid, re_)) * kRec;
                    for (unsigned off_ = s_; off_ < e_; off_ += (unsigned)kG * kRec) {
                        T d_[4]; GE::dists(GE::load(base, off_), sq, d_);
                        if (!(lane_ & 1)) {                                   // (the first half's last group stops at the half's end)
#pragma unroll
                            for (int u = 1; u < 4; ++u) d_[u] = off_ + (unsigned)u * kRec >= e_ ? (T)INFINITY : d_[u];      // (+inf, not kill_if's bit pattern: a signalling NaN would poison the v_min chain)
                        }
                        const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);
                        const bool eq_ = m_ == wbest, lt_ = m_ < wbest;
                        if (!VALUE_ONLY) { wtie2 = !lt_ && (wtie2 || (wtie && eq_)); wtie = !lt_ && (wtie || eq_); wtoff = eq_ ? off_ : wtoff; wboff = lt_ ? off_ : wboff; }
                        wbest = lt_ ? m_ : wbest;
                    }
                }
                T mn = wbest;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const T ot = __shfl_xor(mn, o, 64); mn = ot < mn ? ot : mn; }
                const T lb2 = face_lower_bound_inner(g, sq.x, sq.y, sq.z, bx0, bx1, by0, by1, bz0, bz1);
                if (VALUE_ONLY) {                                             // (the sum / the value-only arg-max need the value only)
                    if (lane_ == l) { best = mn; lb = lb2; }
                    continue;
                }
                const unsigned long long eqm = __ballot(wbest == mn && wboff != 0xffffffffu);
                const int wl = eqm ? __ffsll((long long)eqm) - 1 : 0;        // a lane that holds the minimum
                const bool many = __popcll(eqm) > 1;                          // ... met by two lanes: a possible tie
                const unsigned rb = (unsigned)__shfl((int)wboff, wl, 64), rt = many ? (unsigned)__shfl((int)wboff, 63 - __clzll((long long)eqm), 64) : (unsigned)__shfl((int)wtoff, wl, 64);
                // (the flags of EVERY lane that holds the minimum count: with two such lanes, one of which met the minimum in two of its groups, there are
                // three encounters -- consulting only the first lane's flags let the final "same record met twice" check clear a genuine tie: one query in
                // 250k of a duplicated cloud, depending on the records' order inside their cell, found by the randomised sweep)
                const bool holds = wbest == mn && wboff != 0xffffffffu;
                const bool any_tie = __ballot(holds && wtie) != 0ull, any_tie2 = __ballot(holds && wtie2) != 0ull;
                const bool rtie = many || any_tie, rtie2 = (many && __popcll(eqm) > 2) || any_tie2 || (many && any_tie);
                if (lane_ == l) {
                    best = mn; lb = lb2; boff = eqm ? rb : 0xffffffffu; toff = rt; tie = rtie; tie2 = rtie2;
                }
            }
        }
    }
#undef PCU_K1_EVAL
    // ---- which record of the winning group it was; ties (see k_search1)
    int bi[1] = {0x7fffffff};
    auto which = [&](unsigned goff, int& hits, unsigned& rec_off) {         // records of the group at `goff` whose d2 equals the minimum
        T d_[4]; GE::dists(GE::load(base, goff), q, d_);
        hits = 0; rec_off = 0xffffffffu;
#pragma unroll
        for (int u = kG - 1; u >= 0; --u) { const bool eq = d_[u] == best; hits += eq ? 1 : 0; rec_off = eq ? goff + (unsigned)u * kRec : rec_off; }
    };
    auto row_id = [&](unsigned rec_off) -> int {
        if (!XYZ) return (int)reinterpret_cast<const Pt4<T>*>(base + (size_t)rec_off)->idx;
        return a.ref_idx[rec_off / kRec];
    };
    if (!VALUE_ONLY && boff != 0xffffffffu) {             // (the value-only programs need neither the row id nor the tie flags)
        int hits; unsigned ro;
        which(boff, hits, ro);
        if (ro != 0xffffffffu) bi[0] = row_id(ro);
        if (hits > 1) { tie = true; tie2 = true; }
        if (tie && !tie2) {
            int h2; unsigned ro2;
            which(toff, h2, ro2);
            if (h2 == 1 && ro2 == ro) tie = false;        // the same record met twice (groups run past their run's end)
        }
    }
    if (FUSE == FUSE_NONE) {                              // result rows (finish_lane for k = 1, with the bound computed above)
        if (defer) { wave_append(valid, qpos, a.ties, a.n_ties); return; }       // nothing was scanned: the wave-per-query pass at the same radius takes over
        const bool certified = valid && best < lb;
        if (certified) {
            const size_t o = (size_t)(a.row_out ? (int)q.idx : qpos) * (size_t)a.kreq;
            const bool found = bi[0] != 0x7fffffff;
            a.out_i[o] = found ? (long long)bi[0] : -1ll;
            a.out_d[o] = found ? (a.squared ? best : sqrt(best)) : (T)-1;
        }
        const int us = wave_append(valid && !certified, qpos, a.unresolved, a.n_unresolved);
        if (us >= 0 && a.ubound) a.ubound[us] = best;
        wave_append(certified && tie, qpos, a.ties, a.n_ties);
        return;
    }
    // fused epilogue: the lane's distance goes into the block's partial (kernel wrapper) instead of a result row
    if (defer) {                 // nothing was scanned: the wave-per-query pass at the same radius takes over
        wave_append(valid, qpos, a.ties, a.n_ties);
        return;
    }
    const bool certified = valid && best < lb;
    const int us = wave_append(valid && !certified, qpos, a.unresolved, a.n_unresolved);
    if (us >= 0 && a.ubound) a.ubound[us] = best;
    f_ok = certified;
    f_v = a.squared ? best : sqrt(best);
    f_key = FUSE == FUSE_MAXVAL ? (((long long)q.idx << 32) | kKeyUnresolved | (long long)qpos)
                                : (((long long)q.idx << 32) | (long long)((unsigned)bi[0] | (tie ? 0x80000000u : 0u)));
}
// -------------------------------------------------------------------------------------------------------
// Main pass for k > 1 on an OPEN index (round 5): k_search's scan on the k = 1 kernel's plan. k_search walks the nine rows one after the
// other, every row as long as its longest lane (44 % active lanes at 4M-vs-4M, k = 16: profiles/r04_c3_pmc.txt) and with no cell cuts. Here
//   * the row tables come from the per-axis terms the k = 1 kernel shares between its rows (uniform four-word tables, see search1_flat_body);
//   * the centre row is scanned whole and its candidates inserted, which gives a first k-th best (the default occupancy for k > 1 puts
//     ~3 k / 2 points into those three cells);
//   * the outer cells of the other eight rows are cut against that k-th best, the surviving runs go to the lane's list in LDS, and ONE
//     software-pipelined loop consumes the list -- a wave runs max-over-lanes of the TOTAL group count; a run is dropped when its row bound
//     has been overtaken by the k-th best by the time the lane reaches it.
// Candidates are still the 16 / 32-byte Pt4 records (a candidate that is accepted needs its row id at once), slots past a run's end read the
// +inf sentinel record (a record offered twice would sit twice in the list), accepted candidates are parked and inserted in bursts exactly
// as in k_search: same offers, same tie flags, same certification (finish_lane). Closed sub-box levels keep k_search.
#ifndef PCU_RUNS_MINW
#define PCU_RUNS_MINW 1
#endif
template <typename T, int K>
__global__ __launch_bounds__(kBlock, (sizeof(T) == 4 && K == 16) ? PCU_RUNS_MINW : 1) void k_search_runs(const SearchArgs<T> a) {
    static_assert(K > 1, "k = 1 has k_search1_flat");
    const int per = (int)(gridDim.x >> 3);
    const int vb = (int)(blockIdx.x & 7) * per + (int)(blockIdx.x >> 3);      // XCD-aware block order, see k_search
    const int tid = threadIdx.x;
    const int t = vb * kBlock + tid;
    const int nq = a.qcount_dev ? *a.qcount_dev : a.nq;
    const GridParams<T>& g = *a.gp;
    if (t >= nq) return;
    const int qpos = a.qlist ? a.qlist[t] : t;
    const Pt4<T> q = a.qsorted[qpos];
    if (const int hl = index_not_ready(a, g)) { if (t == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (t == 0) *a.skew_flag = (float)g.sumsq > a.skew_far ? 2 : 1; return; }
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = cell_of_query(g, 0, q.x), ccy = cell_of_query(g, 1, q.y), ccz = cell_of_query(g, 2, q.z);

    T bd[K]; int bi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) { bd[i] = Limits<T>::max_v; bi[i] = 0x7fffffff; }
    bool tie = false;
    constexpr int kBuf = PCU_KBUF;
    constexpr int kGroup = 4;
    __shared__ T s_bd[kBuf][kBlock];
    __shared__ int s_bi[kBuf][kBlock];
    __shared__ uint2 s_rng[8][kBlock];
    int cnt = 0;
    auto flush = [&]() {
#ifdef PCU_EXPERIMENT_NOINSERT          /* (scratch experiment: WRONG results; prices the insertion's share of the kernel) */
        if (cnt > 0) { bd[K - 1] = s_bd[0][tid] < bd[K - 1] ? s_bd[0][tid] : bd[K - 1]; bi[K - 1] = s_bi[0][tid]; }
        cnt = 0; return;
#endif
        int mx = cnt;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) mx = max(mx, __shfl_xor(mx, o, 64));
        for (int i = 0; i < mx; ++i)
            if (i < cnt) offer<T, K>(s_bd[i][tid], s_bi[i][tid], bd, bi, tie);
        cnt = 0;
    };
    const unsigned sentinel = a.n_ref;
#ifndef PCU_RUNS_XYZ
#define PCU_RUNS_XYZ 0
#endif
    // Round 6 (PCU_RUNS_XYZ=1; measured equal to slightly slower on config 3, profiles/r06_c3_ab.txt -- off): candidates from the coordinates-only stream (12 / 24 bytes per record: a group of four = THREE 16-byte gathers instead of
    // four, and 12 instead of 16 registers per group in flight), as the k = 1 kernel's do; an accepted candidate is parked with its POSITION in
    // the stream, and the rows of the kreq best are looked up once, at the end (finish_lane<POS>). Slots at or past the run's end e are real
    // records of the next cells (or the +inf sentinels): their distance is replaced by +inf -- a record offered twice would sit twice in the list.
    constexpr bool XYZ = PCU_RUNS_XYZ != 0;
    typedef GroupEval<T, true> GE;
    struct Grp { typename GE::Raw raw; Pt4<T> c[XYZ ? 1 : kGroup]; unsigned p, e; };
    const char* const base = reinterpret_cast<const char*>(a.ref);
    const char* const xbase = reinterpret_cast<const char*>(a.ref_xyz);
    auto load_group = [&](unsigned p, unsigned e, Grp& gr) {
        gr.p = p; gr.e = e;
        if (XYZ) { gr.raw = GE::load(xbase, rec_bytes<GE::kRec>(p)); return; }
#pragma unroll
        for (int u = 0; u < (XYZ ? 1 : kGroup); ++u) {
            const unsigned idx = (p + u < e) ? p + u : sentinel;
            gr.c[u] = *reinterpret_cast<const Pt4<T>*>(base + (size_t)(idx * (unsigned)sizeof(Pt4<T>)));
        }
    };
    auto eval_group = [&](const Grp& gr) {
        T dd[kGroup];
        if (XYZ) GE::dists(gr.raw, q, dd);
#pragma unroll
        for (int u = 0; u < kGroup; ++u) {
            const T d = XYZ ? kill_if(dd[u], gr.p + (unsigned)u >= gr.e) : dist2(q, gr.c[XYZ ? 0 : u]);
            tie = tie || (d == bd[K - 1]);            // as offer() would flag it (the k-th best may be stale: conservative)
            if (d < bd[K - 1]) { s_bd[cnt][tid] = d; s_bi[cnt][tid] = XYZ ? (int)(gr.p + (unsigned)u) : (int)gr.c[XYZ ? 0 : u].idx; ++cnt; }
        }
        if (__any(cnt > kBuf - kGroup)) flush();
    };

    // ---- row tables (search1_flat_body)
    const bool hasxl = ccx > 0, hasxh = ccx < Gx - 1;
    const bool odd0 = ((ccy ^ ccz) & 1) != 0, zodd = (ccz & 1) != 0;
    const bool hasA = (odd0 && hasxh) || (!odd0 && hasxl), hasB = (odd0 && hasxl) || (!odd0 && hasxh);
    const unsigned Gx4 = (unsigned)Gx << 2;
    const unsigned xe4 = (unsigned)ccx << 2, xo4 = (unsigned)(Gx - 1 - ccx) << 2;
    const unsigned xS4 = odd0 ? xo4 : xe4, xR4 = odd0 ? xe4 : xo4;
    const int yS = zodd ? Gy - 1 - ccy : ccy, sS = zodd ? -1 : 1;
    const int row0 = (int)mad24((unsigned)ccz, (unsigned)Gy, (unsigned)yS);
    const int flip = (Gy - 1 - yS) - yS;
    const int rowM = row0 + flip - Gy, rowP = row0 + flip + Gy;
    const bool okyM = ccy > 0, okyP = ccy < Gy - 1, okzM = ccz > 0, okzP = ccz < Gz - 1;
    const char* const tbase = reinterpret_cast<const char*>(a.cell_start) - 4;
    bool okj[9];
    CellStart4 tb[9];
    auto row_table = [&](int j) {
        const int oy = kRowOy[j], oz = kRowOz[j];
        okj[j] = (oy == 0 || (oy < 0 ? okyM : okyP)) && (oz == 0 || (oz < 0 ? okzM : okzP));
        const int rr = oz == 0 ? row0 + oy * sS : (oz < 0 ? rowM : rowP) - oy * sS;
        const unsigned row = (unsigned)(okj[j] ? rr : row0);
        tb[j] = *reinterpret_cast<const CellStart4*>(tbase + (size_t)(__umul24(row, Gx4) + (((oy + oz) & 1) == 0 ? xS4 : xR4)));
    };
    row_table(0);                                     // (the other eight after the centre scan: 32 registers less while it runs)
    // ---- centre row: whole run, pipelined
    const unsigned c_s = hasA ? tb[0].v[0] : tb[0].v[1], c_e = hasB ? tb[0].v[3] : tb[0].v[2];
    const unsigned cnt0 = c_e - c_s;
    const unsigned cand_cap = a.lane_max_cand < 65535u ? a.lane_max_cand : 65535u;     // a run's record count is packed into 16 bits
    bool defer = cnt0 > cand_cap;
    if (!defer && cnt0 > 0u) {
        unsigned p = c_s;
        Grp ca, cb;
        load_group(p, c_e, ca);
        for (;;) {
            p += kGroup;
            load_group(p, c_e, cb);                  // (past the end: four sentinel records, evaluated to +inf and never parked)
            eval_group(ca);
            if (!(p < c_e)) break;
            p += kGroup;
            load_group(p, c_e, ca);
            eval_group(cb);
            if (!(p < c_e)) break;
        }
    }
#pragma unroll
    for (int j = 1; j < 9; ++j) row_table(j);
    if (__any(cnt > 0)) flush();                      // the cuts below want the k-th best of everything seen so far (the tables' latency is behind it)
    // ---- bounds, cuts, run list
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    T mxl = q.x - face_below(g, 0, ccx); mxl = mxl > (T)0 ? mxl * shrink : (T)0;
    T mxh = face_above(g, 0, ccx) - q.x; mxh = mxh > (T)0 ? mxh * shrink : (T)0;
    const T mxl2 = mxl * mxl, mxh2 = mxh * mxh;
    const T mA2 = odd0 ? mxh2 : mxl2, mB2 = odd0 ? mxl2 : mxh2;
    T my2[3], mz2[3];
    {
        T m;
        my2[0] = (T)0; mz2[0] = (T)0;
        m = q.y - face_below(g, 1, ccy); m = m > (T)0 ? m * shrink : (T)0; my2[1] = m * m;
        m = face_above(g, 1, ccy) - q.y; m = m > (T)0 ? m * shrink : (T)0; my2[2] = m * m;
        m = q.z - face_below(g, 2, ccz); m = m > (T)0 ? m * shrink : (T)0; mz2[1] = m * m;
        m = face_above(g, 2, ccz) - q.z; m = m > (T)0 ? m * shrink : (T)0; mz2[2] = m * m;
    }
    unsigned total = cnt0;
    int n = 0;
    {
        const T kth = bd[K - 1];
#pragma unroll
        for (int j = 1; j < 9; ++j) {
            const int oy = kRowOy[j], oz = kRowOz[j];
            const T ry = my2[oy == 0 ? 0 : (oy < 0 ? 1 : 2)], rz = mz2[oz == 0 ? 0 : (oz < 0 ? 1 : 2)];
            const bool same = ((oy + oz) & 1) == 0;
            const T mF2 = same ? mA2 : mB2, mL2 = same ? mB2 : mA2;
            const bool hasF = same ? hasA : hasB, hasL = same ? hasB : hasA;
            const T rlb = oy == 0 ? rz : (oz == 0 ? ry : ry + rz);
            const T bF = oy == 0 ? mF2 + rz : (oz == 0 ? mF2 + ry : (mF2 + ry) + rz), bL = oy == 0 ? mL2 + rz : (oz == 0 ? mL2 + ry : (mL2 + ry) + rz);
            const bool cutF = !hasF || kth < bF, cutL = !hasL || kth < bL;          // (strict: a tie at the k-th slot is not cut away)
            const unsigned s_run = cutF ? tb[j].v[1] : tb[j].v[0], e_run = cutL ? tb[j].v[2] : tb[j].v[3];
            const bool take = okj[j] && !defer && !(kth < rlb) && e_run > s_run;
            s_rng[n][tid] = make_uint2(s_run, ((e_run - s_run) << 16) | lb_pack(rlb));
            n += take ? 1 : 0;
            total += take ? e_run - s_run : 0u;
        }
    }
    if (total > cand_cap) { defer = true; n = 0; cnt = 0; }
    // ---- one loop over the listed runs
    int r = 0;
    unsigned p = 0, end = 0;
    bool live = false;
    auto next_run = [&]() {
        live = false;
        while (r < n) {
            const uint2 e = s_rng[r][tid];
            ++r;
            if (!(bd[K - 1] < lb_unpack<T>(e.y & 0xffffu))) { p = e.x; end = e.x + (e.y >> 16); live = true; break; }
        }
    };
    next_run();
    if (live) {
        Grp ga, gb;
        load_group(p, end, ga);
        for (;;) {
            p += kGroup;
            if (p >= end) next_run();
            load_group(live ? p : sentinel, live ? end : sentinel, gb);
            eval_group(ga);
            if (!live) break;
            p += kGroup;
            if (p >= end) next_run();
            load_group(live ? p : sentinel, live ? end : sentinel, ga);
            eval_group(gb);
            if (!live) break;
        }
    }
    if (__any(cnt > 0)) flush();
    finish_lane<T, K, XYZ>(a, g, q, qpos, max(ccx - 1, 0), min(ccx + 1, Gx - 1), max(ccy - 1, 0), min(ccy + 1, Gy - 1), max(ccz - 1, 0), min(ccz + 1, Gz - 1),
                           bd, bi, tie, true, defer);
}

// Both directions of a two-sided call (x in y, y in x) in ONE launch: blocks [0, nb0) serve a0, the rest a1.
template <typename T> struct SearchArgs2 { SearchArgs<T> a[2]; };
template <typename T, bool EARLY, int MINW, int FUSE>
__global__ __launch_bounds__(kBlock, MINW) void k_search1_flat(const SearchArgs2<T> p, int nb0) {
    // (the side's arguments are read through an index into the kernel-argument segment; selecting between two by-value
    // structs by reference makes the compiler copy the chosen one to scratch)
    const int side = (int)blockIdx.x >= nb0 ? 1 : 0;
    const int bid = side ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    bool ok = false; T v = (T)0; long long key = 0x7fffffffffffffffll;
    // (nq is picked by a scalar select: read through p.a[side] the compiler parks both sides' values in SCRATCH to index them --
    // 8 bytes of private memory written per lane, 16 MB of HBM writes per 1M-vs-1M launch, profiles/r02a_pmc.txt)
    const int nq_side = side ? p.a[1].nq : p.a[0].nq;
    search1_flat_body<T, EARLY, FUSE>(p.a[side], nq_side, bid, side ? (int)gridDim.x - nb0 : nb0, ok, v, key);
    if (FUSE == FUSE_SUM) {                     // one fp64 partial per block; lanes in a fixed order: reproducible
        const double r = block_sum(ok ? (double)v : 0.0);
        if (threadIdx.x == 0) p.a[side].f_sum[bid] = r;
    } else if (FUSE == FUSE_ARGMAX || FUSE == FUSE_MAXVAL) {           // first maximum by source row (Eigen's maxCoeff visits rows in order, strict '>')
        T bv = ok ? v : -Limits<T>::max_v; long long bk = ok ? key : 0x7fffffffffffffffll;
        block_argmax(bv, bk);
        if (threadIdx.x == 0) { p.a[side].f_max_v[bid] = bv; p.a[side].f_max_k[bid] = bk; }
    }
}

// -------------------------------------------------------------------------------------------------------
// Wave-cooperative search: ONE WAVE per query, for the few queries the lane-per-query pass could not finish
// (escalation to a wider radius R, or re-resolution of a possible tie). The (2R+1)^2 rows of the query's cell
// box are dealt to the 64 lanes (lane <- row); each lane keeps the K best of its rows in registers under the
// total order (d2, dataset row); the K global best are then extracted by K rounds of a wave-wide
// lexicographic arg-min (DPP/shuffle butterflies), which also exposes genuine ties (equal d2 at consecutive
// ranks up to the (k+1)-th). Work items come from a device-side list + count, so the launch needs no host
// round trip: the grid is fixed and waves stride over the list.
template <typename T>
__device__ __forceinline__ bool lex_less(T d, int id, T d2, int id2) { return d < d2 || (d == d2 && id < id2); }

// (Measured and rejected, rounds 2 and 3: letting the block that finishes last also fold the fused call -- ticket, fuse_tail_body with
// the block's 256 threads, hand-off to the host -- instead of the separate one-block k_fuse_tail launch: 21.4 us against 9.3 + 7.0 us.
// The fold then runs strictly after the slowest block on a quarter of the threads; a launch boundary costs less.)
// (scratch experiments: the wave pass without the subsample bound / without dropping candidates beyond the batch bound)
#ifndef PCU_NO_SEED
#define PCU_NO_SEED 0
#endif
#ifndef PCU_NO_BOUND_KILL
#define PCU_NO_BOUND_KILL 0
#endif
template <typename T, int K>
__global__ __launch_bounds__(kBlock) void k_search_wave(const SearchArgs<T> a0, const SearchArgs<T> a1, int njobs, const int n_blocks) {
    const int lane = threadIdx.x & 63;
    // fused sum: the block's exact accumulators per direction (reduce.h: exact_add; LDS atomics, flushed once at the end)
    __shared__ unsigned long long s_limbs[2][kAccLimbs];
    __shared__ double s_special[2];
    if (a0.fuse == FUSE_SUM) {
        for (int i = threadIdx.x; i < 2 * kAccLimbs; i += kBlock) (&s_limbs[0][0])[i] = 0ull;
        if (threadIdx.x < 2) s_special[threadIdx.x] = 0.0;
        __syncthreads();
    }
    // fused arg-max: this wave's best per direction (named scalars: an array indexed by the direction is promoted to LDS, and
    // addressing it needs the block dimensions, i.e. a fetch of the dispatch packet in the prologue)
    T fbv0 = -Limits<T>::max_v, fbv1 = -Limits<T>::max_v;
    long long fbk0 = 0x7fffffffffffffffll, fbk1 = 0x7fffffffffffffffll;
    const int wave = (blockIdx.x * kBlock + threadIdx.x) >> 6;
    if (a0.fuse == FUSE_ARGMAX && a0.f_accum) {        // (only the host-driven straggler passes of a fused call)
        fbv0 = a0.f_wave_v[wave]; fbk0 = a0.f_wave_k[wave];
        if (njobs > 1) { fbv1 = a1.f_wave_v[wave]; fbk1 = a1.f_wave_k[wave]; }
    }
    const int nwaves = (n_blocks * kBlock) >> 6;       // (n_blocks = gridDim.x, as an argument: reading gridDim costs a fetch of the dispatch packet)
    // work items: job 0's list(s), then (two-sided calls) job 1's. This launch usually serves a few hundred queries, so its
    // time is its chain of dependent memory round trips; the six device words its prologue needs are therefore fetched
    // unconditionally and together (absent lists point at a valid dummy word) instead of one pointer test + load + wait each.
    const int* const c00 = a0.qcount_dev ? a0.qcount_dev : a0.skew_flag; const int* const c01 = a0.qlist2 ? a0.qcount2_dev : a0.skew_flag;
    const int* const c10 = a1.qcount_dev ? a1.qcount_dev : a1.skew_flag; const int* const c11 = a1.qlist2 ? a1.qcount2_dev : a1.skew_flag;
    const int v00 = *c00, v01 = *c01, v10 = *c10, v11 = *c11;
    const unsigned long long ss0 = a0.gp->sumsq, ss1 = a1.gp->sumsq;
    const int hl0 = index_not_ready(a0, *a0.gp), hl1 = index_not_ready(a1, *a1.gp);
    int total0 = (a0.qcount_dev ? v00 : a0.nq) + (a0.qlist2 ? v01 : 0);
    int total1 = njobs > 1 ? (a1.qcount_dev ? v10 : a1.nq) + (a1.qlist2 ? v11 : 0) : 0;
    if (a0.skew_limit > 0.f && ((float)ss0 > a0.skew_limit || (float)ss0 < a0.skew_lo)) { if (wave == 0 && lane == 0) *a0.skew_flag = (float)ss0 > a0.skew_far ? 2 : 1; total0 = 0; }
    if (njobs > 1 && a1.skew_limit > 0.f && ((float)ss1 > a1.skew_limit || (float)ss1 < a1.skew_lo)) { if (wave == 0 && lane == 0) *a1.skew_flag = (float)ss1 > a1.skew_far ? 2 : 1; total1 = 0; }
    if (hl0) { if (wave == 0 && lane == 0) a0.skew_flag[kLargeFlag] = hl0; total0 = 0; }
    if (njobs > 1 && hl1) { if (wave == 0 && lane == 0) a1.skew_flag[kLargeFlag] = hl1; total1 = 0; }
    long long t_poll = wall_clock64();
    for (int wg = wave; wg < total0 + total1; wg += nwaves) {
        // (cancellation: one uncached host-memory load, i.e. a PCIe round trip -- at most once per 200 us of a wave's life; looking between ALL work
        // items made a 33k-straggler launch 7 x slower: 2048 waves x 16 items queueing on the link, profiles/r06_configs.jsonl history)
        { const long long t_now = wall_clock64(); if (t_now - t_poll > 20000ll) { t_poll = t_now; if (cancel_seen(a0.cancel_word, a0.cancel_gen)) break; } }
        const bool job1 = wg >= total0;
        const SearchArgs<T>& a = job1 ? a1 : a0;
        const int w = job1 ? wg - total0 : wg;
        const int nq1 = a.qcount_dev ? (job1 ? v10 : v00) : a.nq;
        const GridParams<T>& g = *a.gp;
        const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
        const int kreq = a.kreq;
        const bool second = w >= nq1;
        int R = second ? a.R2 : a.R;
        const int qpos = second ? a.qlist2[w - nq1] : (a.qlist ? a.qlist[w] : w);
        const bool lean = a.lean != 0;              // (uniform) records come as two streams instead of Pt4
        auto rec_at = [&](const Pt4<T>* recs, const T* xyz, const int* ids, unsigned i) -> Pt4<T> {
            if (!lean) return recs[i];
            Pt4<T> r; r.x = xyz[3 * (size_t)i]; r.y = xyz[3 * (size_t)i + 1]; r.z = xyz[3 * (size_t)i + 2]; r.idx = ids[i];
            return r;
        };
        const Pt4<T> q = rec_at(a.qsorted, a.q_xyz, a.q_idx, (unsigned)qpos);
        const int ccx = grid_cell(g, 0, q.x), ccy = grid_cell(g, 1, q.y), ccz = grid_cell(g, 2, q.z);
        const T shrink = (T)1 - (T)4 * Limits<T>::eps;

        T bd[K]; int bi[K];
        // one candidate into this lane's K best under the total order (d2, dataset row)
        auto take = [&](const T d, const int id) {
            if (lex_less(d, id, bd[K - 1], bi[K - 1])) {
#pragma unroll
                for (int i = K - 1; i > 0; --i) {
                    if (K >= 8 && (i & 3) == 3 && i != K - 1) { if (!__any(lex_less(d, id, bd[i], bi[i]))) break; }      // (see offer(): nothing below moves)
                    const bool gm = lex_less(d, id, bd[i - 1], bi[i - 1]);
                    const bool gi = lex_less(d, id, bd[i], bi[i]);
                    bd[i] = gm ? bd[i - 1] : (gi ? d : bd[i]);
                    bi[i] = gm ? bi[i - 1] : (gi ? id : bi[i]);
                }
                if (K == 1 || lex_less(d, id, bd[0], bi[0])) { bd[0] = d; bi[0] = id; }
            }
        };
        constexpr unsigned kHeavyRow = 256;        // a row with more candidates than this is scanned by the whole wave
        T my_d, my_d2, kth; int my_i, my_i2; bool tie, certified;
        // A query is served in up to a few rounds, all inside this launch (SearchArgs::escalate; without it one round, and what is
        // left uncertified goes to the `unresolved` list for the host-driven passes):
        //   box round   the (2R+1)^3 cells around the query's cell, certified by face_lower_bound as in the lane passes;
        //   ball round  if the box round found k points but could not certify them, its k-th best d2 = B bounds the true one, and
        //               every point that can still matter lies in the ball of radius sqrt(B): the rows (cy, cz) whose slab is within
        //               B of the query, and of each such row the cells within sqrt(B - row bound) along x (+-1 cell of margin; the
        //               bounds are the rounding-safe ones of the row pruning, strict '>': no tie can hide in what is skipped).
        //               Whatever the round returns is final by construction -- however far the query is from its neighbours, and
        //               without coarser grids or a host round trip per radius;
        //   wider box   if the box held fewer than k points, its radius grows four-fold (up to the whole grid).
        const bool esc = a.escalate && !g.closed;   // (a closed sub-box level does not hold the points beyond its box: nothing to finish there)
        // distance from the query to the data's bounding box along y / z (0 inside; rounding-safe as in face_lower_bound): a lower bound for every
        // dataset point. (Closed sub-box levels hold a subset of the data, for which the whole cloud's box is still a valid bound.)
        T oy, oz;
        { T m = g.gmin[1] - q.y, n_ = q.y - g.gmax[1]; m = m > n_ ? m : n_; oy = m > (T)0 ? m * shrink : (T)0; }
        { T m = g.gmin[2] - q.z, n_ = q.z - g.gmax[2]; m = m > n_ ? m : n_; oz = m > (T)0 ? m * shrink : (T)0; }
        // < max_v: ball round with this bound. A straggler of the lane pass brings the k-th best that pass found: no box round needed.
        T ball = (esc && second && a.qbound2) ? a.qbound2[w - nq1] : Limits<T>::max_v;
        if (!(ball < Limits<T>::max_v)) ball = Limits<T>::max_v;          // (NaN never: d2 of kept candidates are ordered; defensive)
        // (bound seeding, round 4) A box that holds fewer than k points says the query sits in a sparse part of the grid -- typically a query cloud
        // far from the dataset, every query clamped to the same border cell. Growing the box four-fold until it holds k points made the first bound
        // the k-th best of most of the grid, and the ball round behind it a scan of a thick cap of it (231k such queries against a 167k-point sphere,
        // k = 16: 0.65 s, slower than the reference's kd-tree). Instead the wave first takes the k best of a stratified SUBSAMPLE of the dataset (up
        // to 1024 records, evenly spaced in cell order): an upper bound of the k-th distance that is tight to within the sample's spacing, which the
        // ball round then turns into the exact answer by scanning a thin shell.
        bool seed_next = false, seeded = false;
        for (int round = 0;; ++round) {
            const bool is_seed = seed_next;
            const bool is_ball = ball < Limits<T>::max_v;
            int x0, x1, y0, y1, z0, z1;
            if (!is_ball) {
                x0 = max(ccx - R, 0); x1 = min(ccx + R, Gx - 1);
                y0 = max(ccy - R, 0); y1 = min(ccy + R, Gy - 1);
                z0 = max(ccz - R, 0); z1 = min(ccz + R, Gz - 1);
            } else {
                const T rr = sqrt(ball) * ((T)1 + (T)8 * Limits<T>::eps);
                x0 = 0; x1 = Gx - 1;                // (per row, below)
                y0 = max(grid_cell(g, 1, q.y - rr) - 1, 0); y1 = min(grid_cell(g, 1, q.y + rr) + 1, Gy - 1);
                z0 = max(grid_cell(g, 2, q.z - rr) - 1, 0); z1 = min(grid_cell(g, 2, q.z + rr) + 1, Gz - 1);
            }
            const int ny = y1 - y0 + 1, nrows = ny * (z1 - z0 + 1);
#pragma unroll
            for (int i = 0; i < K; ++i) { bd[i] = Limits<T>::max_v; bi[i] = 0x7fffffff; }
            // (2R+1)^2 <= 32 rows (radius 1 and 2, i.e. nearly every query that gets here): two lanes per row, each takes half of it --
            // the pass is a chain of dependent loads per lane, so halving the chain halves the query
            const int sp = nrows <= 32 ? 2 : 1;
            // Rows go in batches of 64. Between batches the bound tightens: every lane's K-th best so far bounds the query's K-th neighbour
            // from above, so later rows are tested against the smallest of them exactly as a ball round tests against its B (rounding-safe
            // row and cell bounds, strict comparisons: what is skipped lies beyond the bound, ties included never) -- in ball rounds B only
            // shrinks, in box rounds a bound appears as soon as one lane holds K points. Batches go outward from the query's own row.
            // (231k queries at offset 1000 from a 167k-point sphere, k = 16: the first box with 16 points was the whole grid and every
            // query scanned all of it, 1.6 s; scratch/case283.py.) A candidate beyond the batch's bound is not offered to the lists at all (round 4):
            // the bound is at least the true k-th distance, so such a point is never among the k best, nor tied with the k-th -- and the K-wide
            // insertion, which the whole wave executes whenever one lane accepts, is what a far query's thousands of shell candidates cost.
            const int step = 64 / sp, nbat = is_seed ? 0 : (nrows + step - 1) / step;
            if (is_seed) {
                const unsigned nsamp = a.n_ref < 1024u ? a.n_ref : 1024u;
                constexpr int kS = K <= 32 ? 4 : 1;
                for (unsigned j0 = (unsigned)lane; j0 < nsamp; j0 += 64u * kS) {
                    Pt4<T> cc[kS];
#pragma unroll
                    for (int u = 0; u < kS; ++u) {
                        const unsigned j = min(j0 + 64u * (unsigned)u, nsamp - 1u);
                        cc[u] = rec_at(a.ref, a.ref_xyz, a.ref_idx, a.n_ref < 1024u ? j : (unsigned)(((unsigned long long)j * (unsigned long long)a.n_ref) >> 10));
                    }
#pragma unroll
                    for (int u = 0; u < kS; ++u) {
                        const Pt4<T>& c = cc[u];
                        const T dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
                        take(kill_if(((dx * dx) + (dy * dy)) + (dz * dz), u > 0 && j0 + 64u * (unsigned)u >= nsamp), (int)c.idx);
                    }
                }
            }
            const int cb = ((min(max(ccz, z0), z1) - z0) * ny + (min(max(ccy, y0), y1) - y0)) / step;
            T bw = Limits<T>::max_v;
            int b_lo = cb, b_hi = cb + 1;               // batches outward from the query's own, alternating sides
            for (int bt = 0; bt < nbat; ++bt) {
                const bool down = b_hi >= nbat || (b_lo >= 0 && !(bt & 1));
                const int r0 = (down ? b_lo-- : b_hi++) * step;
                const int r = r0 + (sp == 2 ? lane >> 1 : lane);
                // (the bound with the near-tie margin of the extraction below: what is pruned or dropped lies beyond every near tie of the k-th best)
                const T bound0 = is_ball && !(bw < ball) ? ball : bw;
                const T bound = bound0 < Limits<T>::max_v ? bound0 * ((T)1 + (T)8 * Limits<T>::eps) : bound0;
                unsigned s = 0, e = 0;
                if (r < nrows) {
                    const int cz = z0 + r / ny, cy = y0 + r % ny;
                    int xa = x0, xb = x1;
                    bool on = true;
                    if (bound < Limits<T>::max_v) {
                        // distance from the query to the slab of this row (0: the query's own), as in row_lower_bounds
                        T my = (T)0, mz = (T)0;
                        if (cy < ccy) { const T m = q.y - face_below(g, 1, cy + 1); my = m > (T)0 ? m * shrink : (T)0; }
                        if (cy > ccy) { const T m = face_above(g, 1, cy - 1) - q.y; my = m > (T)0 ? m * shrink : (T)0; }
                        if (cz < ccz) { const T m = q.z - face_below(g, 2, cz + 1); mz = m > (T)0 ? m * shrink : (T)0; }
                        if (cz > ccz) { const T m = face_above(g, 2, cz - 1) - q.z; mz = m > (T)0 ? m * shrink : (T)0; }
                        my = my > oy ? my : oy; mz = mz > oz ? mz : oz;      // (a query outside the data's box: at least its distance to the box, also in its own -- clamped -- rows)
                        const T rlb = (my * my) + (mz * mz);
                        on = !(bound < rlb);
                        const T rx2 = bound * ((T)1 + (T)8 * Limits<T>::eps) - rlb;
                        const T rx = (rx2 > (T)0 ? sqrt(rx2) : (T)0) * ((T)1 + (T)8 * Limits<T>::eps);
                        xa = max(grid_cell(g, 0, q.x - rx) - 1, x0); xb = min(grid_cell(g, 0, q.x + rx) + 1, x1);
                        on = on && xa <= xb;
                    }
                    if (on) {
                        const int lo = row_run_lo(Gx, grid_row(Gy, cy, cz), xa, xb);
                        s = a.cell_start[lo]; e = a.cell_start[lo + (xb - xa + 1)];
                    }
                }
                const bool heavy = e - s > kHeavyRow;
                unsigned ls = s, le = e;                // this lane's share of the row
                if (sp == 2) { const unsigned mid = s + ((e - s + 1u) >> 1); if (lane & 1) ls = mid; else le = mid; }
                // light rows. K <= 32: four candidates per trip, their loads issued together (the pass is
                // latency-bound); slots past the share's end are killed (+inf / NaN d2 never enters the list)
                constexpr int kU = K <= 32 ? 4 : 1;            // (K = 64, 128: the lists fill the register file)
                for (unsigned p = ls; p < (heavy ? ls : le); p += kU) {
                    Pt4<T> cc[kU];
#pragma unroll
                    for (int u = 0; u < kU; ++u) cc[u] = rec_at(a.ref, a.ref_xyz, a.ref_idx, min(p + (unsigned)u, le - 1u));
#pragma unroll
                    for (int u = 0; u < kU; ++u) {
                        const Pt4<T>& c = cc[u];
                        const T dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
                        const T d = ((dx * dx) + (dy * dy)) + (dz * dz);
                        take(kill_if(d, (u > 0 && p + (unsigned)u >= le) || (!PCU_NO_BOUND_KILL && d > bound)), (int)c.idx);      // (beyond the bound: never among the k best, see above)
                    }
                }
                // heavy rows (a dense cluster next to the query): all 64 lanes stride over the row together, coalesced; every
                // lane keeps the best of its share, the rounds below merge the lanes' lists as for light rows
                // kH records per lane and trip, their loads issued together: one wave scanning a cell of 100k points is a chain of dependent
                // round trips (1560 of them at one record per trip: 0.48 ms for ONE query -- the whole launch on the tight-cluster cloud).
                // (Measured and rejected: walking a heavy row cell by cell and skipping cells beyond the lists' current bound -- no gain on
                // the cluster cloud, whose expensive queries sit INSIDE the heavy cell, and ruinous on long sparse rows: a lattice-vs-plane
                // pair of the randomised sweep went from milliseconds to 25 s.)
                unsigned long long hm = __ballot(heavy && (sp == 1 || !(lane & 1)));        // (once per row)
                while (hm) {
                    const int owner = __ffsll((long long)hm) - 1;
                    hm &= hm - 1;
                    const unsigned hs = (unsigned)__shfl((int)s, owner, 64), he = (unsigned)__shfl((int)e, owner, 64);
                    constexpr int kH = K <= 4 ? 16 : (K <= 32 ? 8 : 2);
                    for (unsigned p = hs + (unsigned)lane; p < he; p += 64u * kH) {
                        Pt4<T> hc[kH];
#pragma unroll
                        for (int u = 0; u < kH; ++u) hc[u] = rec_at(a.ref, a.ref_xyz, a.ref_idx, min(p + 64u * (unsigned)u, he - 1u));
#pragma unroll
                        for (int u = 0; u < kH; ++u) {
                            const Pt4<T>& c = hc[u];
                            const T dx = q.x - c.x, dy = q.y - c.y, dz = q.z - c.z;
                            const T d = ((dx * dx) + (dy * dy)) + (dz * dz);
                            take(kill_if(d, (u > 0 && p + 64u * (unsigned)u >= he) || (!PCU_NO_BOUND_KILL && d > bound)), (int)c.idx);
                        }
                    }
                }
                if (nbat > 1) {
                    T t = bd[K - 1];
#pragma unroll
                    for (int o = 32; o > 0; o >>= 1) { const T ot = __shfl_xor(t, o, 64); t = ot < t ? ot : t; }
                    bw = t < bw ? t : bw;
                }
            }
            // K rounds: global lexicographic minimum of the lanes' heads; the owner pops.
            my_d = Limits<T>::max_v; my_i = 0x7fffffff;     // lane j keeps rank j ...
            my_d2 = Limits<T>::max_v; my_i2 = 0x7fffffff;   // ... and rank j + 64 (K = 128)
            T prev_d = (T)-1; kth = Limits<T>::max_v; tie = false;
#pragma unroll 1
            for (int j = 0; j < K; ++j) {
                T md = bd[0]; int mi = bi[0];
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) {
                    const T od = __shfl_xor(md, o, 64); const int oi = __shfl_xor(mi, o, 64);
                    if (lex_less(od, oi, md, mi)) { md = od; mi = oi; }
                }
                if (bi[0] == mi && bd[0] == md && mi != 0x7fffffff) {          // owner (dataset rows are unique)
#pragma unroll
                    for (int i = 0; i < K - 1; ++i) { bd[i] = bd[i + 1]; bi[i] = bi[i + 1]; }
                    bd[K - 1] = Limits<T>::max_v; bi[K - 1] = 0x7fffffff;
                }
                if (lane == (j & 63)) { if (j < 64) { my_d = md; my_i = mi; } else { my_d2 = md; my_i2 = mi; } }
                // "Tied" here includes NEAR ties, within a few ulps of the distance: nanoflann prunes a branch by `mindistsq`, a sum it updates
                // incrementally in the input type (nanoflann.hpp:1601-1613), and when the terms are huge against the spacing of the candidates -- a
                // float32 query cloud at offset 1000 from a unit-size dataset: d2 ~ 3e6, ulp 0.25 -- its rounding discards the branch of the true
                // minimum: the reference then returns a neighbour one ulp WORSE than the minimum of its own distance arithmetic (141 of 2652
                // queries in the randomised sweep's seed 405 case 289). Such a query is not ours to answer by a minimum: it goes to the
                // reference's own traversal (kd_order.h) with the genuinely tied ones. Near ties between well-conditioned distances are as rare
                // as exact ones.
                if (j <= kreq && mi != 0x7fffffff && prev_d >= (T)0 && md - prev_d <= prev_d * ((T)8 * Limits<T>::eps)) tie = true;
                if (j == kreq - 1) kth = md;
                prev_d = md;
            }
            if (is_seed) {                                          // the sample's k-th best bounds the answer; fewer than k records in it: wider boxes as before
                seed_next = false;
                if (kth < Limits<T>::max_v) ball = kth; else R = min(4 * R, 4096);
                continue;
            }
            T kw = kth * ((T)1 + (T)8 * Limits<T>::eps);            // (the near-tie margin; none where it would overflow)
            if (!(kw <= Limits<T>::max_v)) kw = kth;
            certified = is_ball || kw < face_lower_bound(g, q.x, q.y, q.z, x0, x1, y0, y1, z0, z1);
            if (certified || !esc) break;
            if (kth < Limits<T>::max_v) ball = kth;                // k points seen: the ball round finishes the query
            else if (x0 == 0 && y0 == 0 && z0 == 0 && x1 == Gx - 1 && y1 == Gy - 1 && z1 == Gz - 1) break;      // (cannot happen on an open grid: its whole box certifies)
            else if (!seeded && !PCU_NO_SEED) { seed_next = true; seeded = true; }  // fewer than k points in the box: a bound from the subsample first
            else R = min(4 * R, 4096);                             // ... then a wider box (whole grid: certified)
        }
        if (certified && a.fuse != FUSE_NONE) {
            // fused epilogue (k = 1): the query's distance joins the direction's exact sum / this wave's arg-max; no row, no tie list
            const T v0 = (T)__shfl(a.squared ? my_d : sqrt(my_d), 0, 64);
            const int i0 = __shfl(my_i, 0, 64);
            if (a.fuse == FUSE_SUM) { if (lane == 0) exact_add(s_limbs[job1 ? 1 : 0], &s_special[job1 ? 1 : 0], (double)v0); }
            else {
                const long long key = ((long long)q.idx << 32) | (long long)((unsigned)i0 | (tie ? 0x80000000u : 0u));
                if (job1) argmax_combine(fbv1, fbk1, v0, key); else argmax_combine(fbv0, fbk0, v0, key);
            }
        } else if (certified) {
            if (lane < kreq) {
                const size_t o = (size_t)(a.row_out ? (int)q.idx : qpos) * (size_t)kreq + lane;
                const bool found = my_i != 0x7fffffff;
                a.out_i[o] = found ? (long long)my_i : -1ll;
                a.out_d[o] = found ? (a.squared ? my_d : sqrt(my_d)) : (T)-1;
            }
            if (K > 64 && lane + 64 < kreq) {
                const size_t o = (size_t)(a.row_out ? (int)q.idx : qpos) * (size_t)kreq + lane + 64;
                const bool found = my_i2 != 0x7fffffff;
                a.out_i[o] = found ? (long long)my_i2 : -1ll;
                a.out_d[o] = found ? (a.squared ? my_d2 : sqrt(my_d2)) : (T)-1;
            }
            if (tie && lane == 0) a.ties[atomicAdd(a.n_ties, 1)] = qpos;
        } else if (lane == 0) {
            a.unresolved[atomicAdd(a.n_unresolved, 1)] = qpos;
        }
    }
    // fused sum: the block's exact accumulators go to the call's, one atomic per non-zero limb and block. (One set of global atomics
    // per QUERY -- three words that every query of a cloud shares -- serialised at a single L2 channel: 33k stragglers of a Gaussian
    // cloud took 600 us.)
    if (a0.fuse == FUSE_SUM) {
        __syncthreads();
        for (int i = threadIdx.x; i < 2 * (kAccLimbs + 1); i += kBlock) {
            const int d = i / (kAccLimbs + 1), l = i % (kAccLimbs + 1);
            if (d == 1 && njobs < 2) continue;
            const SearchArgs<T>& a = d ? a1 : a0;
            if (l < kAccLimbs) { const unsigned long long v = s_limbs[d][l]; if (v) atomicAdd(&a.f_limbs[l], v); }
            else { const double v = s_special[d]; if (v != 0.0) atomicAdd(a.f_special, v); }
        }
    }
    // fused arg-max: one partial per wave and direction (every wave writes its slots, so the fold needs no counts)
    if (a0.fuse == FUSE_ARGMAX && lane == 0) {
        a0.f_wave_v[wave] = fbv0; a0.f_wave_k[wave] = fbk0;
        if (njobs > 1) { a1.f_wave_v[wave] = fbv1; a1.f_wave_k[wave] = fbk1; }
    }
}

// Restores the caller's row order: out[i, :] = res[pos_of[i], :], pos_of[i] = the slot the index build gave row i. One
// thread per output element: reads of pos_of and writes of out are coalesced; the cell-ordered result rows are gathered
// (they were just written and are L2/MALL resident). The main pass therefore writes full coalesced rows instead of
// scattering 4/8-byte values over the row-ordered arrays (which cost ~5x the algorithmic write traffic,
// profiles/r01_pmc.txt).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_unpermute(const unsigned* __restrict__ pos_of, const T* __restrict__ res_d,
                                                      const long long* __restrict__ res_i, T* __restrict__ out_d,
                                                      long long* __restrict__ out_i, long long n_elems, int k,
                                                      const int* __restrict__ result_block, int* host_block, unsigned seq, const int* __restrict__ giveup) {
    // (block 0's first wave also hands the call's result block -- the search counters, final by now -- to pinned host
    // memory, sequence word last: see k_pnorm_pair)
    if (host_block && blockIdx.x == 0 && threadIdx.x < 64) {
        if (threadIdx.x < 63) host_block[threadIdx.x] = result_block[threadIdx.x];
        __threadfence_system();
        if (threadIdx.x == 63) __hip_atomic_store(&host_block[63], (int)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= n_elems) return;
    // (giveup: the search's skew / unplaced-buckets flags. When the passes gave up there are no rows yet, and pos_of may be
    // incomplete -- the restore is repeated after the host has dealt with it)
    if (giveup && (giveup[0] | giveup[kLargeFlag])) return;
    const long long i = t / k; const int j = (int)(t - i * k);
    const size_t src = (size_t)pos_of[i] * (size_t)k + j;
    if (out_d) out_d[t] = res_d[src];
    if (out_i) out_i[t] = res_i[src];
}

// The call's result block to pinned host memory (sequence word last), for call shapes without an unpermute launch.
static __global__ void k_result_block_to_host(const int* __restrict__ result_block, int* host_block, unsigned seq) {
    if (threadIdx.x < 63) host_block[threadIdx.x] = result_block[threadIdx.x];
    __threadfence_system();
    if (threadIdx.x == 63) __hip_atomic_store(&host_block[63], (int)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}

// Same for the rows of a list of query slots only (after the tie-order resolver rewrote a handful of rows).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_unpermute_rows(const int* __restrict__ slots, int n_slots, const Pt4<T>* __restrict__ qsorted,
                                                           const T* __restrict__ res_d, const long long* __restrict__ res_i,
                                                           T* __restrict__ out_d, long long* __restrict__ out_i, int k) {
    const long long t = (long long)blockIdx.x * kBlock + threadIdx.x;
    if (t >= (long long)n_slots * k) return;
    const int s = slots[t / k]; const int j = (int)(t % k);
    const size_t src = (size_t)s * (size_t)k + j, dst = (size_t)qsorted[s].idx * (size_t)k + j;
    if (out_d) out_d[dst] = res_d[src];
    if (out_i) out_i[dst] = res_i[src];
}

}  // namespace pcu
