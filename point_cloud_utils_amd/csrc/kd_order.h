// csrc/kd_order.h -- order of exact distance ties, as the reference orders them.
//
// Exact ties (two dataset points at the same computed d2 from a query) are the one place where the result of
// pcu.k_nearest_neighbors depends on nanoflann's kd-tree and not only on the distance arithmetic: the
// reference keeps whichever tied point its depth-first traversal meets first (KNNResultSet::addPoint,
// nanoflann.hpp:194-227: strict '>' shift; leaf test `dist < worst_dist`, :1563), and that order is a property
// of the tree built by divideTree / middleSplit_ / planeSplit (:1001-1162) with `max_points_per_leaf`.
// The grid search (search.h) reports the queries that have a genuine tie inside their top-(k+1); for exactly
// those this file
//   1. rebuilds nanoflann's tree ON THE GPU, bit-faithfully (same bbox, same split-dimension rule, same cut
//      value arithmetic, and the same permutation that planeSplit's sequential two-cursor swaps produce), and
//   2. runs nanoflann's searchLevel recursion (:1544-1624), arithmetic and visiting order included, for each
//      such query, one lane per query, and overwrites that query's output row.
// The result for those rows is therefore what the reference computes by construction.
//
// Build = level-synchronous. Elements are {x,y,z,row} records permuted in place (so passes stream instead of
// gathering through vAcc). planeSplit's first loop pairs the j-th misplaced element from the left (>= cutval,
// left of lim1) with the j-th misplaced element from the right (< cutval, right of lim1) -- exactly the pairs
// the sequential cursors swap -- so the permutation is reproduced with two ranked lists per node and a
// parallel pairwise swap; the second loop (<= / >) is the same on [lim1, right). A node's work is cut into
// chunks of kKdChunk elements; per level a small set of kernels runs over all (node, chunk) work items.
#pragma once
#include "pcu_types.h"
#include "grid.h"

namespace pcu {

#ifndef PCU_KD_CHUNK
#define PCU_KD_CHUNK 4096
#endif
constexpr int kKdChunk = PCU_KD_CHUNK;  // elements per work item (tuning knob: 1024 / 2048 / 4096 -> 5.59 / 5.41 / 5.24 ms on the 4M k=16 config, the per-block locate chain amortises)
constexpr int kKdItems = kKdChunk / kBlock;

template <typename T>
struct KdNode {
    int left, right;                     // element range [left, right)
    int child1, child2;                  // node ids; -1 = leaf
    int divfeat;
    int active;                          // 1 = stub: a node outside every region of interest; it only gets its tight min/max (its
                                         // parent's divlow/divhigh need it) and is never split (see KdBuild::roi)
    T cutval;
    T bb_lo[3], bb_hi[3];                // the bbox handed DOWN to divideTree (input to middleSplit_)
    typename EncT<T>::type mm_lo[3], mm_hi[3];   // tight min/max of the node's points (encoded, atomics); = computeMinMax
    typename EncT<T>::type cmm_lo[2][3], cmm_hi[2][3];   // tight boxes of {< cutval} and {>= cutval}, gathered by the count pass: the children's
                                         // boxes when no element equals the cut value (then the children are exactly these two sets)
    int mm_ready;                        // mm_lo / mm_hi were installed by the parent's level (k_kd_advance): k_kd_minmax has nothing to do
    int lt, le;                          // # elements < cutval, <= cutval
    int nbad[2];                         // misplaced pairs in planeSplit loop 1 / loop 2
    int chunk_base, nchunks;
    int depth;
};

template <typename T>
struct KdBuild {
    Pt4<T>* E;                           // permuted elements
    KdNode<T>* nodes;
    int* n_nodes;                        // node-id allocator (upper bound of ids in use)
    int* n_real;                         // number of nodes actually created (diagnostics)
    int* n_cur;                          // number of nodes in level_nodes (device-resident: no host sync per level)
    int* level_nodes;                    // node ids of the current level
    int* next_nodes; int* n_next;        // node ids created for the next level
    int* level_cbase; int* next_cbase;   // exclusive prefix of chunk counts over the level's nodes (+ total)
    int* n_items;
    int* chunk_bl; int* chunk_br;        // per work item: misplaced-left / misplaced-right counts, then offsets
    int* BLpos; int* BRpos;              // ranked positions, indexed by node.left + rank
    int* sub_nodes; int* n_sub;          // nodes small enough to be finished inside one workgroup's LDS (k_kd_subtree)
    int* max_depth;
    int leaf_max;
    int sub_max;                         // nodes with <= sub_max elements go to sub_nodes
    long long* prof;                     // nullable: per-stage cycle counters of sub-tree block 0 (diagnostics)
    // Regions of interest (few tied queries): balls {x, y, z, R^2} around the tied queries, R = 3 x their k-th distance, i.e.
    // every tied candidate lies well inside. A child whose handed-down bbox misses all balls becomes a stub. The traversal
    // treats a stub as one big leaf: distances and the result *set* stay exact, only the order among exactly tied points
    // INSIDE a stub would be undefined -- and there are none, the tied points are inside the balls. *n_roi == 0: build all.
    const T* roi; const int* n_roi;
    int* need_ph2;                       // set when some node has elements equal to its cut value (planeSplit's second loop has work)
    int* level_ph2;                      // the same for the level in flight (reset by k_kd_advance): loop 2's three launches exit at once when it is 0
};

template <typename T>
__device__ __forceinline__ bool kd_in_roi(const KdBuild<T>& b, const T* lo, const T* hi) {
    const int n = *b.n_roi;
    if (n == 0) return true;
    for (int r = 0; r < n; ++r) {
        const T* q = b.roi + 4 * r;
        T d2 = 0;
        for (int j = 0; j < 3; ++j) { const T a = lo[j] - q[j], c = q[j] - hi[j]; const T m = a > c ? a : c; if (m > 0) d2 += m * m; }
        if (!(d2 > q[3])) return true;          // (NaN-safe: undecidable counts as inside)
    }
    return false;
}

// Coordinate d of element p, read straight from memory at a computed offset. (A `d == 0 ? x : d == 1 ? y : z`
// select chain on the wave-uniform d was miscompiled by hipcc 7.2 for gfx950 -- the z arm dereferenced an
// unset address register -- so no select chain here.)
template <typename T>
__device__ __forceinline__ T kd_coord(const Pt4<T>* E, int p, int d) { return reinterpret_cast<const T*>(E + p)[d]; }

// Regions of interest from the list of tied queries (cell-ordered result rows give the k-th distance).
constexpr int kKdMaxRoi = 64;
template <typename T>
__global__ void k_kd_roi(const Pt4<T>* __restrict__ qsorted, const int* __restrict__ qlist, int n_tied, const T* __restrict__ res_d, int k, int squared,
                         int row_out, T* __restrict__ roi, int* n_roi) {
    const int t = threadIdx.x;
    if (t == 0) *n_roi = n_tied;
    if (t >= n_tied) return;
    const int qpos = qlist[t];
    const Pt4<T> q = qsorted[qpos];
    const T dk = res_d[(size_t)(row_out ? (int)q.idx : qpos) * k + (k - 1)];
    T r2 = squared ? dk : dk * dk;
    r2 = dk < 0 ? (T)INFINITY : r2 * (T)9 * ((T)1 + (T)1e-3);            // R = 3 x the k-th distance (fewer than k found: everything)
    roi[4 * t] = q.x; roi[4 * t + 1] = q.y; roi[4 * t + 2] = q.z; roi[4 * t + 3] = r2;
}

// ---- element array + root node ----------------------------------------------------------------------------
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_init_elems(const T* __restrict__ pts, int n, Pt4<T>* __restrict__ E) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    Pt4<T> p; p.x = pts[3 * (size_t)i]; p.y = pts[3 * (size_t)i + 1]; p.z = pts[3 * (size_t)i + 2]; p.idx = i;
    E[i] = p;                              // vAcc[i] = i (nanoflann.hpp:1491-1499 init_vind)
}

template <typename T>
__device__ __forceinline__ void kd_node_init(KdNode<T>& nd, int left, int right) {
    nd.left = left; nd.right = right; nd.child1 = nd.child2 = -1; nd.divfeat = 0; nd.active = 0; nd.cutval = 0;
    for (int j = 0; j < 3; ++j) { nd.mm_lo[j] = ~(typename EncT<T>::type)0; nd.mm_hi[j] = 0; }
    for (int k = 0; k < 2; ++k) for (int j = 0; j < 3; ++j) { nd.cmm_lo[k][j] = ~(typename EncT<T>::type)0; nd.cmm_hi[k][j] = 0; }
    nd.mm_ready = 0;
    nd.lt = nd.le = 0; nd.nbad[0] = nd.nbad[1] = 0; nd.chunk_base = 0; nd.nchunks = 0; nd.depth = 0;
}

// Root: bbox = exact min/max of the data (computeBoundingBox, nanoflann.hpp:1501-1536), taken from the grid
// index's GridParams which holds the same exact bounds.
template <typename T>
__global__ void k_kd_root(KdBuild<T> b, const GridParams<T>* gp, int n) {
    if (threadIdx.x || blockIdx.x) return;
    KdNode<T>& nd = b.nodes[0];
    kd_node_init(nd, 0, n);
    // nanoflann's computeBoundingBox (nanoflann.hpp:1513-1541) runs over ALL points; the grid's box is that of the finite ones, and the raw
    // non-finite mask (grid.h: bits 1..3 = +inf, 4..6 = -inf on axis j) says where an infinity widens it (NaN never gets here: rejected)
    const unsigned raw = (unsigned)gp->nonfinite >> 8;
    for (int j = 0; j < 3; ++j) {
        nd.bb_lo[j] = (raw >> (4 + j)) & 1u ? -(T)INFINITY : gp->gmin[j];
        nd.bb_hi[j] = (raw >> (1 + j)) & 1u ? (T)INFINITY : gp->gmax[j];
    }
    *b.n_nodes = 1; *b.n_real = 1; *b.n_next = 0; *b.n_sub = 0; *b.max_depth = 0;
    if (n <= b.sub_max) { b.sub_nodes[0] = 0; *b.n_sub = 1; *b.n_items = 0; *b.n_cur = 0; }
    else { *b.n_cur = 1; b.level_nodes[0] = 0; b.level_cbase[0] = 0; b.level_cbase[1] = (n + kKdChunk - 1) / kKdChunk; *b.n_items = b.level_cbase[1]; }
}

// ---- per level (nodes too large for one workgroup): 9 launches -------------------------------------------------
// Work items are (node, chunk) pairs; a block finds its pair by a binary search in the level's chunk-prefix table.
template <typename T>
__device__ __forceinline__ bool kd_locate(const KdBuild<T>& b, int wi, int& node_id, int& chunk) {
    if (wi >= *b.n_items) return false;
    int lo = 0, hi = *b.n_cur;                         // level_cbase[lo] <= wi < level_cbase[hi]
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b.level_cbase[mid] <= wi) lo = mid; else hi = mid; }
    node_id = b.level_nodes[lo]; chunk = wi - b.level_cbase[lo];
    return true;
}

// K1: tight min/max (computeMinMax) of every node of the level
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_minmax(KdBuild<T> b) {
    int id, chunk;
    if (!kd_locate(b, blockIdx.x, id, chunk)) return;
    KdNode<T>& nd = b.nodes[id];
    if (nd.mm_ready) return;               // installed with the node (see k_kd_count)
    const int s = nd.left + chunk * kKdChunk, e = min(s + kKdChunk, nd.right);
    T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
    T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
    for (int p = s + threadIdx.x; p < e; p += kBlock) {
        const Pt4<T> v = b.E[p];
        lo[0] = v.x < lo[0] ? v.x : lo[0]; hi[0] = v.x > hi[0] ? v.x : hi[0];
        lo[1] = v.y < lo[1] ? v.y : lo[1]; hi[1] = v.y > hi[1] ? v.y : hi[1];
        lo[2] = v.z < lo[2] ? v.z : lo[2]; hi[2] = v.z > hi[2] ? v.z : hi[2];
    }
    __shared__ T s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        T a = wave_min(lo[j]), c = wave_max(hi[j]);
        if (lane == 0) { s_lo[wave][j] = a; s_hi[wave][j] = c; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        T a = s_lo[0][j], c = s_hi[0][j];
        for (int w = 1; w < kBlock / 64; ++w) { a = s_lo[w][j] < a ? s_lo[w][j] : a; c = s_hi[w][j] > c ? s_hi[w][j] : c; }
        if (a <= c) { atomicMin(&nd.mm_lo[j], enc(a)); atomicMax(&nd.mm_hi[j], enc(c)); }
    }
}

// middleSplit_ head (nanoflann.hpp:1061-1099): cut dimension and cut value from the hand-down box and the tight box.
template <typename T>
__device__ __forceinline__ void kd_choose(const T* bb_lo, const T* bb_hi, const T* mn, const T* mx, int& cutfeat_out, T& cutval_out) {
    const T EPS = (T)0.00001;
    T max_span = bb_hi[0] - bb_lo[0];
    for (int d = 1; d < 3; ++d) { const T span = bb_hi[d] - bb_lo[d]; if (span > max_span) max_span = span; }
    T max_spread = -1; int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
        const T span = bb_hi[d] - bb_lo[d];
        if (span > ((T)1 - EPS) * max_span) {
            const T spread = mx[d] - mn[d];
            if (spread > max_spread) { cutfeat = d; max_spread = spread; }
        }
    }
    const T split_val = (bb_lo[cutfeat] + bb_hi[cutfeat]) / (T)2;
    T cutval;
    if (split_val < mn[cutfeat]) cutval = mn[cutfeat]; else if (split_val > mx[cutfeat]) cutval = mx[cutfeat]; else cutval = split_val;
    cutfeat_out = cutfeat; cutval_out = cutval;
}

// K2: every block re-derives its node's cut (cheap scalar work, saves a launch), chunk 0 records it;
//     lim1 - left = #(< cutval), lim2 - left = #(<= cutval)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_count(KdBuild<T> b) {
    int id, chunk;
    if (!kd_locate(b, blockIdx.x, id, chunk)) return;
    KdNode<T>& nd = b.nodes[id];
    if (nd.active) return;                 // stub: min/max only
    T mn[3], mx[3];
    for (int j = 0; j < 3; ++j) { mn[j] = dec(nd.mm_lo[j]); mx[j] = dec(nd.mm_hi[j]); }
    int f; T cut;
    kd_choose(nd.bb_lo, nd.bb_hi, mn, mx, f, cut);
    if (chunk == 0 && threadIdx.x == 0) { nd.divfeat = f; nd.cutval = cut; }
    const int s = nd.left + chunk * kKdChunk, e = min(s + kKdChunk, nd.right);
    unsigned lt = 0, le = 0;
    // The same read also gives the tight boxes of the two value classes {< cut} and {>= cut}: when no element equals the cut value
    // (lt == le: every level of generic data) they ARE the two children (middleSplit_'s index is then lim1 = lim2 whatever count / 2
    // is), so the children are created with their boxes and the next level's k_kd_minmax pass -- a sixth of a level -- has nothing to do.
    T lo0[3], hi0[3], lo1[3], hi1[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) { lo0[j] = lo1[j] = Limits<T>::max_v; hi0[j] = hi1[j] = -Limits<T>::max_v; }
    const T big = Limits<T>::max_v;
    for (int p = s + threadIdx.x; p < e; p += kBlock) {
        const Pt4<T> r = b.E[p];
        const T v = kd_coord(b.E, p, f);
        const bool a = v < cut;
        lt += a; le += v <= cut;
        const T xa = a ? r.x : big, ya = a ? r.y : big, za = a ? r.z : big, xb = a ? big : r.x, yb = a ? big : r.y, zb = a ? big : r.z;
        lo0[0] = xa < lo0[0] ? xa : lo0[0]; lo0[1] = ya < lo0[1] ? ya : lo0[1]; lo0[2] = za < lo0[2] ? za : lo0[2];
        lo1[0] = xb < lo1[0] ? xb : lo1[0]; lo1[1] = yb < lo1[1] ? yb : lo1[1]; lo1[2] = zb < lo1[2] ? zb : lo1[2];
        const T xc = a ? r.x : -big, yc = a ? r.y : -big, zc = a ? r.z : -big, xd = a ? -big : r.x, yd = a ? -big : r.y, zd = a ? -big : r.z;
        hi0[0] = xc > hi0[0] ? xc : hi0[0]; hi0[1] = yc > hi0[1] ? yc : hi0[1]; hi0[2] = zc > hi0[2] ? zc : hi0[2];
        hi1[0] = xd > hi1[0] ? xd : hi1[0]; hi1[1] = yd > hi1[1] ? yd : hi1[1]; hi1[2] = zd > hi1[2] ? zd : hi1[2];
    }
    unsigned tl, te;
    block_exclusive_scan(lt, &tl); block_exclusive_scan(le, &te);
    if (threadIdx.x == 0) { if (tl) atomicAdd(&nd.lt, (int)tl); if (te) atomicAdd(&nd.le, (int)te); b.chunk_bl[blockIdx.x] = (int)tl; }   // (chunk_bl: # < cut of this work item, see k_kd_lists)
    __shared__ T s_mm[kBlock / 64][12];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const T a0 = wave_min(lo0[j]), c0 = wave_max(hi0[j]), a1 = wave_min(lo1[j]), c1 = wave_max(hi1[j]);
        if (lane == 0) { s_mm[wave][j] = a0; s_mm[wave][3 + j] = c0; s_mm[wave][6 + j] = a1; s_mm[wave][9 + j] = c1; }
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x / 3, j = threadIdx.x % 3;
        T a = s_mm[0][6 * k + j], c = s_mm[0][6 * k + 3 + j];
        for (int w = 1; w < kBlock / 64; ++w) { const T x = s_mm[w][6 * k + j], y = s_mm[w][6 * k + 3 + j]; a = x < a ? x : a; c = y > c ? y : c; }
        if (a <= c) { atomicMin(&nd.cmm_lo[k][j], enc(a)); atomicMax(&nd.cmm_hi[k][j], enc(c)); }      // (as k_kd_minmax folds a chunk; a class without elements here: a > c)
    }
}

// Misplaced flags of planeSplit loop PH (0: "< cutval" about lim1 on [left,right); 1: "<= cutval" about lim2 on [lim1,right)).
template <typename T>
__device__ __forceinline__ void kd_flags(const KdNode<T>& nd, int ph, int p, T v, bool& bad_left, bool& bad_right) {
    const int lo = ph == 0 ? nd.left : nd.left + nd.lt;
    const int lim = ph == 0 ? nd.left + nd.lt : nd.left + nd.le;
    const bool good = ph == 0 ? (v < nd.cutval) : (v <= nd.cutval);
    bad_left = p >= lo && p < lim && !good;
    bad_right = p >= lim && good;
}

// K3: per work item counts of misplaced-left / misplaced-right. Loop 2 has nothing to do when no element equals
// the cut value (lt == le): its three launches then exit at once.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_bad_count(KdBuild<T> b, int ph) {
    if (ph == 1 && !*b.level_ph2) return;
    int id, chunk;
    if (!kd_locate(b, blockIdx.x, id, chunk)) return;
    KdNode<T>& nd = b.nodes[id];
    const int wi = blockIdx.x;
    if (ph == 0 && chunk == 0 && threadIdx.x == 0 && !nd.active && nd.lt != nd.le) *b.need_ph2 = 1;
    if (nd.active || (ph == 1 && nd.lt == nd.le)) { if (threadIdx.x == 0) { b.chunk_bl[wi] = 0; b.chunk_br[wi] = 0; } return; }
    const int s = nd.left + chunk * kKdChunk, e = min(s + kKdChunk, nd.right);
    unsigned nl = 0, nr = 0;
    for (int p = s + threadIdx.x; p < e; p += kBlock) {
        bool bl, br; kd_flags(nd, ph, p, kd_coord(b.E, p, nd.divfeat), bl, br);
        nl += bl; nr += br;
    }
    unsigned tl, tr;
    block_exclusive_scan(nl, &tl); block_exclusive_scan(nr, &tr);
    if (threadIdx.x == 0) { b.chunk_bl[wi] = (int)tl; b.chunk_br[wi] = (int)tr; }
}

// K4: ranked position lists. A block first folds its node's chunk counts into its own offsets (misplaced-left
// ranks count from the node's left end, misplaced-right ranks from its right end). Thread t owns kKdItems
// consecutive positions so ranks follow position order.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_lists(KdBuild<T> b, int ph) {
    if (ph == 1 && !*b.level_ph2) return;
    int id, chunk;
    if (!kd_locate(b, blockIdx.x, id, chunk)) return;
    KdNode<T>& nd = b.nodes[id];
    if (nd.active) { if (chunk == 0 && threadIdx.x == 0) nd.nbad[ph] = 0; return; }
    if (ph == 0 && chunk == 0 && threadIdx.x == 0 && nd.lt != nd.le) { *b.need_ph2 = 1; *b.level_ph2 = 1; }
    if (ph == 1 && nd.lt == nd.le) { if (chunk == 0 && threadIdx.x == 0) nd.nbad[1] = 0; return; }
    const int wi = blockIdx.x, wi0 = wi - chunk, nc = (nd.right - nd.left + kKdChunk - 1) / kKdChunk;
    const int s = nd.left + chunk * kKdChunk, e = min(s + kKdChunk, nd.right);
    const int p0 = s + threadIdx.x * kKdItems;
    if (ph == 0) {
        // Loop 1 needs no count of the misplaced elements per work item (k_kd_bad_count): every ">= cut" element to the left of a
        // misplaced-left one is itself left of lim1, hence misplaced too -- so the rank of a misplaced-left element is the number of
        // ">= cut" elements before it, whatever lim1 is, and likewise the rank of a misplaced-right element is the number of "< cut"
        // elements after it. Both come from the "< cut" counts per work item that the count pass left in chunk_bl.
        unsigned before_ge = 0, after_lt = 0;
        for (int c = threadIdx.x; c < nc; c += kBlock) {
            const unsigned vlt = (unsigned)b.chunk_bl[wi0 + c];
            const unsigned size_c = (unsigned)(min(nd.left + (c + 1) * kKdChunk, nd.right) - (nd.left + c * kKdChunk));
            if (c < chunk) before_ge += size_c - vlt;
            if (c > chunk) after_lt += vlt;
        }
        unsigned t_bg, t_al;
        block_exclusive_scan(before_ge, &t_bg); block_exclusive_scan(after_lt, &t_al);
        const int lim = nd.left + nd.lt;
        const T cut = nd.cutval; const int f = nd.divfeat;
        bool ge[kKdItems], ltf[kKdItems];
        unsigned nge = 0, nlt = 0;
#pragma unroll
        for (int j = 0; j < kKdItems; ++j) {
            const int p = p0 + j;
            ge[j] = ltf[j] = false;
            if (p < e) { const bool a = kd_coord(b.E, p, f) < cut; ltf[j] = a; ge[j] = !a; }
            nge += ge[j]; nlt += ltf[j];
        }
        unsigned tg, tl2;
        unsigned eg = block_exclusive_scan(nge, &tg);
        unsigned el2 = block_exclusive_scan(nlt, &tl2);
        const int base_l = nd.left + (int)t_bg, base_r = nd.left + (int)t_al;
#pragma unroll
        for (int j = 0; j < kKdItems; ++j) {
            const int p = p0 + j;
            if (p == lim) nd.nbad[0] = (int)(t_bg + eg);                           // # ">= cut" before lim1 = the number of pairs (lim1 < right always)
            if (ge[j]) { if (p < lim) b.BLpos[base_l + eg] = p; ++eg; }
            if (ltf[j]) { if (p >= lim) b.BRpos[base_r + (tl2 - 1 - el2)] = p; ++el2; }   // rank from the right end of the chunk
        }
        return;
    }
    unsigned before_l = 0, after_r = 0, all_l = 0;
    for (int c = threadIdx.x; c < nc; c += kBlock) {
        const unsigned vl = (unsigned)b.chunk_bl[wi0 + c], vr = (unsigned)b.chunk_br[wi0 + c];
        all_l += vl; if (c < chunk) before_l += vl; if (c > chunk) after_r += vr;
    }
    unsigned t_bl, t_ar, t_all;
    block_exclusive_scan(before_l, &t_bl); block_exclusive_scan(after_r, &t_ar); block_exclusive_scan(all_l, &t_all);
    if (chunk == 0 && threadIdx.x == 0) nd.nbad[ph] = (int)t_all;
    bool bl[kKdItems], br[kKdItems];
    unsigned nl = 0, nr = 0;
#pragma unroll
    for (int j = 0; j < kKdItems; ++j) {
        const int p = p0 + j;
        bl[j] = br[j] = false;
        if (p < e) kd_flags(nd, ph, p, kd_coord(b.E, p, nd.divfeat), bl[j], br[j]);
        nl += bl[j]; nr += br[j];
    }
    unsigned tl, tr;
    unsigned el = block_exclusive_scan(nl, &tl);
    unsigned er = block_exclusive_scan(nr, &tr);
    const int base_l = nd.left + (int)t_bl;
    const int base_r = nd.left + (int)t_ar;
#pragma unroll
    for (int j = 0; j < kKdItems; ++j) {
        const int p = p0 + j;
        if (bl[j]) { b.BLpos[base_l + el] = p; ++el; }
        if (br[j]) { b.BRpos[base_r + (tr - 1 - er)] = p; ++er; }   // rank from the right end of the chunk
    }
}

// K5: swap the j-th misplaced-left with the j-th misplaced-right (std::swap in planeSplit, :1137 / :1155)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_swap(KdBuild<T> b, int ph) {
    if (ph == 1 && !*b.level_ph2) return;
    int id, chunk;
    if (!kd_locate(b, blockIdx.x, id, chunk)) return;
    KdNode<T>& nd = b.nodes[id];
    if (nd.active) return;
    const int j0 = chunk * kKdChunk, nb = nd.nbad[ph];
    for (int j = j0 + threadIdx.x; j < min(j0 + kKdChunk, nb); j += kBlock) {
        const int pl = b.BLpos[nd.left + j], pr = b.BRpos[nd.left + j];
        const Pt4<T> a = b.E[pl], c = b.E[pr];
        b.E[pl] = c; b.E[pr] = a;
    }
}

// K6 (one block): split index (middleSplit_ tail, :1104-1109), the two children with their hand-down boxes
// (:1040-1046), routing of each child (next level / LDS sub-tree list), and the next level's chunk-prefix table.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_advance(KdBuild<T> b) {
    const int n_level = *b.n_cur;
    __shared__ int s_next, s_items;
    if (threadIdx.x == 0) { s_next = 0; s_items = 0; }
    __syncthreads();
    for (int base = 0; base < n_level; base += kBlock) {
        const int i = base + threadIdx.x;
        if (i < n_level) {
            KdNode<T>& nd = b.nodes[b.level_nodes[i]];
            if (nd.active) continue;            // stub: its min/max is all that was wanted
            const int count = nd.right - nd.left, lim1 = nd.lt, lim2 = nd.le;
            int index;
            if (lim1 > count / 2) index = lim1; else if (lim2 < count / 2) index = lim2; else index = count / 2;
            const int c = atomicAdd(b.n_nodes, 2);
            atomicAdd(b.n_real, 2);
            KdNode<T>& l = b.nodes[c]; KdNode<T>& r = b.nodes[c + 1];
            kd_node_init(l, nd.left, nd.left + index);
            kd_node_init(r, nd.left + index, nd.right);
            for (int j = 0; j < 3; ++j) { l.bb_lo[j] = r.bb_lo[j] = nd.bb_lo[j]; l.bb_hi[j] = r.bb_hi[j] = nd.bb_hi[j]; }
            l.bb_hi[nd.divfeat] = nd.cutval;
            r.bb_lo[nd.divfeat] = nd.cutval;
            if (lim1 == lim2) {                 // no element equals the cut value: the children are the two value classes (index == lim1), boxes known
                for (int j = 0; j < 3; ++j) { l.mm_lo[j] = nd.cmm_lo[0][j]; l.mm_hi[j] = nd.cmm_hi[0][j]; r.mm_lo[j] = nd.cmm_lo[1][j]; r.mm_hi[j] = nd.cmm_hi[1][j]; }
                l.mm_ready = r.mm_ready = 1;
            }
            nd.child1 = c; nd.child2 = c + 1;
            l.depth = r.depth = nd.depth + 1;
            atomicMax(b.max_depth, nd.depth + 1);
            for (int k = 0; k < 2; ++k) {
                const int id2 = c + k; const int cnt = k ? count - index : index;
                KdNode<T>& ch = k ? r : l;
                if (cnt > b.leaf_max && !kd_in_roi(b, ch.bb_lo, ch.bb_hi)) {            // outside every region of interest: stub
                    ch.active = 1;
                    b.next_nodes[atomicAdd(&s_next, 1)] = id2;                            // one more level, for its min/max only
                } else if (cnt <= b.sub_max) b.sub_nodes[atomicAdd(b.n_sub, 1)] = id2;   // finished in LDS later
                else b.next_nodes[atomicAdd(&s_next, 1)] = id2;
            }
        }
    }
    __syncthreads();
    const int n_next = s_next;
    for (int base = 0; base < n_next; base += kBlock) {      // chunk-prefix table of the next level
        const int i = base + threadIdx.x;
        unsigned nc = 0;
        if (i < n_next) { const KdNode<T>& nd = b.nodes[b.next_nodes[i]]; nc = (unsigned)((nd.right - nd.left + kKdChunk - 1) / kKdChunk); }
        unsigned total;
        const unsigned ex = block_exclusive_scan(nc, &total);
        if (i < n_next) b.next_cbase[i] = s_items + (int)ex;
        __syncthreads();
        if (threadIdx.x == 0) s_items += (int)total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { b.next_cbase[n_next] = s_items; *b.n_items = s_items; *b.n_next = n_next; *b.level_ph2 = 0; }
}

// ---- one workgroup per node, elements in place ----------------------------------------------------------------------------------------
// Once the level passes have brought the nodes down to a few ten thousand elements, a level is ~9 launches of 3-5 us that each
// touch little data -- with regions of interest only a few dozen nodes are still alive and the chain of launches IS the cost (round
// 3, config 3: 0.55 ms for six such levels). From there on one workgroup takes a node of the level list and finishes everything
// below it that the level passes would: the same choose / count / ranked misplaced lists / pairwise swap (twice when elements equal
// the cut value) / split, with __syncthreads() for kernel boundaries and the elements in global memory (a node is L2-resident);
// children that fit the LDS sub-tree kernel go to its list, children outside every region of interest become stubs (their tight
// box is computed here, with the sibling's, in one pass over the parent), the others are pushed on the workgroup's own stack
// (larger child first, so the stack stays below log2 of the node). Nodes are independent, so nothing is synchronised across
// workgroups; node ids come from the global allocator (their order is not part of the tree).
constexpr int kFinThreads = 1024;
constexpr int kFinBatch = 8;

template <typename T>
__device__ __forceinline__ void kd_fin_minmax(const Pt4<T>* __restrict__ E, int s, int e, T (&lo)[3], T (&hi)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { lo[j] = Limits<T>::max_v; hi[j] = -Limits<T>::max_v; }
#pragma unroll 1
    for (int base = s + (int)threadIdx.x; base < e; base += kFinThreads * 4) {
        Pt4<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int p = base + u * kFinThreads; v[u] = E[p < e ? p : base]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            lo[0] = v[u].x < lo[0] ? v[u].x : lo[0]; hi[0] = v[u].x > hi[0] ? v[u].x : hi[0];
            lo[1] = v[u].y < lo[1] ? v[u].y : lo[1]; hi[1] = v[u].y > hi[1] ? v[u].y : hi[1];
            lo[2] = v[u].z < lo[2] ? v[u].z : lo[2]; hi[2] = v[u].z > hi[2] ? v[u].z : hi[2];
        }
    }
}

// tight boxes of [s, m) and [m, e) in one pass
template <typename T>
__device__ __forceinline__ void kd_fin_minmax2(const Pt4<T>* __restrict__ E, int s, int m, int e, T (&lo)[3], T (&hi)[3], T (&lo2)[3], T (&hi2)[3]) {
#pragma unroll
    for (int j = 0; j < 3; ++j) { lo[j] = lo2[j] = Limits<T>::max_v; hi[j] = hi2[j] = -Limits<T>::max_v; }
#pragma unroll 1
    for (int base = s + (int)threadIdx.x; base < e; base += kFinThreads * 4) {
        Pt4<T> v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { const int p = base + u * kFinThreads; v[u] = E[p < e ? p : base]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int p = base + u * kFinThreads;
            const bool a = (p < e ? p : base) < m;
            const T big = Limits<T>::max_v;
            // an element takes part in one box only: for the other it is replaced by the neutral value
            const T xa = a ? v[u].x : big, ya = a ? v[u].y : big, za = a ? v[u].z : big;
            const T xb = a ? big : v[u].x, yb = a ? big : v[u].y, zb = a ? big : v[u].z;
            lo[0] = xa < lo[0] ? xa : lo[0]; lo[1] = ya < lo[1] ? ya : lo[1]; lo[2] = za < lo[2] ? za : lo[2];
            lo2[0] = xb < lo2[0] ? xb : lo2[0]; lo2[1] = yb < lo2[1] ? yb : lo2[1]; lo2[2] = zb < lo2[2] ? zb : lo2[2];
            const T xc = a ? v[u].x : -big, yc = a ? v[u].y : -big, zc = a ? v[u].z : -big;
            const T xd = a ? -big : v[u].x, yd = a ? -big : v[u].y, zd = a ? -big : v[u].z;
            hi[0] = xc > hi[0] ? xc : hi[0]; hi[1] = yc > hi[1] ? yc : hi[1]; hi[2] = zc > hi[2] ? zc : hi[2];
            hi2[0] = xd > hi2[0] ? xd : hi2[0]; hi2[1] = yd > hi2[1] ? yd : hi2[1]; hi2[2] = zd > hi2[2] ? zd : hi2[2];
        }
    }
}

template <typename T>
__global__ __launch_bounds__(kFinThreads) void k_kd_finish(KdBuild<T> b) {
    typedef typename EncT<T>::type Enc;
    __shared__ int s_stack[64];
    __shared__ int s_i[8];                               // [0] lt [1] le [2] first child id [3] stack size [4] root is a stub
    __shared__ unsigned s_cnt[kFinThreads / 64][2];
    __shared__ T s_red[kFinThreads / 64][12];
    __shared__ T s_mm[12];                               // lo xyz, hi xyz of the first range; the same of the second
    __shared__ T s_roi[4 * kKdMaxRoi];
    __shared__ int s_in[2];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n_level = *b.n_cur;
    const int n_roi = *b.n_roi;
    for (int i = tid; i < 4 * n_roi; i += kFinThreads) s_roi[i] = b.roi[i];
    // box [lo, hi] against the regions: waves 0 and 1 test the two boxes handed in (kd_in_roi's arithmetic, one region per lane)
    auto in_roi2 = [&](const T (&box0)[6], const T (&box1)[6]) {
        if (wave < 2) {
            T bx[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) bx[j] = wave ? box1[j] : box0[j];
            bool hit = n_roi == 0;
            if (lane < n_roi) {
                const T* q = s_roi + 4 * lane;
                T d2 = 0;
#pragma unroll
                for (int j = 0; j < 3; ++j) { const T a = bx[j] - q[j], c = q[j] - bx[3 + j]; const T m = a > c ? a : c; if (m > 0) d2 += m * m; }
                hit = !(d2 > q[3]);
            }
            const bool any = __ballot(hit) != 0ull;
            if (lane == 0) s_in[wave] = any ? 1 : 0;
        }
    };
    auto fold_mm = [&](const T (&lo)[3], const T (&hi)[3], int slot) {     // per-wave part of a block min/max; slot 0 / 1 = first / second range
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const T a = wave_min(lo[j]), c = wave_max(hi[j]);
            if (lane == 0) { s_red[wave][6 * slot + j] = a; s_red[wave][6 * slot + 3 + j] = c; }
        }
    };
    auto finish_mm = [&](int n_slots) {                   // (after a barrier) threads 0 .. 6 n_slots - 1 fold the waves
        if (tid < 6 * n_slots) {
            const bool is_min = (tid % 6) < 3;
            T r = s_red[0][tid];
            for (int w = 1; w < kFinThreads / 64; ++w) { const T x = s_red[w][tid]; r = is_min ? (x < r ? x : r) : (x > r ? x : r); }
            s_mm[tid] = r;
        }
    };
#pragma unroll 1
    for (int li = blockIdx.x; li < n_level; li += gridDim.x) {
        if (tid == 0) s_i[3] = 0;
        __syncthreads();
        int id = b.level_nodes[li];
        bool from_level = true;                           // the level list's nodes do not have their tight box yet
#pragma unroll 1
        for (;;) {
            KdNode<T>& nd = b.nodes[id];
            const int left = nd.left, right = nd.right, count = right - left;
            T bb[6];
#pragma unroll
            for (int j = 0; j < 3; ++j) { bb[j] = nd.bb_lo[j]; bb[3 + j] = nd.bb_hi[j]; }
            if (from_level) {
                const bool ready = nd.mm_ready != 0;          // (installed by k_kd_advance with the node)
                if (!ready) {
                    T lo[3], hi[3];
                    kd_fin_minmax(b.E, left, right, lo, hi);
                    fold_mm(lo, hi, 0);
                }
                if (tid == 0) s_i[4] = nd.active;
                __syncthreads();
                if (!ready) finish_mm(1);
                else if (tid < 6) s_mm[tid] = tid < 3 ? dec(nd.mm_lo[tid]) : dec(nd.mm_hi[tid - 3]);
                in_roi2(bb, bb);
                __syncthreads();
                if (!ready && tid < 3 && s_mm[tid] <= s_mm[3 + tid]) { nd.mm_lo[tid] = enc(s_mm[tid]); nd.mm_hi[tid] = enc(s_mm[3 + tid]); }
                // a node handed over by complete (speculative) top levels is tested against the regions here
                const bool stub = s_i[4] || (count > b.leaf_max && !s_in[0]);
                if (stub) { if (tid == 0) nd.active = 1; break; }
            }
            T mn[3], mx[3];
            if (from_level) {
#pragma unroll
                for (int j = 0; j < 3; ++j) { mn[j] = s_mm[j]; mx[j] = s_mm[3 + j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 3; ++j) { mn[j] = dec(nd.mm_lo[j]); mx[j] = dec(nd.mm_hi[j]); }
            }
            int f; T cut;
            kd_choose(bb, bb + 3, mn, mx, f, cut);
            f = __builtin_amdgcn_readfirstlane(f);
            const T* const coord = reinterpret_cast<const T*>(b.E) + f;      // coordinate f of element p: coord[4 p]
            const int depth = nd.depth;
            __syncthreads();                              // (s_mm, s_in are free again)
            // lim1, lim2
            {
                unsigned lt = 0, le = 0;
#pragma unroll 1
                for (int base = left + tid; base < right; base += kFinThreads * kFinBatch) {
                    T v[kFinBatch];
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) { const int p = base + u * kFinThreads; v[u] = p < right ? coord[4 * (size_t)p] : Limits<T>::max_v; }
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) { const bool on = base + u * kFinThreads < right; lt += on && v[u] < cut; le += on && v[u] <= cut; }
                }
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { lt += __shfl_xor(lt, o, 64); le += __shfl_xor(le, o, 64); }
                if (lane == 0) { s_cnt[wave][0] = lt; s_cnt[wave][1] = le; }
                __syncthreads();
                if (tid < 2) { unsigned r = 0; for (int w = 0; w < kFinThreads / 64; ++w) r += s_cnt[w][tid]; s_i[tid] = (int)r; }
                __syncthreads();
            }
            const int lim1 = s_i[0], lim2 = s_i[1];
            // planeSplit: ranked misplaced lists + pairwise swap. Wave w owns the w-th sixteenth of the node's positions and walks it 64
            // consecutive positions at a time (coalesced), so ranks follow position order: rank = misplaced elements of earlier waves
            // + of this wave's earlier steps + of the lower lanes (ballot).
            const int wpt = (count + kFinThreads / 64 - 1) / (kFinThreads / 64);
            const int w0 = min(left + wave * wpt, right), w1 = min(w0 + wpt, right);
            const unsigned long long lower = (1ull << lane) - 1ull;
#pragma unroll 1
            for (int ph = 0; ph < 2; ++ph) {
                if (ph == 1 && lim1 == lim2) break;
                const int lo_p = ph == 0 ? left : left + lim1;
                const int lim = ph == 0 ? left + lim1 : left + lim2;
                auto flags = [&](int p, T v, bool& bl, bool& br) {
                    const bool good = ph == 0 ? (v < cut) : (v <= cut);
                    bl = p >= lo_p && p < lim && !good;
                    br = p >= lim && good;
                };
                unsigned nl = 0, nr = 0;                  // wave-uniform
#pragma unroll 1
                for (int base = w0; base < w1; base += 64 * kFinBatch) {
                    T v[kFinBatch];
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) { const int p = base + u * 64 + lane; v[u] = coord[4 * (size_t)min(p, w1 - 1)]; }
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) {
                        const int p = base + u * 64 + lane;
                        bool bl, br; flags(p, v[u], bl, br);
                        nl += (unsigned)__popcll(__ballot(p < w1 && bl)); nr += (unsigned)__popcll(__ballot(p < w1 && br));
                    }
                }
                if (lane == 0) { s_cnt[wave][0] = nl; s_cnt[wave][1] = nr; }
                __syncthreads();
                unsigned el = 0, er = 0, tl = 0, tr = 0;
#pragma unroll
                for (int w = 0; w < kFinThreads / 64; ++w) {
                    const unsigned a = s_cnt[w][0], c2 = s_cnt[w][1];
                    if (w < wave) { el += a; er += c2; }
                    tl += a; tr += c2;
                }
#pragma unroll 1
                for (int base = w0; base < w1; base += 64 * kFinBatch) {
                    T v[kFinBatch];
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) { const int p = base + u * 64 + lane; v[u] = coord[4 * (size_t)min(p, w1 - 1)]; }
#pragma unroll
                    for (int u = 0; u < kFinBatch; ++u) {
                        const int p = base + u * 64 + lane;
                        bool bl, br; flags(p, v[u], bl, br);
                        bl = bl && p < w1; br = br && p < w1;
                        const unsigned long long ml = __ballot(bl), mr = __ballot(br);
                        if (bl) b.BLpos[left + (int)(el + (unsigned)__popcll(ml & lower))] = p;
                        if (br) b.BRpos[left + (int)(tr - 1 - (er + (unsigned)__popcll(mr & lower)))] = p;      // rank from the right end
                        el += (unsigned)__popcll(ml); er += (unsigned)__popcll(mr);
                    }
                }
                __syncthreads();
                const int nbad = (int)tl;
#pragma unroll 1
                for (int base = tid; base < nbad; base += kFinThreads * 4) {
                    int pl[4], pr[4]; Pt4<T> a[4], c[4];
#pragma unroll
                    for (int u = 0; u < 4; ++u) { const int j = min(base + u * kFinThreads, nbad - 1); pl[u] = b.BLpos[left + j]; pr[u] = b.BRpos[left + j]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) { a[u] = b.E[pl[u]]; c[u] = b.E[pr[u]]; }
#pragma unroll
                    for (int u = 0; u < 4; ++u) if (base + u * kFinThreads < nbad) { b.E[pl[u]] = c[u]; b.E[pr[u]] = a[u]; }
                }
                __syncthreads();
            }
            // split index (middleSplit_ tail), the children, their tight boxes
            int index;
            if (lim1 > count / 2) index = lim1; else if (lim2 < count / 2) index = lim2; else index = count / 2;
            if (tid == 0) { s_i[2] = atomicAdd(b.n_nodes, 2); atomicAdd(b.n_real, 2); atomicMax(b.max_depth, depth + 1); }
            {
                T lo[3], hi[3], lo2[3], hi2[3];
                kd_fin_minmax2(b.E, left, left + index, right, lo, hi, lo2, hi2);
                fold_mm(lo, hi, 0); fold_mm(lo2, hi2, 1);
            }
            T cb0[6], cb1[6];
#pragma unroll
            for (int j = 0; j < 6; ++j) { cb0[j] = bb[j]; cb1[j] = bb[j]; }
#pragma unroll
            for (int j = 0; j < 3; ++j) if (j == f) { cb0[3 + j] = cut; cb1[j] = cut; }
            __syncthreads();
            finish_mm(2);
            in_roi2(cb0, cb1);
            __syncthreads();
            const int c = s_i[2];
            if (tid < 2) {
                KdNode<T>& ch = b.nodes[c + tid];
                kd_node_init(ch, tid ? left + index : left, tid ? right : left + index);
                for (int j = 0; j < 3; ++j) { ch.bb_lo[j] = tid ? cb1[j] : cb0[j]; ch.bb_hi[j] = tid ? cb1[3 + j] : cb0[3 + j]; }
                for (int j = 0; j < 3; ++j)
                    if (s_mm[6 * tid + j] <= s_mm[6 * tid + 3 + j]) { ch.mm_lo[j] = enc(s_mm[6 * tid + j]); ch.mm_hi[j] = enc(s_mm[6 * tid + 3 + j]); }
                ch.depth = depth + 1;
                const int cnt = tid ? count - index : index;
                if (cnt > b.leaf_max && !s_in[tid]) ch.active = 1;                                     // outside every region of interest: stub
                else if (cnt <= b.sub_max) b.sub_nodes[atomicAdd(b.n_sub, 1)] = c + tid;             // finished in LDS later
                else s_in[tid] = 2;                                                                    // goes on
            }
            if (tid == 0) { nd.divfeat = f; nd.cutval = cut; nd.lt = lim1; nd.le = lim2; nd.child1 = c; nd.child2 = c + 1; }
            __syncthreads();
            if (tid == 0) {
                int sp = s_i[3];
                const bool big_first = index >= count - index;          // the larger child is pushed first, the smaller is taken next
                for (int k = 0; k < 2; ++k) {
                    const int side = big_first ? k : 1 - k;
                    if (s_in[side] != 2) continue;
                    if (sp >= 64) __builtin_trap();                     // (cannot happen: the chain of "next" nodes at least halves)
                    s_stack[sp++] = c + side;
                }
                s_i[3] = sp;
            }
            __syncthreads();
            from_level = false;
            const int sp = s_i[3];
            if (sp == 0) break;
            id = s_stack[sp - 1];
            __syncthreads();
            if (tid == 0) s_i[3] = sp - 1;
            __syncthreads();
        }
    }
}

// ---- sub-trees in LDS -------------------------------------------------------------------------------------------------
// One workgroup takes a node of <= S elements, copies the elements into LDS and runs the *same* level-by-level
// algorithm there (choose / count / ranked misplaced lists / pairwise swap, twice / split / children boxes) with
// __syncthreads() in place of kernel boundaries, then writes the permuted elements back. This removes the many
// launches over ever smaller nodes that dominate a purely level-synchronous build (the last ~15 of ~25 levels).
template <typename T> struct KdSub;
// Sizes chosen so that two workgroups fit in a CU's 160 KB of LDS (~75 KB / ~63 KB each): with about M/S sub-trees in
// flight the second round of blocks that a 1-per-CU footprint would cause on 256 CUs is avoided.
template <> struct KdSub<float>  { static constexpr int S = 2048, CAP = 192; };
template <> struct KdSub<double> { static constexpr int S = 1024, CAP = 96; };
constexpr int kSubThreads = 512;

template <typename T>
__host__ __device__ constexpr size_t kd_sub_lds_bytes() {
    typedef typename EncT<T>::type Enc;
    return (size_t)KdSub<T>::S * sizeof(Pt4<T>) + 3 * (size_t)KdSub<T>::S * 2 + 13 * (size_t)KdSub<T>::CAP * 4 + (size_t)KdSub<T>::CAP * sizeof(T) +
           18 * (size_t)KdSub<T>::CAP * sizeof(Enc) + 2 * (size_t)KdSub<T>::CAP * 2 + 6 * (size_t)KdSub<T>::CAP * sizeof(T) + 256;
}

__device__ __forceinline__ void block_scan2(unsigned a, unsigned b2, unsigned& ea, unsigned& eb, unsigned& ta, unsigned& tb, unsigned* s_w /*34 words*/) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned ia = a, ib = b2;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { unsigned x = __shfl_up(ia, o, 64), y = __shfl_up(ib, o, 64); if (lane >= o) { ia += x; ib += y; } }
    if (lane == 63) { s_w[wave] = ia; s_w[16 + wave] = ib; }
    __syncthreads();
    if (threadIdx.x == 0) {
        unsigned ra = 0, rb = 0;
        for (int w = 0; w < kSubThreads / 64; ++w) { unsigned x = s_w[w], y = s_w[16 + w]; s_w[w] = ra; s_w[16 + w] = rb; ra += x; rb += y; }
        s_w[32] = ra; s_w[33] = rb;
    }
    __syncthreads();
    ea = ia - a + s_w[wave]; eb = ib - b2 + s_w[16 + wave]; ta = s_w[32]; tb = s_w[33];
    __syncthreads();
}

template <typename T>
__global__ __launch_bounds__(kSubThreads) void k_kd_subtree(KdBuild<T> b) {
    typedef typename EncT<T>::type Enc;
    constexpr int S = KdSub<T>::S, CAP = KdSub<T>::CAP, IPT = S / kSubThreads;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    Pt4<T>* E = reinterpret_cast<Pt4<T>*>(smem);
    unsigned short* seg = reinterpret_cast<unsigned short*>(E + S);
    unsigned short* BL = seg + S; unsigned short* BR = BL + S;
    int* n_gid = reinterpret_cast<int*>(BR + S);
    int* n_left = n_gid + CAP; int* n_right = n_left + CAP; int* n_feat = n_right + CAP; int* n_lt = n_feat + CAP; int* n_le = n_lt + CAP;
    int* n_pbl0 = n_le + CAP; int* n_pbl1 = n_pbl0 + CAP; int* n_pbr1 = n_pbl1 + CAP; int* n_idx = n_pbr1 + CAP;
    int* x_gid = n_idx + CAP; int* x_left = x_gid + CAP; int* x_right = x_left + CAP;   // next level's nodes
    T* n_cut = reinterpret_cast<T*>(x_right + CAP);
    Enc* cur_mm = reinterpret_cast<Enc*>(n_cut + CAP);                         // [CAP][6]
    Enc* c_mm = cur_mm + 6 * CAP;                                              // [2*CAP][6]
    unsigned short* child_slot = reinterpret_cast<unsigned short*>(c_mm + 12 * CAP);   // [2*CAP]
    unsigned* s_w = reinterpret_cast<unsigned*>(child_slot + 2 * CAP);        // 34 words scan scratch + misc
    int* s_misc = reinterpret_cast<int*>(s_w + 40);                            // [0]=n_next [1]=id base [2]=loop-2 needed [3]=nodes created
    T* n_bb = reinterpret_cast<T*>(s_misc + 8);                                // [CAP][6] hand-down boxes of the active nodes

    long long t_prev = b.prof ? wall_clock64() : 0;
    // (the launcher may not know the list's length: without a host read-back it launches a fixed grid that strides over the list)
    const int n_sub_nodes = *b.n_sub;
    for (int si = blockIdx.x; si < n_sub_nodes; si += gridDim.x) {
    const int root_gid = b.sub_nodes[si];
#define KD_PROF(slot) do { if (b.prof && blockIdx.x == 0 && threadIdx.x == 0) { long long t_now = wall_clock64(); b.prof[slot] += t_now - t_prev; t_prev = t_now; } } while (0)
    KdNode<T>& root = b.nodes[root_gid];
    const int g0 = root.left, n = root.right - root.left;
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += kSubThreads) { E[p] = b.E[g0 + p]; seg[p] = 0; }
    for (int p = n + tid; p < S; p += kSubThreads) seg[p] = 0xFFFF;
    if (tid < 6) cur_mm[tid] = tid < 3 ? ~(Enc)0 : (Enc)0;
    if (tid == 0) {
        n_gid[0] = root_gid; n_left[0] = 0; n_right[0] = n;
        for (int j = 0; j < 3; ++j) { n_bb[j] = root.bb_lo[j]; n_bb[3 + j] = root.bb_hi[j]; }
        // ids for the whole sub-tree (at most 2n-2 nodes below the sub-root) are reserved with ONE global atomic
        s_misc[1] = n > b.leaf_max ? atomicAdd(b.n_nodes, 2 * n) : 0; s_misc[3] = 0;
    }
    __syncthreads();
    const int root_depth = root.depth;
    int sub_level = 0;
    {   // tight box of the sub-root (computeMinMax); wave pre-reduction, then LDS atomics
        T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v}, hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
        for (int p = tid; p < n; p += kSubThreads) {
            const Pt4<T> v = E[p];
            lo[0] = v.x < lo[0] ? v.x : lo[0]; hi[0] = v.x > hi[0] ? v.x : hi[0];
            lo[1] = v.y < lo[1] ? v.y : lo[1]; hi[1] = v.y > hi[1] ? v.y : hi[1];
            lo[2] = v.z < lo[2] ? v.z : lo[2]; hi[2] = v.z > hi[2] ? v.z : hi[2];
        }
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const T a = wave_min(lo[j]), c = wave_max(hi[j]);
            if ((tid & 63) == 0 && a <= c) { atomicMin(&cur_mm[j], enc(a)); atomicMax(&cur_mm[3 + j], enc(c)); }
        }
    }
    __syncthreads();
    if (tid < 3) { root.mm_lo[tid] = cur_mm[tid]; root.mm_hi[tid] = cur_mm[3 + tid]; }
    int n_act = (n > b.leaf_max) ? 1 : 0;
    const int p0 = tid * IPT;                       // this thread's contiguous positions [p0, p0+IPT)

    KD_PROF(0);
    while (n_act > 0) {
        // S1: leaf test is implied (only nodes with count > leaf_max are active); middleSplit_ head
        if (tid < n_act) {
            int cutfeat; T cutval;
            T mn[3], mx[3];
            for (int d = 0; d < 3; ++d) { mn[d] = dec(cur_mm[6 * tid + d]); mx[d] = dec(cur_mm[6 * tid + 3 + d]); }
            kd_choose(n_bb + 6 * tid, n_bb + 6 * tid + 3, mn, mx, cutfeat, cutval);
            KdNode<T>& nd = b.nodes[n_gid[tid]];
            nd.divfeat = cutfeat; nd.cutval = cutval;
            n_feat[tid] = cutfeat; n_cut[tid] = cutval; n_lt[tid] = 0; n_le[tid] = 0;
        }
        if (tid == 0) { s_misc[2] = 0; s_misc[0] = 0; }
        __syncthreads();
        KD_PROF(1);
        // S2: lim1, lim2. Lanes hold consecutive positions, so a wave sees at most a few distinct nodes:
        // one ballot-popcount + one LDS atomic per distinct node instead of two atomics per element.
        for (int base = 0; base < n; base += kSubThreads) {
            const int p = base + tid;
            const int i = p < n ? seg[p] : 0xFFFF;
            bool flt = false, fle = false;
            if (i != 0xFFFF) { const T v = reinterpret_cast<const T*>(E + p)[n_feat[i]]; flt = v < n_cut[i]; fle = v <= n_cut[i]; }
            unsigned long long rem = __ballot(i != 0xFFFF);
            while (rem) {
                const int leader = __ffsll((long long)rem) - 1;
                const int key = __shfl(i, leader, 64);
                const unsigned long long m = __ballot(i == key);
                const unsigned long long mlt = __ballot(i == key && flt), mle = __ballot(i == key && fle);
                if ((tid & 63) == leader) { atomicAdd(&n_lt[key], __popcll(mlt)); atomicAdd(&n_le[key], __popcll(mle)); }
                rem &= ~m;
            }
        }
        __syncthreads();
        KD_PROF(2);
        if (tid < n_act && n_lt[tid] != n_le[tid]) s_misc[2] = 1;      // some element equals its node's cut value
        __syncthreads();
        const int n_ph = s_misc[2] ? 2 : 1;                             // planeSplit's second loop has nothing to move otherwise
        // S3/S4: ranked misplaced lists + pairwise swap (planeSplit loops 1 and 2)
        for (int ph = 0; ph < n_ph; ++ph) {
            bool bl[IPT], br[IPT];
            unsigned nl = 0, nr = 0;
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int p = p0 + j;
                bl[j] = br[j] = false;
                const int i = p < n ? seg[p] : 0xFFFF;
                if (i != 0xFFFF) {
                    const T v = reinterpret_cast<const T*>(E + p)[n_feat[i]];
                    const int lo = ph == 0 ? n_left[i] : n_left[i] + n_lt[i];
                    const int lim = ph == 0 ? n_left[i] + n_lt[i] : n_left[i] + n_le[i];
                    const bool good = ph == 0 ? (v < n_cut[i]) : (v <= n_cut[i]);
                    bl[j] = p >= lo && p < lim && !good;
                    br[j] = p >= lim && good;
                }
                nl += bl[j]; nr += br[j];
            }
            unsigned el, er, tl, tr;
            block_scan2(nl, nr, el, er, tl, tr, s_w);
            {   // record the prefixes at node boundaries
                unsigned rl = el, rr = er;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const int p = p0 + j;
                    const int i = p < n ? seg[p] : 0xFFFF;
                    if (i != 0xFFFF) {
                        if (p == n_left[i]) n_pbl0[i] = (int)rl;
                        if (p == n_right[i] - 1) { n_pbl1[i] = (int)(rl + bl[j]); n_pbr1[i] = (int)(rr + br[j]); }
                    }
                    rl += bl[j]; rr += br[j];
                }
            }
            __syncthreads();
            {
                unsigned rl = el, rr = er;
#pragma unroll
                for (int j = 0; j < IPT; ++j) {
                    const int p = p0 + j;
                    const int i = p < n ? seg[p] : 0xFFFF;
                    if (i != 0xFFFF) {
                        if (bl[j]) BL[n_left[i] + ((int)rl - n_pbl0[i])] = (unsigned short)p;
                        if (br[j]) BR[n_left[i] + (n_pbr1[i] - 1 - (int)rr)] = (unsigned short)p;
                    }
                    rl += bl[j]; rr += br[j];
                }
            }
            __syncthreads();
            for (int x = tid; x < n; x += kSubThreads) {
                const int i = seg[x];
                if (i == 0xFFFF) continue;
                const int j = x - n_left[i];
                if (j < n_pbl1[i] - n_pbl0[i]) {
                    const int pl = BL[n_left[i] + j], pr = BR[n_left[i] + j];
                    const Pt4<T> a = E[pl], c = E[pr];
                    E[pl] = c; E[pr] = a;
                }
            }
            __syncthreads();
        }
        KD_PROF(3);
        // S5: split index, children (ids come from the block's reserved range: no global round trip)
        T cbb[12];                                   // hand-down boxes of this thread's two children, installed in S7
        if (tid < n_act) {
            const int gid = n_gid[tid];
            KdNode<T>& nd = b.nodes[gid];
            const int count = n_right[tid] - n_left[tid], lim1 = n_lt[tid], lim2 = n_le[tid];
            int index;
            if (lim1 > count / 2) index = lim1; else if (lim2 < count / 2) index = lim2; else index = count / 2;
            n_idx[tid] = index;
            const int c = s_misc[1] + s_misc[3] + 2 * tid;
            KdNode<T>& l = b.nodes[c]; KdNode<T>& r = b.nodes[c + 1];
            kd_node_init(l, g0 + n_left[tid], g0 + n_left[tid] + index);
            kd_node_init(r, g0 + n_left[tid] + index, g0 + n_right[tid]);
            const int f = n_feat[tid]; const T cut = n_cut[tid];
            for (int j = 0; j < 6; ++j) { cbb[j] = n_bb[6 * tid + j]; cbb[6 + j] = n_bb[6 * tid + j]; }
            for (int j = 0; j < 3; ++j) if (j == f) { cbb[3 + j] = cut; cbb[6 + j] = cut; }
            for (int j = 0; j < 3; ++j) { l.bb_lo[j] = cbb[j]; l.bb_hi[j] = cbb[3 + j]; r.bb_lo[j] = cbb[6 + j]; r.bb_hi[j] = cbb[9 + j]; }
            l.depth = r.depth = root_depth + sub_level + 1;
            nd.child1 = c; nd.child2 = c + 1;
            for (int k = 0; k < 2; ++k) {
                const int cl = k ? n_left[tid] + index : n_left[tid], cr = k ? n_right[tid] : n_left[tid] + index;
                for (int j = 0; j < 3; ++j) { c_mm[6 * (2 * tid + k) + j] = ~(Enc)0; c_mm[6 * (2 * tid + k) + 3 + j] = 0; }
                if (cr - cl > b.leaf_max) {
                    const int slot = atomicAdd(&s_misc[0], 1);
                    x_gid[slot] = c + k; x_left[slot] = cl; x_right[slot] = cr;
                    child_slot[2 * tid + k] = (unsigned short)slot;
                } else child_slot[2 * tid + k] = 0xFFFF;
            }
        }
        __syncthreads();
        KD_PROF(4);
        // S6: children's tight boxes + re-label elements. Each thread folds its IPT consecutive positions, then the
        // wave runs a segmented min/max reduction over lanes (children are contiguous position ranges, so equal
        // keys are contiguous lanes); the first lane of each segment issues the LDS atomics.
        {
            int key = -1;                              // child (2*node + side) of the run this thread is accumulating
            T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v}, hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
            auto flush = [&](int k2) {
                Enc* mm = c_mm + 6 * k2;
                atomicMin(&mm[0], enc(lo[0])); atomicMax(&mm[3], enc(hi[0]));
                atomicMin(&mm[1], enc(lo[1])); atomicMax(&mm[4], enc(hi[1]));
                atomicMin(&mm[2], enc(lo[2])); atomicMax(&mm[5], enc(hi[2]));
            };
#pragma unroll
            for (int j = 0; j < IPT; ++j) {
                const int p = p0 + j;
                const int i = p < n ? seg[p] : 0xFFFF;
                if (i == 0xFFFF) continue;
                const int ck = 2 * i + (p >= n_left[i] + n_idx[i]);
                const Pt4<T> v = E[p];
                if (ck != key) {                       // a run ended inside this thread's positions: flush it directly
                    if (key >= 0) flush(key);
                    key = ck;
                    lo[0] = hi[0] = v.x; lo[1] = hi[1] = v.y; lo[2] = hi[2] = v.z;
                } else {
                    lo[0] = v.x < lo[0] ? v.x : lo[0]; hi[0] = v.x > hi[0] ? v.x : hi[0];
                    lo[1] = v.y < lo[1] ? v.y : lo[1]; hi[1] = v.y > hi[1] ? v.y : hi[1];
                    lo[2] = v.z < lo[2] ? v.z : lo[2]; hi[2] = v.z > hi[2] ? v.z : hi[2];
                }
                seg[p] = child_slot[ck];
            }
            // `key` is the LAST run of this thread; earlier runs were flushed. Lanes with the same key form
            // contiguous stretches except where a thread's first run differs from its last: treat the pair
            // (key of first run == key of last run) conservatively by requiring equality with the neighbour's key.
            const int lane = tid & 63;
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) {
                const int ok = __shfl_down(key, o, 64);
                T olo[3], ohi[3];
#pragma unroll
                for (int d = 0; d < 3; ++d) { olo[d] = __shfl_down(lo[d], o, 64); ohi[d] = __shfl_down(hi[d], o, 64); }
                if (lane + o < 64 && ok == key && key >= 0) {
#pragma unroll
                    for (int d = 0; d < 3; ++d) { lo[d] = olo[d] < lo[d] ? olo[d] : lo[d]; hi[d] = ohi[d] > hi[d] ? ohi[d] : hi[d]; }
                }
            }
            const int pk = __shfl_up(key, 1, 64);
            if (key >= 0 && (lane == 0 || pk != key)) flush(key);      // head of a segment holds its reduction
        }
        __syncthreads();
        KD_PROF(5);
        // S7: publish children's boxes; install the next level
        const int n_next = s_misc[0];
        for (int cidx = tid; cidx < 2 * n_act; cidx += kSubThreads) {
            KdNode<T>& ch = b.nodes[s_misc[1] + s_misc[3] + cidx];
            for (int j = 0; j < 3; ++j) { ch.mm_lo[j] = c_mm[6 * cidx + j]; ch.mm_hi[j] = c_mm[6 * cidx + 3 + j]; }
        }
        __syncthreads();
        for (int cidx = tid; cidx < 2 * n_act; cidx += kSubThreads) {
            const int slot = child_slot[cidx];
            if (slot != 0xFFFF) for (int j = 0; j < 6; ++j) cur_mm[6 * slot + j] = c_mm[6 * cidx + j];
        }
        int xg = 0, xl = 0, xr = 0;
        if (tid < n_next) { xg = x_gid[tid]; xl = x_left[tid]; xr = x_right[tid]; }
        __syncthreads();
        if (tid < n_next) { n_gid[tid] = xg; n_left[tid] = xl; n_right[tid] = xr; }
        if (tid < n_act) {
            for (int k = 0; k < 2; ++k) { const int slot = child_slot[2 * tid + k]; if (slot != 0xFFFF) for (int j = 0; j < 6; ++j) n_bb[6 * slot + j] = cbb[6 * k + j]; }
        }
        if (tid == 0) { s_misc[3] += 2 * n_act; atomicMax(b.max_depth, root_depth + sub_level + 1); }
        n_act = n_next; ++sub_level;
        __syncthreads();
        KD_PROF(6);
    }
    for (int p = tid; p < n; p += kSubThreads) b.E[g0 + p] = E[p];
    if (tid == 0 && s_misc[3]) atomicAdd(b.n_real, s_misc[3]);
    KD_PROF(7);
    __syncthreads();
    }
#undef KD_PROF
}

// ---- nanoflann search for the tied queries ---------------------------------------------------------------------------
template <typename T>
struct KdSearchArgs {
    const Pt4<T>* E; const KdNode<T>* nodes;
    const Pt4<T>* qsorted; const int* qlist; const int* qcount_dev;
    int k, squared, row_out;
    T* out_d; long long* out_i;
    void* stack; int stack_cap;         // per work item: stack_cap frames (tree depth + 2) in global memory
    const unsigned* cancel_word = nullptr; unsigned cancel_gen = 0;       // pcu_types.h: cancel_seen (looked at every 1024 traversal steps)
    int t0 = 0;                         // k_kd_search: first query of this launch (long query lists are enqueued in pieces, pcu_hip.hip: kd_search_launch)
    int* error_flag;
    // whole-cloud mode (k beyond the grid search's capacity, kd_search<T, true>): queries are the raw (nq, 3) rows, every block
    // strides over them, the result set lives in dynamic LDS (rs_d == nullptr) or in a per-block global scratch of k slots
    const T* qraw; int nq_raw;
    T* rs_d; int* rs_i;
};

template <typename T>
struct KdFrame { int node, other, idx, stage; T mindistsq, cut, dst; };

// One WAVE per query: findNeighbors / computeInitialDistances / searchLevel / addPoint verbatim in behaviour
// (nanoflann.hpp:1393-1418, :1164-1187, :1544-1624, :194-227). The control flow is wave-uniform (every lane runs the
// same scalar recursion, unrolled onto an explicit stack); at a leaf the lanes fetch the leaf's points together and
// evaluate their distances in parallel, then the points are offered to the result set one by one in vAcc order,
// exactly as the reference's loop does.
//   BIG = false: the tied queries of a grid search (k <= 128, result set in static LDS, lane 0 inserts serially).
//   BIG = true : every query of a call whose k exceeds the grid search's capacity -- this IS the reference's algorithm, so the
//                result is the reference's for any k (it accepts every k > 0, src/point_cloud_distance.cpp:133-135). The result
//                set is addressed through generic pointers (LDS or global) and an insertion is done by the whole wave: the
//                position is found by a uniform binary search, the tail is shifted 64 entries per step from the top down --
//                the same final array as addPoint's serial shift loop.
template <typename T, bool BIG>
__device__ __forceinline__ void kd_search_one(const KdSearchArgs<T>& a, const int t, const int slot, const Pt4<T> q, const size_t out_row,
                                              T* rd, int* ri, T* s_dists, T* s_vec) {
    const int lane = threadIdx.x;
    const T vec[3] = {q.x, q.y, q.z};
    const int k = a.k;
    int count = 0;
    if (lane == 0) rd[k - 1] = Limits<T>::max_v;                        // KNNResultSet::init (:176-183)
    __syncthreads();
    const KdNode<T>& root = a.nodes[0];
    T dists[3] = {0, 0, 0};
    T distsq = 0;
#pragma unroll
    for (int i = 0; i < 3; ++i) {                                       // computeInitialDistances on root_bbox
        const T lo = dec(root.mm_lo[i]), hi = dec(root.mm_hi[i]);
        if (vec[i] < lo) { dists[i] = (vec[i] - lo) * (vec[i] - lo); distsq += dists[i]; }
        if (vec[i] > hi) { dists[i] = (vec[i] - hi) * (vec[i] - hi); distsq += dists[i]; }
    }
    if (lane < 3) { s_dists[lane] = lane == 0 ? dists[0] : (lane == 1 ? dists[1] : dists[2]); s_vec[lane] = lane == 0 ? q.x : (lane == 1 ? q.y : q.z); }
    __syncthreads();
    typedef KdFrame<T> Frame;
    Frame* st = reinterpret_cast<Frame*>(a.stack) + (size_t)slot * a.stack_cap;     // recursion depth <= tree depth
    int sp = 0;
    Frame f; f.node = 0; f.stage = 0; f.mindistsq = distsq; f.other = 0; f.idx = 0; f.cut = 0; f.dst = 0;
    // `f` is the frame on top (kept in registers, wave-uniform); st[] holds the frames below it.
    unsigned steps = 0;
    long long t_poll = wall_clock64();
    while (true) {
        // the call has been abandoned: so is this traversal. (The request word lives in host memory -- a PCIe round trip per look: at most one per
        // 200 us of a wave's life, the clock read every 64 steps)
        if ((++steps & 63u) == 0u) { const long long t_now = wall_clock64(); if (t_now - t_poll > 20000ll) { t_poll = t_now; if (cancel_seen(a.cancel_word, a.cancel_gen)) return; } }
        const KdNode<T>& nd = a.nodes[f.node];
        bool pop = false;
        if (f.stage == 0) {
            if (nd.child1 < 0) {                                        // leaf (:1552-1572) -- or a stub, scanned like one
                if (nd.active && nd.right - nd.left > 4096) { if (lane == 0) *a.error_flag = 2; return; }   // big unbuilt part: ask for the whole tree
                const T worst_dist = rd[k - 1];                         // sampled once per leaf
                for (int base = nd.left; base < nd.right; base += 64) {
                    const int cnt = min(64, nd.right - base);
                    T d = 0; int id = 0;
                    if (lane < cnt) {
                        const Pt4<T> c = a.E[base + lane];
                        { const T diff = vec[0] - c.x; d += diff * diff; }
                        { const T diff = vec[1] - c.y; d += diff * diff; }
                        { const T diff = vec[2] - c.z; d += diff * diff; }
                        id = (int)c.idx;
                    }
                    for (int e = 0; e < cnt; ++e) {
                        const T de = __shfl(d, e, 64); const int ie = __shfl(id, e, 64);
                        if (de < worst_dist) {                          // addPoint (:194-227)
                            if (!BIG) {                                 // lane 0 owns the arrays
                                if (lane == 0) {
                                    int j;
                                    for (j = count; j > 0; --j) {
                                        if (rd[j - 1] > de) { if (j < k) { rd[j] = rd[j - 1]; ri[j] = ri[j - 1]; } }
                                        else break;
                                    }
                                    if (j < k) { rd[j] = de; ri[j] = ie; }
                                }
                            } else {
                                // where the serial loop stops: p = number of kept entries <= de (they form a prefix of the sorted array)
                                int lo = 0, hi = count;
                                while (lo < hi) { const int mid = (lo + hi) >> 1; if (rd[mid] <= de) lo = mid + 1; else hi = mid; }
                                const int p = lo, top = min(count, k - 1);           // entries [p, top) move up by one slot
                                for (int c = top; c > p; c -= 64) {
                                    const int j = c - 1 - lane;
                                    const bool on = j >= p;
                                    T v = 0; int vi = 0;
                                    if (on) { v = rd[j]; vi = ri[j]; }
                                    __syncthreads();                    // (one wave per block) every read of this step precedes its writes
                                    if (on) { rd[j + 1] = v; ri[j + 1] = vi; }
                                    __syncthreads();
                                }
                                if (lane == 0 && p < k) { rd[p] = de; ri[p] = ie; }
                                __syncthreads();
                            }
                            if (count < k) count++;
                        }
                    }
                    __syncthreads();
                }
                pop = true;
            } else {
                const int idx = nd.divfeat;
                const T val = s_vec[idx];
                const T divlow = dec(a.nodes[nd.child1].mm_hi[idx]);    // left_bbox[cutfeat].high after recursion (:1048)
                const T divhigh = dec(a.nodes[nd.child2].mm_lo[idx]);   // right_bbox[cutfeat].low (:1049)
                const T diff1 = val - divlow, diff2 = val - divhigh;
                int best;
                if ((diff1 + diff2) < 0) { best = nd.child1; f.other = nd.child2; f.cut = (val - divhigh) * (val - divhigh); }
                else { best = nd.child2; f.other = nd.child1; f.cut = (val - divlow) * (val - divlow); }
                f.idx = idx; f.stage = 1;
                if (sp + 1 >= a.stack_cap) { if (lane == 0) *a.error_flag = 1; return; }
                if (lane == 0) st[sp] = f;
                ++sp;
                const T m = f.mindistsq;
                f.node = best; f.stage = 0; f.mindistsq = m;
            }
        } else if (f.stage == 1) {
            f.dst = s_dists[f.idx];
            const T m2 = f.mindistsq + f.cut - f.dst;
            __syncthreads();
            if (lane == 0) s_dists[f.idx] = f.cut;
            __syncthreads();
            f.stage = 2;
            if (m2 * 1.0f <= rd[k - 1]) {
                if (sp + 1 >= a.stack_cap) { if (lane == 0) *a.error_flag = 1; return; }
                if (lane == 0) st[sp] = f;
                ++sp;
                f.node = f.other; f.stage = 0; f.mindistsq = m2;
            }
        } else {
            __syncthreads();
            if (lane == 0) s_dists[f.idx] = f.dst;
            __syncthreads();
            pop = true;
        }
        if (pop) {
            if (sp == 0) break;
            --sp;
            __syncthreads();
            f = st[sp];                                                  // written by lane 0 of this wave
        }
    }
    __syncthreads();
    const size_t o = out_row * (size_t)k;
    for (int j = lane; j < k; j += 64) {                                // src/point_cloud_distance.cpp:82-93
        if (j < count) { a.out_i[o + j] = ri[j]; a.out_d[o + j] = a.squared ? rd[j] : sqrt(rd[j]); }
        else { a.out_i[o + j] = -1; a.out_d[o + j] = (T)-1; }
    }
    (void)t;
}

template <typename T>
__global__ __launch_bounds__(64) void k_kd_search(const KdSearchArgs<T> a) {
    const int t = a.t0 + (int)blockIdx.x;
    if (t >= *a.qcount_dev) return;
    __shared__ T rd[128]; __shared__ int ri[128];                       // KNNResultSet storage (k <= 128)
    __shared__ T s_dists[3]; __shared__ T s_vec[3];                     // indexed by the split dimension: kept in LDS so
                                                                        // that no select chain on a uniform index is generated
    const Pt4<T> q = a.qsorted[a.qlist[t]];
    kd_search_one<T, false>(a, t, t, q, (size_t)(a.row_out ? (int)q.idx : a.qlist[t]), rd, ri, s_dists, s_vec);      // cell-ordered result rows unless row_out
}

// Whole-cloud mode: blocks (one wave each) stride over the raw query rows.
template <typename T>
__global__ __launch_bounds__(64) void k_kd_search_all(const KdSearchArgs<T> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char s_rs[];
    __shared__ T s_dists[3]; __shared__ T s_vec[3];
    T* rd; int* ri;
    if (a.rs_d) { rd = a.rs_d + (size_t)blockIdx.x * a.k; ri = a.rs_i + (size_t)blockIdx.x * a.k; }
    else { rd = reinterpret_cast<T*>(s_rs); ri = reinterpret_cast<int*>(s_rs + (size_t)a.k * sizeof(T)); }
    long long t_poll = wall_clock64();
    for (int t = blockIdx.x; t < a.nq_raw; t += gridDim.x) {
        Pt4<T> q; q.x = a.qraw[3 * (size_t)t]; q.y = a.qraw[3 * (size_t)t + 1]; q.z = a.qraw[3 * (size_t)t + 2]; q.idx = t;
        kd_search_one<T, true>(a, t, (int)blockIdx.x, q, (size_t)t, rd, ri, s_dists, s_vec);
        if (*(volatile int*)a.error_flag) return;
        { const long long t_now = wall_clock64(); if (t_now - t_poll > 20000ll) { t_poll = t_now; if (cancel_seen(a.cancel_word, a.cancel_gen)) return; } }
        __syncthreads();
    }
}

}  // namespace pcu
