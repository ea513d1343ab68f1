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

constexpr int kKdChunk = 1024;          // elements per work item (kBlock threads x 4)
constexpr int kKdItems = kKdChunk / kBlock;

template <typename T>
struct KdNode {
    int left, right;                     // element range [left, right)
    int child1, child2;                  // node ids; -1 = leaf
    int divfeat;
    int active;                          // 1 while the node still has to be split at the current level
    T cutval;
    T bb_lo[3], bb_hi[3];                // the bbox handed DOWN to divideTree (input to middleSplit_)
    typename EncT<T>::type mm_lo[3], mm_hi[3];   // tight min/max of the node's points (encoded, atomics); = computeMinMax
    int lt, le;                          // # elements < cutval, <= cutval
    int nbad[2];                         // misplaced pairs in planeSplit loop 1 / loop 2
    int chunk_base, nchunks;
    int depth;
};

template <typename T>
struct KdBuild {
    Pt4<T>* E;                           // permuted elements
    KdNode<T>* nodes;
    int* n_nodes;                        // device counter
    int* level_nodes;                    // node ids of the current level
    int* next_nodes; int* n_next;        // node ids created for the next level
    int* item_node; int* item_chunk;     // work items
    int* n_items;
    int* chunk_bl; int* chunk_br;        // per work item: misplaced-left / misplaced-right counts, then offsets
    int* BLpos; int* BRpos;              // ranked positions, indexed by node.left + rank
    int* sub_nodes; int* n_sub;          // nodes small enough to be finished inside one workgroup's LDS (k_kd_subtree)
    int* max_depth;
    int leaf_max;
    int sub_max;                         // nodes with <= sub_max elements go to sub_nodes
};

// Coordinate d of element p, read straight from memory at a computed offset. (A `d == 0 ? x : d == 1 ? y : z`
// select chain on the wave-uniform d was miscompiled by hipcc 7.2 for gfx950 -- the z arm dereferenced an
// unset address register -- so no select chain here.)
template <typename T>
__device__ __forceinline__ T kd_coord(const Pt4<T>* E, int p, int d) { return reinterpret_cast<const T*>(E + p)[d]; }

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
    nd.lt = nd.le = 0; nd.nbad[0] = nd.nbad[1] = 0; nd.chunk_base = 0; nd.nchunks = 0; nd.depth = 0;
}

// Root: bbox = exact min/max of the data (computeBoundingBox, nanoflann.hpp:1501-1536), taken from the grid
// index's GridParams which holds the same exact bounds.
template <typename T>
__global__ void k_kd_root(KdBuild<T> b, const GridParams<T>* gp, int n) {
    if (threadIdx.x || blockIdx.x) return;
    KdNode<T>& nd = b.nodes[0];
    kd_node_init(nd, 0, n);
    for (int j = 0; j < 3; ++j) { nd.bb_lo[j] = gp->gmin[j]; nd.bb_hi[j] = gp->gmax[j]; }
    *b.n_nodes = 1; *b.n_next = 0; *b.n_sub = 0; *b.max_depth = 0;
    if (n <= b.sub_max) { b.sub_nodes[0] = 0; *b.n_sub = 1; *b.n_items = 0; b.level_nodes[0] = -1; }
    else b.level_nodes[0] = 0;
}

// ---- per level: plan work items -------------------------------------------------------------------------------
// One block: chunk counts of the level's nodes -> exclusive scan -> (node, chunk) work-item table.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_plan(KdBuild<T> b, int n_level) {
    __shared__ unsigned s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int base = 0; base < n_level; base += kBlock) {
        const int i = base + threadIdx.x;
        unsigned nc = 0; int id = -1;
        if (i < n_level) { id = b.level_nodes[i]; const KdNode<T>& nd = b.nodes[id]; nc = (unsigned)((nd.right - nd.left + kKdChunk - 1) / kKdChunk); }
        unsigned total;
        const unsigned ex = block_exclusive_scan(nc, &total) + s_carry;
        if (id >= 0) {
            b.nodes[id].chunk_base = (int)ex; b.nodes[id].nchunks = (int)nc;
            for (unsigned c = 0; c < nc; ++c) { b.item_node[ex + c] = id; b.item_chunk[ex + c] = (int)c; }
        }
        __syncthreads();
        if (threadIdx.x == 0) s_carry += total;
        __syncthreads();
    }
    if (threadIdx.x == 0) { *b.n_items = (int)s_carry; *b.n_next = 0; }
}

// ---- K1: tight min/max of every node of the level (leaves included: their boxes give the parents' divlow/divhigh)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_minmax(KdBuild<T> b) {
    const int wi = blockIdx.x;
    if (wi >= *b.n_items) return;
    KdNode<T>& nd = b.nodes[b.item_node[wi]];
    const int s = nd.left + b.item_chunk[wi] * kKdChunk, e = min(s + kKdChunk, nd.right);
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

// ---- K2: leaf test + middleSplit_ head (nanoflann.hpp:1008, :1061-1099): cut dimension and cut value
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_choose(KdBuild<T> b, int n_level) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_level) return;
    KdNode<T>& nd = b.nodes[b.level_nodes[i]];
    const int count = nd.right - nd.left;
    if (count <= b.leaf_max) { nd.active = 0; return; }           // (right - left) <= leaf_max_size -> leaf
    const T EPS = (T)0.00001;
    T max_span = nd.bb_hi[0] - nd.bb_lo[0];
    for (int d = 1; d < 3; ++d) { const T span = nd.bb_hi[d] - nd.bb_lo[d]; if (span > max_span) max_span = span; }
    T max_spread = -1; int cutfeat = 0;
    for (int d = 0; d < 3; ++d) {
        const T span = nd.bb_hi[d] - nd.bb_lo[d];
        if (span > ((T)1 - EPS) * max_span) {
            const T spread = dec(nd.mm_hi[d]) - dec(nd.mm_lo[d]);
            if (spread > max_spread) { cutfeat = d; max_spread = spread; }
        }
    }
    const T split_val = (nd.bb_lo[cutfeat] + nd.bb_hi[cutfeat]) / (T)2;
    const T mn = dec(nd.mm_lo[cutfeat]), mx = dec(nd.mm_hi[cutfeat]);
    T cutval;
    if (split_val < mn) cutval = mn; else if (split_val > mx) cutval = mx; else cutval = split_val;
    nd.divfeat = cutfeat; nd.cutval = cutval; nd.active = 1; nd.lt = 0; nd.le = 0; nd.nbad[0] = nd.nbad[1] = 0;
}

// ---- K3: lim1 - left = #(< cutval), lim2 - left = #(<= cutval)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_count(KdBuild<T> b) {
    const int wi = blockIdx.x;
    if (wi >= *b.n_items) return;
    KdNode<T>& nd = b.nodes[b.item_node[wi]];
    if (!nd.active) return;
    const int s = nd.left + b.item_chunk[wi] * kKdChunk, e = min(s + kKdChunk, nd.right);
    const int f = nd.divfeat; const T cut = nd.cutval;
    unsigned lt = 0, le = 0;
    for (int p = s + threadIdx.x; p < e; p += kBlock) { const T v = kd_coord(b.E, p, f); lt += v < cut; le += v <= cut; }
    unsigned tl, te;
    block_exclusive_scan(lt, &tl); block_exclusive_scan(le, &te);
    if (threadIdx.x == 0) { if (tl) atomicAdd(&nd.lt, (int)tl); if (te) atomicAdd(&nd.le, (int)te); }
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

// ---- K4: per work item counts of misplaced-left / misplaced-right
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_bad_count(KdBuild<T> b, int ph) {
    const int wi = blockIdx.x;
    if (wi >= *b.n_items) return;
    KdNode<T>& nd = b.nodes[b.item_node[wi]];
    if (!nd.active) return;
    const int s = nd.left + b.item_chunk[wi] * kKdChunk, e = min(s + kKdChunk, nd.right);
    unsigned nl = 0, nr = 0;
    for (int p = s + threadIdx.x; p < e; p += kBlock) {
        bool bl, br; kd_flags(nd, ph, p, kd_coord(b.E, p, nd.divfeat), bl, br);
        nl += bl; nr += br;
    }
    unsigned tl, tr;
    block_exclusive_scan(nl, &tl); block_exclusive_scan(nr, &tr);
    if (threadIdx.x == 0) { b.chunk_bl[wi] = (int)tl; b.chunk_br[wi] = (int)tr; }
}

// ---- K5: per node (one block each), chunk offsets: misplaced-left ranks count from the left, misplaced-right ranks
// from the right end of the node
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_chunk_scan(KdBuild<T> b, int n_level, int ph) {
    if ((int)blockIdx.x >= n_level) return;
    KdNode<T>& nd = b.nodes[b.level_nodes[blockIdx.x]];
    if (!nd.active) return;
    int* cbl = b.chunk_bl + nd.chunk_base; int* cbr = b.chunk_br + nd.chunk_base;
    const int nc = nd.nchunks;
    unsigned tot_r = 0;
    for (int base = 0; base < nc; base += kBlock) {           // total of misplaced-right first
        const int c = base + threadIdx.x;
        unsigned t; block_exclusive_scan(c < nc ? (unsigned)cbr[c] : 0u, &t);
        tot_r += t;
    }
    unsigned carry_l = 0, carry_r = 0;
    for (int base = 0; base < nc; base += kBlock) {
        const int c = base + threadIdx.x;
        const unsigned vl = c < nc ? (unsigned)cbl[c] : 0u, vr = c < nc ? (unsigned)cbr[c] : 0u;
        unsigned tl, tr;
        const unsigned el = block_exclusive_scan(vl, &tl), er = block_exclusive_scan(vr, &tr);
        if (c < nc) { cbl[c] = (int)(carry_l + el); cbr[c] = (int)(tot_r - (carry_r + er) - vr); }   // # misplaced-right in chunks to the right
        carry_l += tl; carry_r += tr;
    }
    if (threadIdx.x == 0) nd.nbad[ph] = (int)carry_l;      // == tot_r: both cursors stop together
}

// ---- K6: ranked position lists. Thread t owns kKdItems consecutive positions so ranks follow position order.
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_lists(KdBuild<T> b, int ph) {
    const int wi = blockIdx.x;
    if (wi >= *b.n_items) return;
    KdNode<T>& nd = b.nodes[b.item_node[wi]];
    if (!nd.active) return;
    const int s = nd.left + b.item_chunk[wi] * kKdChunk, e = min(s + kKdChunk, nd.right);
    bool bl[kKdItems], br[kKdItems];
    unsigned nl = 0, nr = 0;
    const int p0 = s + threadIdx.x * kKdItems;
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
    const int base_l = nd.left + b.chunk_bl[wi];
    const int base_r = nd.left + b.chunk_br[wi];
#pragma unroll
    for (int j = 0; j < kKdItems; ++j) {
        const int p = p0 + j;
        if (bl[j]) { b.BLpos[base_l + el] = p; ++el; }
        if (br[j]) { b.BRpos[base_r + (tr - 1 - er)] = p; ++er; }   // rank from the right end of the chunk
    }
}

// ---- K7: swap the j-th misplaced-left with the j-th misplaced-right (std::swap in planeSplit, :1137 / :1155)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_swap(KdBuild<T> b, int ph) {
    const int wi = blockIdx.x;
    if (wi >= *b.n_items) return;
    KdNode<T>& nd = b.nodes[b.item_node[wi]];
    if (!nd.active) return;
    const int j0 = b.item_chunk[wi] * kKdChunk, nb = nd.nbad[ph];
    for (int j = j0 + threadIdx.x; j < min(j0 + kKdChunk, nb); j += kBlock) {
        const int pl = b.BLpos[nd.left + j], pr = b.BRpos[nd.left + j];
        const Pt4<T> a = b.E[pl], c = b.E[pr];
        b.E[pl] = c; b.E[pr] = a;
    }
}

// ---- K8: split index (middleSplit_ tail, :1104-1109) and the two children with their hand-down boxes (:1040-1046)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_kd_split(KdBuild<T> b, int n_level) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n_level) return;
    const int id = b.level_nodes[i];
    KdNode<T>& nd = b.nodes[id];
    if (!nd.active) return;
    const int count = nd.right - nd.left, lim1 = nd.lt, lim2 = nd.le;
    int index;
    if (lim1 > count / 2) index = lim1; else if (lim2 < count / 2) index = lim2; else index = count / 2;
    const int c = atomicAdd(b.n_nodes, 2);
    KdNode<T>& l = b.nodes[c]; KdNode<T>& r = b.nodes[c + 1];
    kd_node_init(l, nd.left, nd.left + index);
    kd_node_init(r, nd.left + index, nd.right);
    for (int j = 0; j < 3; ++j) { l.bb_lo[j] = r.bb_lo[j] = nd.bb_lo[j]; l.bb_hi[j] = r.bb_hi[j] = nd.bb_hi[j]; }
    l.bb_hi[nd.divfeat] = nd.cutval;
    r.bb_lo[nd.divfeat] = nd.cutval;
    nd.child1 = c; nd.child2 = c + 1; nd.active = 0;
    l.depth = r.depth = nd.depth + 1;
    atomicMax(b.max_depth, nd.depth + 1);
    for (int k = 0; k < 2; ++k) {
        const int id2 = c + k; const KdNode<T>& ch = k ? r : l;
        if (ch.right - ch.left <= b.sub_max) b.sub_nodes[atomicAdd(b.n_sub, 1)] = id2;     // finished in LDS later
        else b.next_nodes[atomicAdd(b.n_next, 1)] = id2;
    }
}

// ---- sub-trees in LDS -------------------------------------------------------------------------------------------------
// One workgroup takes a node of <= S elements, copies the elements into LDS and runs the *same* level-by-level
// algorithm there (choose / count / ranked misplaced lists / pairwise swap, twice / split / children boxes) with
// __syncthreads() in place of kernel boundaries, then writes the permuted elements back. This removes the many
// launches over ever smaller nodes that dominate a purely level-synchronous build (the last ~15 of ~25 levels).
template <typename T> struct KdSub;
template <> struct KdSub<float>  { static constexpr int S = 4096, CAP = 384; };
template <> struct KdSub<double> { static constexpr int S = 2048, CAP = 256; };
constexpr int kSubThreads = 512;

template <typename T>
__host__ __device__ constexpr size_t kd_sub_lds_bytes() {
    typedef typename EncT<T>::type Enc;
    return (size_t)KdSub<T>::S * sizeof(Pt4<T>) + 3 * (size_t)KdSub<T>::S * 2 + 13 * (size_t)KdSub<T>::CAP * 4 + (size_t)KdSub<T>::CAP * sizeof(T) +
           18 * (size_t)KdSub<T>::CAP * sizeof(Enc) + 2 * (size_t)KdSub<T>::CAP * 2 + 256;
}

__device__ __forceinline__ void block_scan2_512(unsigned a, unsigned b2, unsigned& ea, unsigned& eb, unsigned& ta, unsigned& tb, unsigned* s_w /*34 words*/) {
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
    int* s_misc = reinterpret_cast<int*>(s_w + 40);                            // [0]=n_next [1]=id base

    const int root_gid = b.sub_nodes[blockIdx.x];
    KdNode<T>& root = b.nodes[root_gid];
    const int g0 = root.left, n = root.right - root.left;
    const int tid = threadIdx.x;
    for (int p = tid; p < n; p += kSubThreads) { E[p] = b.E[g0 + p]; seg[p] = 0; }
    for (int p = n + tid; p < S; p += kSubThreads) seg[p] = 0xFFFF;
    if (tid < 6) cur_mm[tid] = tid < 3 ? ~(Enc)0 : (Enc)0;
    if (tid == 0) { n_gid[0] = root_gid; n_left[0] = 0; n_right[0] = n; }
    __syncthreads();
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

    while (n_act > 0) {
        // S1: leaf test is implied (only nodes with count > leaf_max are active); middleSplit_ head
        if (tid < n_act) {
            KdNode<T>& nd = b.nodes[n_gid[tid]];
            const T EPS = (T)0.00001;
            T max_span = nd.bb_hi[0] - nd.bb_lo[0];
            for (int d = 1; d < 3; ++d) { const T span = nd.bb_hi[d] - nd.bb_lo[d]; if (span > max_span) max_span = span; }
            T max_spread = -1; int cutfeat = 0;
            for (int d = 0; d < 3; ++d) {
                const T span = nd.bb_hi[d] - nd.bb_lo[d];
                if (span > ((T)1 - EPS) * max_span) {
                    const T spread = dec(cur_mm[6 * tid + 3 + d]) - dec(cur_mm[6 * tid + d]);
                    if (spread > max_spread) { cutfeat = d; max_spread = spread; }
                }
            }
            const T split_val = (nd.bb_lo[cutfeat] + nd.bb_hi[cutfeat]) / (T)2;
            const T mn = dec(cur_mm[6 * tid + cutfeat]), mx = dec(cur_mm[6 * tid + 3 + cutfeat]);
            T cutval;
            if (split_val < mn) cutval = mn; else if (split_val > mx) cutval = mx; else cutval = split_val;
            nd.divfeat = cutfeat; nd.cutval = cutval;
            n_feat[tid] = cutfeat; n_cut[tid] = cutval; n_lt[tid] = 0; n_le[tid] = 0;
        }
        if (tid == 0) s_misc[0] = 0;
        __syncthreads();
        // S2: lim1, lim2
        for (int p = tid; p < n; p += kSubThreads) {
            const int i = seg[p];
            if (i == 0xFFFF) continue;
            const T v = reinterpret_cast<const T*>(E + p)[n_feat[i]];
            if (v < n_cut[i]) atomicAdd(&n_lt[i], 1);
            if (v <= n_cut[i]) atomicAdd(&n_le[i], 1);
        }
        __syncthreads();
        // S3/S4 twice: ranked misplaced lists + pairwise swap (planeSplit loops 1 and 2)
        for (int ph = 0; ph < 2; ++ph) {
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
            block_scan2_512(nl, nr, el, er, tl, tr, s_w);
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
        // S5: split index, children
        if (tid == 0) s_misc[1] = atomicAdd(b.n_nodes, 2 * n_act);
        __syncthreads();
        if (tid < n_act) {
            const int gid = n_gid[tid];
            KdNode<T>& nd = b.nodes[gid];
            const int count = n_right[tid] - n_left[tid], lim1 = n_lt[tid], lim2 = n_le[tid];
            int index;
            if (lim1 > count / 2) index = lim1; else if (lim2 < count / 2) index = lim2; else index = count / 2;
            n_idx[tid] = index;
            const int c = s_misc[1] + 2 * tid;
            KdNode<T>& l = b.nodes[c]; KdNode<T>& r = b.nodes[c + 1];
            kd_node_init(l, g0 + n_left[tid], g0 + n_left[tid] + index);
            kd_node_init(r, g0 + n_left[tid] + index, g0 + n_right[tid]);
            for (int j = 0; j < 3; ++j) { l.bb_lo[j] = r.bb_lo[j] = nd.bb_lo[j]; l.bb_hi[j] = r.bb_hi[j] = nd.bb_hi[j]; }
            l.bb_hi[nd.divfeat] = nd.cutval; r.bb_lo[nd.divfeat] = nd.cutval;
            l.depth = r.depth = nd.depth + 1;
            nd.child1 = c; nd.child2 = c + 1;
            atomicMax(b.max_depth, nd.depth + 1);
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
        // S6: children's tight boxes + re-label elements
        for (int p = tid; p < n; p += kSubThreads) {
            const int i = seg[p];
            if (i == 0xFFFF) continue;
            const int k = p >= n_left[i] + n_idx[i];
            const Pt4<T> v = E[p];
            Enc* mm = c_mm + 6 * (2 * i + k);
            atomicMin(&mm[0], enc(v.x)); atomicMax(&mm[3], enc(v.x));
            atomicMin(&mm[1], enc(v.y)); atomicMax(&mm[4], enc(v.y));
            atomicMin(&mm[2], enc(v.z)); atomicMax(&mm[5], enc(v.z));
            seg[p] = child_slot[2 * i + k];
        }
        __syncthreads();
        // S7: publish children's boxes; install the next level
        const int n_next = s_misc[0];
        for (int cidx = tid; cidx < 2 * n_act; cidx += kSubThreads) {
            KdNode<T>& ch = b.nodes[s_misc[1] + cidx];
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
        n_act = n_next;
        __syncthreads();
    }
    for (int p = tid; p < n; p += kSubThreads) b.E[g0 + p] = E[p];
}

// ---- nanoflann search for the tied queries ---------------------------------------------------------------------------
template <typename T>
struct KdSearchArgs {
    const Pt4<T>* E; const KdNode<T>* nodes;
    const Pt4<T>* qsorted; const int* qlist; const int* qcount_dev;
    int k, squared;
    T* out_d; long long* out_i;
    T* scratch_d; int* scratch_i;       // k slots per work item (KNNResultSet storage)
    void* stack; int stack_cap;         // per work item: stack_cap frames (tree depth + 2) in global memory
    int* error_flag;
};

template <typename T>
struct KdFrame { int node, other, idx, stage; T mindistsq, cut, dst; };

// One lane per query: findNeighbors / computeInitialDistances / searchLevel / addPoint verbatim in behaviour
// (nanoflann.hpp:1393-1418, :1164-1187, :1544-1624, :194-227), recursion unrolled onto an explicit stack.
template <typename T>
__global__ __launch_bounds__(64) void k_kd_search(const KdSearchArgs<T> a) {
    const int t = blockIdx.x * 64 + threadIdx.x;
    if (t >= *a.qcount_dev) return;
    const Pt4<T> q = a.qsorted[a.qlist[t]];
    const T vec[3] = {q.x, q.y, q.z};
    const int k = a.k;
    T* rd = a.scratch_d + (size_t)t * k; int* ri = a.scratch_i + (size_t)t * k;
    int count = 0;
    rd[k - 1] = Limits<T>::max_v;                                       // KNNResultSet::init (:176-183)
    const KdNode<T>& root = a.nodes[0];
    T dists[3] = {0, 0, 0};
    T distsq = 0;
    for (int i = 0; i < 3; ++i) {                                       // computeInitialDistances on root_bbox
        const T lo = dec(root.mm_lo[i]), hi = dec(root.mm_hi[i]);
        if (vec[i] < lo) { dists[i] = (vec[i] - lo) * (vec[i] - lo); distsq += dists[i]; }
        if (vec[i] > hi) { dists[i] = (vec[i] - hi) * (vec[i] - hi); distsq += dists[i]; }
    }
    typedef KdFrame<T> Frame;
    Frame* st = reinterpret_cast<Frame*>(a.stack) + (size_t)t * a.stack_cap;     // recursion depth <= tree depth
    int sp = 0;
    st[0].node = 0; st[0].stage = 0; st[0].mindistsq = distsq;
    while (sp >= 0) {
        Frame& f = st[sp];
        const KdNode<T>& nd = a.nodes[f.node];
        if (f.stage == 0) {
            if (nd.child1 < 0) {                                        // leaf (:1552-1572)
                const T worst_dist = rd[k - 1];
                for (int i = nd.left; i < nd.right; ++i) {
                    const Pt4<T> c = a.E[i];
                    T d = 0;
                    { const T diff = vec[0] - c.x; d += diff * diff; }
                    { const T diff = vec[1] - c.y; d += diff * diff; }
                    { const T diff = vec[2] - c.z; d += diff * diff; }
                    if (d < worst_dist) {                               // addPoint (:194-227)
                        int j;
                        for (j = count; j > 0; --j) {
                            if (rd[j - 1] > d) { if (j < k) { rd[j] = rd[j - 1]; ri[j] = ri[j - 1]; } }
                            else break;
                        }
                        if (j < k) { rd[j] = d; ri[j] = (int)c.idx; }
                        if (count < k) count++;
                    }
                }
                --sp;
                continue;
            }
            const int idx = nd.divfeat;
            const T val = vec[idx];
            const T divlow = dec(a.nodes[nd.child1].mm_hi[idx]);        // left_bbox[cutfeat].high after recursion (:1048)
            const T divhigh = dec(a.nodes[nd.child2].mm_lo[idx]);       // right_bbox[cutfeat].low (:1049)
            const T diff1 = val - divlow, diff2 = val - divhigh;
            int best;
            if ((diff1 + diff2) < 0) { best = nd.child1; f.other = nd.child2; f.cut = (val - divhigh) * (val - divhigh); }
            else { best = nd.child2; f.other = nd.child1; f.cut = (val - divlow) * (val - divlow); }
            f.idx = idx; f.stage = 1;
            if (sp + 1 >= a.stack_cap) { *a.error_flag = 1; return; }
            ++sp; st[sp].node = best; st[sp].stage = 0; st[sp].mindistsq = f.mindistsq;
        } else if (f.stage == 1) {
            f.dst = dists[f.idx];
            const T m2 = f.mindistsq + f.cut - f.dst;
            dists[f.idx] = f.cut;
            f.stage = 2;
            if (m2 * 1.0f <= rd[k - 1]) {
                if (sp + 1 >= a.stack_cap) { *a.error_flag = 1; return; }
                const int other = f.other;
                ++sp; st[sp].node = other; st[sp].stage = 0; st[sp].mindistsq = m2;
            }
        } else {
            dists[f.idx] = f.dst;
            --sp;
        }
    }
    const size_t o = (size_t)q.idx * (size_t)k;
    for (int j = 0; j < k; ++j) {                                       // src/point_cloud_distance.cpp:82-93
        if (j < count) { a.out_i[o + j] = ri[j]; a.out_d[o + j] = a.squared ? rd[j] : sqrt(rd[j]); }
        else { a.out_i[o + j] = -1; a.out_d[o + j] = (T)-1; }
    }
}

}  // namespace pcu
