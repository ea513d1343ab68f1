// csrc/voxel.h -- grouping points by an integer key triple: voxel-grid downsampling and duplicate removal.
//
//   downsample_point_cloud_to_voxels   src/sample_point_cloud.cpp:163-235 (binding :336-368, wrapper __init__.py:123-200)
//   remove_duplicate_vertices          src/remove_duplicates.cpp:11-36 on libigl's round + unique_rows (binding :108-129)
// Both are "sort the points by a 3-component key, then work on the runs of equal keys". The reference does the first with a
// std::unordered_map (output in hash-table order, an artefact of libstdc++) and the second with libigl's sortrows. Here:
//   1. one kernel computes the key triple of every point with the reference's arithmetic (voxel index = int(floor((p - min) /
//      size)) in the point type; rounded coordinate = round(p / eps));
//   2. the triple is packed into one key of as many bits as the components' ranges need and the point ids are ordered by the repo's own STABLE
//      radix passes (radix.h; rounds 3-5: three rocPRIM radix sorts of full 32-bit keys): lexicographic by (k0, k1, k2), equal keys in input order;
//   3. run heads are flagged and scanned into run ids;
//   4. one thread per run adds its points IN INPUT ORDER in the point / attribute type -- exactly the sequence of additions
//      AccumulatedPoint::AddPoint performs (:119-128) -- so the voxel means are bit-identical to the reference's; duplicates
//      keep their first (lowest-index) point.
// Output order: ascending key (voxels: x-major lexicographic voxel index; duplicates: lexicographic rounded coordinates, which
// is libigl's unique_rows order). The reference's voxel output order is its hash table's and is not reproducible.
#pragma once
#include <cstring>
#include "pcu_types.h"
#include "grid.h"
#include "radix.h"

namespace pcu {

// voxel index of a point (:200-202): ref_coord = (p - min_bound) / voxel_size (element-wise, in T), int(floor(.))
template <typename T>
__global__ __launch_bounds__(kBlock) void k_voxel_keys(const T* __restrict__ pts, int n, T sx, T sy, T sz, T mx, T my, T mz,
                                                       int* __restrict__ k0, int* __restrict__ k1, int* __restrict__ k2, unsigned* __restrict__ ids) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const T x = pts[3 * (size_t)i], y = pts[3 * (size_t)i + 1], z = pts[3 * (size_t)i + 2];
    k0[i] = (int)floor((x - mx) / sx); k1[i] = (int)floor((y - my) / sy); k2[i] = (int)floor((z - mz) / sz);
    ids[i] = (unsigned)i;
}
// rounded coordinate of a point (remove_duplicates.cpp:27-28: igl::round(V / epsilon); epsilon <= 0: the coordinate itself),
// as an order-preserving unsigned key (-0 and +0 are the same number)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_round_keys(const T* __restrict__ pts, int n, T eps, typename EncT<T>::type* __restrict__ k0,
                                                       typename EncT<T>::type* __restrict__ k1, typename EncT<T>::type* __restrict__ k2, unsigned* __restrict__ ids) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    T v[3] = {pts[3 * (size_t)i], pts[3 * (size_t)i + 1], pts[3 * (size_t)i + 2]};
#pragma unroll
    for (int j = 0; j < 3; ++j) { if (eps > (T)0) v[j] = round(v[j] / eps); v[j] = v[j] + (T)0; }
    k0[i] = enc(v[0]); k1[i] = enc(v[1]); k2[i] = enc(v[2]);
    ids[i] = (unsigned)i;
}
template <typename K>
__global__ __launch_bounds__(kBlock) void k_gather_keys(const K* __restrict__ src, const unsigned* __restrict__ perm, int n, K* __restrict__ dst) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) dst[i] = src[perm[i]];
}
// flag[j] = 1 where the j-th point in sorted order starts a new run of equal keys
template <typename K>
__global__ __launch_bounds__(kBlock) void k_run_heads(const K* __restrict__ k0, const K* __restrict__ k1, const K* __restrict__ k2,
                                                      const unsigned* __restrict__ perm, int n, unsigned* __restrict__ flag) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    if (j == 0) { flag[0] = 1u; return; }
    const unsigned a = perm[j], b = perm[j - 1];
    flag[j] = (k0[a] != k0[b] || k1[a] != k1[b] || k2[a] != k2[b]) ? 1u : 0u;
}
// run starts from the inclusive scan of the head flags: run r = scan[j] - 1 begins at the j with flag[j] set; start[runs] = n
__global__ __launch_bounds__(kBlock) void k_run_starts(const unsigned* __restrict__ flag, const unsigned* __restrict__ scan, int n, unsigned* __restrict__ start) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    if (flag[j]) start[scan[j] - 1] = (unsigned)j;
    if (j == n - 1) start[scan[j]] = (unsigned)n;
}
// keep[r] = run r holds at least min_pts points (:217-219)
__global__ __launch_bounds__(kBlock) void k_run_keep(const unsigned* __restrict__ start, const unsigned* __restrict__ n_runs_dev, int min_pts, unsigned* __restrict__ keep) {
    const unsigned r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= *n_runs_dev) return;
    keep[r] = (int)(start[r + 1] - start[r]) >= min_pts ? 1u : 0u;
}
// One thread per voxel: the mean of its points (and of one attribute row each), added in input order in the input types
// (AccumulatedPoint::AddPoint / GetAveragePoint / GetAverageAttrib, :119-137). Kept voxels are written compacted.
template <typename T, typename A>
__global__ __launch_bounds__(kBlock) void k_voxel_means(const T* __restrict__ pts, const A* __restrict__ attrib, int cols, const unsigned* __restrict__ perm,
                                                        const unsigned* __restrict__ start, const unsigned* __restrict__ n_runs_dev, const unsigned* __restrict__ keep,
                                                        const unsigned* __restrict__ keep_scan, T* __restrict__ out_v, A* __restrict__ out_a) {
    const unsigned r = blockIdx.x * kBlock + threadIdx.x;
    if (r >= *n_runs_dev || !keep[r]) return;
    const unsigned s = start[r], e = start[r + 1], o = keep_scan[r] - 1u;
    T px = 0, py = 0, pz = 0;
    for (unsigned j = s; j < e; ++j) { const size_t i = perm[j]; px += pts[3 * i]; py += pts[3 * i + 1]; pz += pts[3 * i + 2]; }
    const T cnt = (T)(double)(e - s);                    // point_ / double(num_of_points_): Eigen brings the scalar to the matrix type
    out_v[3 * (size_t)o] = px / cnt; out_v[3 * (size_t)o + 1] = py / cnt; out_v[3 * (size_t)o + 2] = pz / cnt;
    for (int c = 0; c < cols; ++c) {
        A acc = 0;
        for (unsigned j = s; j < e; ++j) acc += attrib[(size_t)perm[j] * cols + c];
        out_a[(size_t)o * cols + c] = acc / (A)(e - s);
    }
}
// Duplicate removal: run r's representative is its first point in input order; svi[r] = that row, svj[i] = run of row i,
// out[r] = pts[svi[r]]  (SV = V(SVI, :), remove_duplicates.cpp:29)
template <typename T>
__global__ __launch_bounds__(kBlock) void k_dedup_write(const T* __restrict__ pts, const unsigned* __restrict__ perm, const unsigned* __restrict__ flag,
                                                        const unsigned* __restrict__ scan, int n, T* __restrict__ out, int* __restrict__ svi, int* __restrict__ svj) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j >= n) return;
    const unsigned i = perm[j], r = scan[j] - 1u;
    svj[i] = (int)r;
    if (flag[j]) {
        svi[r] = (int)i;
        out[3 * (size_t)r] = pts[3 * (size_t)i]; out[3 * (size_t)r + 1] = pts[3 * (size_t)i + 1]; out[3 * (size_t)r + 2] = pts[3 * (size_t)i + 2];
    }
}

}  // namespace pcu
