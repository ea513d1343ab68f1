// csrc/pcu_types.h -- device-side data layout shared by the grid-index and search kernels.
//
// HBM layout of one "grid index" (uniform grid over a point cloud, one per cloud per call):
//   GridParams<T>           1 struct   bbox, cell edge, cell counts (written by k_make_grid, read by everyone
//                                      through the scalar cache: it is wave-uniform)
//   cell_start[ncells+1]    uint32     exclusive prefix sum of per-cell point counts, x-fastest cell order
//   sorted[n]               Pt4<T>     the cloud permuted into cell order, AoS {x,y,z,original row}: one
//                                      16-byte (f32) / 32-byte (f64) record per point so that a lane fetches a
//                                      whole candidate with a single dwordx4 (two for f64) load
// Scratch while building: cell_of[n] (uint32), rank[n] (uint32).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <float.h>

namespace pcu {

template <typename T> struct Pt4;
template <> struct alignas(16) Pt4<float>  { float x, y, z; int idx; };
template <> struct alignas(32) Pt4<double> { double x, y, z; long long idx; };

// Coordinates-only copy of a cell-ordered cloud (3 T per record, no row id): the candidate stream of k_search1_flat (search.h,
// PCU_FLAT_XYZ). It lives right behind the n + 8 Pt4 records (incl. the 8 +inf sentinels) of the same allocation and has 8 sentinel
// records of its own. Behind it: the row ids of the same records as 32-bit integers (idx32_of; n + 8 of them). Every index build writes
// both streams; a LEAN build (grid2.h: fused k = 1 calls, which never look at a Pt4 record) writes only them and leaves the Pt4 array
// unwritten until some other kernel needs it (k_make_pt4).
#ifndef PCU_FLAT_XYZ
#define PCU_FLAT_XYZ 1
#endif
template <typename T> __host__ __device__ __forceinline__ T* xyz_of(Pt4<T>* sorted, int n) { return reinterpret_cast<T*>(sorted + n + 8); }
template <typename T> __host__ __device__ __forceinline__ const T* xyz_of(const Pt4<T>* sorted, int n) { return reinterpret_cast<const T*>(sorted + n + 8); }
template <typename T> __host__ __device__ __forceinline__ int* idx32_of(Pt4<T>* sorted, int n) { return reinterpret_cast<int*>(xyz_of(sorted, n) + 3 * (size_t)(n + 8)); }
template <typename T> __host__ __device__ __forceinline__ const int* idx32_of(const Pt4<T>* sorted, int n) { return reinterpret_cast<const int*>(xyz_of(sorted, n) + 3 * (size_t)(n + 8)); }
template <typename T> __device__ __forceinline__ void put_xyz(Pt4<T>* sorted, int n, unsigned pos, const Pt4<T>& r) {
#if PCU_FLAT_XYZ
    T* o = xyz_of(sorted, n) + 3 * (size_t)pos;
    o[0] = r.x; o[1] = r.y; o[2] = r.z;
    idx32_of(sorted, n)[pos] = (int)r.idx;
#endif
}
template <typename T> __device__ __forceinline__ void put_sentinels(Pt4<T>* sorted, int n) {       // records n..n+7 of all streams: see k_search / k_search1_flat
    for (int j = 0; j < 8; ++j) {
        Pt4<T> s; s.x = s.y = s.z = (T)INFINITY; s.idx = 0x7fffffff;
        sorted[n + j] = s;
        put_xyz(sorted, n, (unsigned)(n + j), s);
    }
}
constexpr size_t sorted_records_bytes(size_t n, size_t rec_bytes, size_t scalar_bytes) { return (n + 8) * rec_bytes + (PCU_FLAT_XYZ ? (n + 8) * (3 * scalar_bytes + 4) : 0); }

// Cancellation inside long-running kernels (pcu_hip.hip: g_cancel_mirror): `word` is a pinned host word holding the process's request counter,
// `gen` its value when the kernel's call began. True once a request NEWER than the call has been filed. One uncached system-scope load (a PCIe
// round trip): callers look every ~1000 steps, wave-uniformly.
__device__ __forceinline__ bool cancel_seen(const unsigned* word, unsigned gen) {
    return word != nullptr && (int)(__hip_atomic_load(word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - gen) > 0;
}

template <typename T> struct Limits;
template <> struct Limits<float>  { static constexpr float  max_v = FLT_MAX; static constexpr float  eps = FLT_EPSILON; };
template <> struct Limits<double> { static constexpr double max_v = DBL_MAX; static constexpr double eps = DBL_EPSILON; };

// Order-preserving float -> unsigned encoding, so that bbox min/max can use integer atomics.
__device__ __forceinline__ unsigned int enc(float f) {
    unsigned int u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec(unsigned int u) {
    return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u);
}
__device__ __forceinline__ unsigned long long enc(double f) {
    unsigned long long u = (unsigned long long)__double_as_longlong(f);
    return (u & 0x8000000000000000ull) ? ~u : (u | 0x8000000000000000ull);
}
__device__ __forceinline__ double dec(unsigned long long u) {
    return __longlong_as_double((long long)((u & 0x8000000000000000ull) ? (u & 0x7fffffffffffffffull) : ~u));
}
template <typename T> struct EncT;
template <> struct EncT<float>  { using type = unsigned int; };
template <> struct EncT<double> { using type = unsigned long long; };

template <typename T>
struct GridParams {
    T gmin[3], gmax[3];       // exact data bounding box
    T h, inv_h;               // cell edge (same in x,y,z) and fl(1/h)
    T slack[3];               // conservative slack on cell-face positions (certification, see search.h)
    int G[3];                 // cells per axis
    int ncells;               // G[0]*G[1]*G[2]
    T org[3];                 // grid origin: = gmin, or the low corner of the core range when far outliers were
                              // clipped (clouds refitted after an unbalanced first grid); cells at the grid border
                              // then also hold everything beyond it
    int closed;               // 1: the index holds only the points inside [org, org + G*h) (sub-box level of an unbalanced
                              // cloud); points beyond the grid border exist but are not in it, so border faces count
    int has_large;            // 1: the bucketed build met over-full buckets whose records are not placed yet (k_bucket_large is
                              // launched only on demand: searches see this flag, give up and tell the host, pcu_hip.hip C_LARGE)
    int nonfinite;            // kNf* flags (grid.h): non-finite coordinates met by the bbox pass
    unsigned long long sumsq; // sum over cells of count^2 (balance metric: sumsq / n = mean number of cell mates)
};

// Cell coordinate of value v along one axis. Separate subtract and multiply (the TU is built with
// -ffp-contract=off); NaN maps to cell 0; values on/after the last face are clamped into the last cell.
template <typename T>
__device__ __forceinline__ int cell_coord(T v, T gmin, T inv_h, int G) {
    T t = (v - gmin) * inv_h;
    return (t >= (T)0) ? ((t < (T)G) ? (int)t : G - 1) : 0;
}

template <typename T>
__device__ __forceinline__ int grid_cell(const GridParams<T>& g, const int axis, const T v) {
    return cell_coord(v, g.org[axis], g.inv_h, g.G[axis]);
}
// Every point whose cell coordinate along `axis` is <  c has a coordinate <  face_below(c)  (c >= 1);
// every point whose cell coordinate along `axis` is >  c has a coordinate >= face_above(c)  (c <= G-2).
// The slack covers the rounding of (v - org) * inv_h in cell_coord.
template <typename T>
__device__ __forceinline__ T face_below(const GridParams<T>& g, const int axis, const int c) { return g.org[axis] + (T)c * g.h + g.slack[axis]; }
template <typename T>
__device__ __forceinline__ T face_above(const GridParams<T>& g, const int axis, const int c) { return g.org[axis] + (T)(c + 1) * g.h - g.slack[axis]; }

// Cell order: boustrophedon ("snake"). Rows (y,z) are numbered z-major with y reversed on odd z, and cells inside
// a row run in +x on even rows and -x on odd rows, so consecutive cells of the linear order are always face
// neighbours: 64 consecutive points of a cell-ordered cloud form a compact snake instead of wrapping around the
// grid at row ends. The cells [xa..xb] of one row are still one contiguous run of the linear order.
__device__ __forceinline__ int grid_row(int Gy, int cy, int cz) { return cz * Gy + ((cz & 1) ? Gy - 1 - cy : cy); }
// linear index of the first cell of the run covering cells xa..xb (xa <= xb) of row `row`
__device__ __forceinline__ int row_run_lo(int Gx, int row, int xa, int xb) { return row * Gx + ((row & 1) ? Gx - 1 - xb : xa); }

}  // namespace pcu
