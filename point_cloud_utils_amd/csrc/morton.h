// csrc/morton.h -- 64-bit 3-D Morton codes: element-wise integer kernels, HBM-bound.
//
// Replaces MortonCode64 (src/common/morton_code.cpp:12-166: 21 bits per axis in two's complement, x in the lowest bit of every
// triple, the three sign bits -- bits 60..62 -- stored inverted so that unsigned order of the codes follows the signed order
// of the coordinates) and the loops of morton_encode / morton_decode / morton_add / morton_subtract / morton_knn
// (src/morton.cpp:185-414). All arithmetic is integer: results are bit-identical to the reference.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace pcu {

constexpr uint64_t kMortonSign = 0x7000000000000000ull;
constexpr uint64_t kMortonX = 0x1249249249249249ull;          // ...001001001: the x bit of every triple

__host__ __device__ __forceinline__ uint64_t morton_split21(uint64_t r) {      // SplitBy3Bits21 (:13-25)
    r = (r | r << 32) & 0x1f00000000ffffull;
    r = (r | r << 16) & 0x1f0000ff0000ffull;
    r = (r | r << 8) & 0x100f00f00f00f00full;
    r = (r | r << 4) & 0x10c30c30c30c30c3ull;
    r = (r | r << 2) & 0x1249249249249249ull;
    return r;
}
__host__ __device__ __forceinline__ int32_t morton_compact21(uint64_t x) {     // CompactBy3Bits21 (:27-40)
    uint64_t d = x & 0x1249249249249249ull;
    d = (d | d >> 2) & 0x10c30c30c30c30c3ull;
    d = (d | d >> 4) & 0x100f00f00f00f00full;
    d = (d | d >> 8) & 0x1f0000ff0000ffull;
    d = (d | d >> 16) & 0x1f00000000ffffull;
    d = (d | d >> 32);
    d = (d & 0x100000ull) ? (d | 0xffe00000ull) : d;          // sign extension
    return (int32_t)d;
}
// MortonCode64(int32_t x, int32_t y, int32_t z) (:46-66). The sign bit moves to bit 20; SplitBy3Bits21 takes an int32_t and
// widens it to uint64_t, which for the (non-negative) 21-bit value is the value itself.
__host__ __device__ __forceinline__ uint64_t morton_encode3(int32_t x, int32_t y, int32_t z) {
    const uint32_t ux = (((uint32_t)x & 0x80000000u) >> 11) | ((uint32_t)x & 0x0fffffu);
    const uint32_t uy = (((uint32_t)y & 0x80000000u) >> 11) | ((uint32_t)y & 0x0fffffu);
    const uint32_t uz = (((uint32_t)z & 0x80000000u) >> 11) | ((uint32_t)z & 0x0fffffu);
    const uint64_t data = morton_split21(ux) | morton_split21(uy) << 1 | morton_split21(uz) << 2;
    return data ^ kMortonSign;
}
__host__ __device__ __forceinline__ void morton_decode3(uint64_t data, int32_t& x, int32_t& y, int32_t& z) {   // decode (:77-85)
    const uint64_t d = data ^ kMortonSign;
    x = morton_compact21(d); y = morton_compact21(d >> 1); z = morton_compact21(d >> 2);
}
__host__ __device__ __forceinline__ uint64_t morton_add2(uint64_t a, uint64_t b) {            // operator+ (:131-145)
    const uint64_t c1 = a ^ kMortonSign, c2 = b ^ kMortonSign;
    const uint64_t ym = kMortonX << 1, zm = kMortonX << 2;
    const uint64_t xs = (c1 | ~kMortonX) + (c2 & kMortonX), ys = (c1 | ~ym) + (c2 & ym), zs = (c1 | ~zm) + (c2 & zm);
    return ((xs & kMortonX) | (ys & ym) | (zs & zm)) ^ kMortonSign;
}
__host__ __device__ __forceinline__ uint64_t morton_negate(uint64_t data) {                   // Negate (:116-129)
    const uint64_t ym = kMortonX << 1, zm = kMortonX << 2;
    const uint64_t d = ~data;
    const uint64_t xs = (d | ~kMortonX) + 1, ys = (d | ~ym) + 1, zs = (d | ~zm) + 1;
    return (xs & kMortonX) | (ys & ym) | (zs & zm);
}

// pts (n,3) of I (int32 / int64; the reference narrows to int32_t, src/morton.cpp:236) -> codes (n)
template <typename I>
__global__ __launch_bounds__(256) void k_morton_encode(const I* __restrict__ pts, long long n, uint64_t* __restrict__ codes) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256)
        codes[i] = morton_encode3((int32_t)pts[3 * i], (int32_t)pts[3 * i + 1], (int32_t)pts[3 * i + 2]);
}
// codes (n) of C (uint32 / uint64) -> pts (n,3) int32
template <typename C>
__global__ __launch_bounds__(256) void k_morton_decode(const C* __restrict__ codes, long long n, int32_t* __restrict__ pts) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        int32_t x, y, z;
        morton_decode3((uint64_t)codes[i], x, y, z);
        pts[3 * i] = x; pts[3 * i + 1] = y; pts[3 * i + 2] = z;
    }
}
// out = c1 + c2 (sub = 0) or c1 - c2 = c1 + Negate(c2) (sub = 1)   (src/morton.cpp:81-83, :163-165)
template <typename C1, typename C2>
__global__ __launch_bounds__(256) void k_morton_addsub(const C1* __restrict__ c1, const C2* __restrict__ c2, long long n, int sub, uint64_t* __restrict__ out) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        const uint64_t a = (uint64_t)c1[i], b = (uint64_t)c2[i];
        out[i] = morton_add2(a, sub ? morton_negate(b) : b);
    }
}

// morton_knn (src/morton.cpp:339-412): for every query code the window of k consecutive entries of the SORTED code array
// around its lower bound -- k/2 above, k - k/2 below, shifted back inside the array at its ends. sort_dist: the reference sorts
// the window with a comparator that reads the query coordinates before ever decoding them (uninitialised q_x/q_y/q_z, :386-398:
// undefined behaviour); here the window is ordered by the squared distance between the decoded query and the decoded entries
// (ties: lower index first), which is what the comparator was written to do.
template <typename C>
__global__ __launch_bounds__(256) void k_morton_knn(const C* __restrict__ codes, long long n, const C* __restrict__ qcodes, long long m, int k, int sort_dist,
                                                    long long* __restrict__ nn) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= m) return;
    const C q = qcodes[i];
    long long lo = 0, hi = n;                                     // std::lower_bound
    while (lo < hi) { const long long mid = lo + ((hi - lo) >> 1); if (codes[mid] < q) lo = mid + 1; else hi = mid; }
    const long long idx = lo;
    const int half_up = k / 2, half_down = k - half_up;
    long long upper = idx + half_up, lower = idx - half_down;
    if (upper >= n) { lower -= (upper - n); upper = n; }
    if (lower < 0) { upper += -lower; lower = 0; }
    long long* row = nn + i * (long long)k;
    const int cnt = (int)(upper - lower);
    if (!sort_dist) { for (int j = 0; j < cnt; ++j) row[j] = lower + j; return; }
    int32_t qx, qy, qz;
    morton_decode3((uint64_t)q, qx, qy, qz);
    auto dist = [&](long long e) {
        int32_t x, y, z; morton_decode3((uint64_t)codes[e], x, y, z);
        const double dx = (double)qx - x, dy = (double)qy - y, dz = (double)qz - z;
        return dx * dx + dy * dy + dz * dz;
    };
    for (int j = 0; j < cnt; ++j) {                               // insertion sort of the row (k is small)
        const long long e = lower + j; const double de = dist(e);
        int p = j;
        while (p > 0 && dist(row[p - 1]) > de) { row[p] = row[p - 1]; --p; }
        row[p] = e;
    }
}

}  // namespace pcu
