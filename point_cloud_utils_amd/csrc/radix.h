// csrc/radix.h -- the repo's own stable LSD radix sort of (64-bit key, 32-bit id) pairs and inclusive scan, for the rows that group points by an
// integer key triple (voxel-grid downsampling, src/sample_point_cloud.cpp:163-235; duplicate removal, src/remove_duplicates.cpp:11-36).
// Rounds 3-5 ordered the points with three rocPRIM radix sorts of full 32-bit keys (one per key component: ~30 launches, 0.51 of the 0.66 ms of
// a 1M-point voxel call) and scanned with rocPRIM. Round 6:
//   * the three components are packed into ONE key of only as many bits as their ranges need (a 128^3 voxel grid: 21 bits instead of 96),
//     so the sort is ceil(bits / 8) passes instead of twelve;
//   * a pass is three launches over wave tiles of 512 keys: per-tile digit histograms (LDS atomics), one block per digit scans its row of the
//     histogram table, and the scatter ranks every key inside its wave by an 8-step ballot match (equal digits keep their order: lanes in
//     order, rounds in order, tiles in order -- the sort is STABLE, which is what keeps a voxel's points in input order and its mean
//     bit-identical to the reference's sequence of additions).
// Keys wider than 64 bits in total (duplicate removal on raw coordinates: three full float patterns) are sorted component by component, least
// significant first, with the same passes.
#pragma once
#include "pcu_types.h"
#include "grid.h"

namespace pcu {

constexpr int kRsThreads = 256, kRsItems = 8, kRsWaveTile = 64 * kRsItems, kRsTile = kRsThreads * kRsItems;

// order-preserving map of a key component onto uint64
__device__ __forceinline__ unsigned long long key_u64(int v) { return (unsigned long long)((unsigned)v ^ 0x80000000u); }
__device__ __forceinline__ unsigned long long key_u64(unsigned v) { return (unsigned long long)v; }
__device__ __forceinline__ unsigned long long key_u64(unsigned long long v) { return v; }

struct KeyRange { unsigned long long lo[3], hi[3]; };          // per component, as key_u64 values (lo initialised to ~0, hi to 0)

// (one block-level fold, then six atomics per block: 1024 blocks x 4 waves of 64-bit atomics on six addresses serialised for 280 us)
template <typename K>
__global__ __launch_bounds__(kBlock) void k_key_range(const K* __restrict__ k0, const K* __restrict__ k1, const K* __restrict__ k2, int n, KeyRange* __restrict__ rng) {
    __shared__ unsigned long long s_lo[kBlock / 64][3], s_hi[kBlock / 64][3];
    unsigned long long lo[3] = {~0ull, ~0ull, ~0ull}, hi[3] = {0ull, 0ull, 0ull};
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock) {
        const unsigned long long u[3] = {key_u64(k0[i]), key_u64(k1[i]), key_u64(k2[i])};
#pragma unroll
        for (int j = 0; j < 3; ++j) { lo[j] = u[j] < lo[j] ? u[j] : lo[j]; hi[j] = u[j] > hi[j] ? u[j] : hi[j]; }
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned long long a = __shfl_xor(lo[j], o, 64), b = __shfl_xor(hi[j], o, 64);
            lo[j] = a < lo[j] ? a : lo[j]; hi[j] = b > hi[j] ? b : hi[j];
        }
        if ((threadIdx.x & 63) == 0) { s_lo[threadIdx.x >> 6][j] = lo[j]; s_hi[threadIdx.x >> 6][j] = hi[j]; }
    }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int j = threadIdx.x;
        unsigned long long l = s_lo[0][j], h = s_hi[0][j];
        for (int w = 1; w < kBlock / 64; ++w) { l = s_lo[w][j] < l ? s_lo[w][j] : l; h = s_hi[w][j] > h ? s_hi[w][j] : h; }
        atomicMin(&rng->lo[j], l); atomicMax(&rng->hi[j], h);
    }
}
// new run where the sorted packed key changes (the packed path's run heads: coalesced, no gathers through the permutation)
__global__ __launch_bounds__(kBlock) void k_run_heads_sorted(const unsigned long long* __restrict__ keys, int n, unsigned* __restrict__ flag) {
    const int j = blockIdx.x * kBlock + threadIdx.x;
    if (j < n) flag[j] = (j == 0 || keys[j] != keys[j - 1]) ? 1u : 0u;
}
// key = the three components, each less its minimum, in w0 + w1 + w2 <= 64 bits (component 0 most significant)
template <typename K>
__global__ __launch_bounds__(kBlock) void k_key_pack(const K* __restrict__ k0, const K* __restrict__ k1, const K* __restrict__ k2, int n, const KeyRange* __restrict__ rng,
                                                     int w1, int w2, unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i >= n) return;
    const unsigned long long a = key_u64(k0[i]) - rng->lo[0], b = key_u64(k1[i]) - rng->lo[1], c = key_u64(k2[i]) - rng->lo[2];
    out[i] = (w1 + w2 >= 64 ? 0ull : a << (w1 + w2)) | (w2 >= 64 ? 0ull : b << w2) | c;
}
// one component's keys in the order of a permutation, less the component's minimum (component-by-component sorts of wide keys)
template <typename K>
__global__ __launch_bounds__(kBlock) void k_key_gather(const K* __restrict__ src, const unsigned* __restrict__ perm, int n, const KeyRange* __restrict__ rng, int comp,
                                                       unsigned long long* __restrict__ out) {
    const int i = blockIdx.x * kBlock + threadIdx.x;
    if (i < n) out[i] = key_u64(src[perm ? perm[i] : (unsigned)i]) - rng->lo[comp];
}

// ---- one radix pass (digit = bits [shift, shift + 8) of the key)
// hist[d * nwt + t] = number of keys of wave tile t with digit d
__global__ __launch_bounds__(kRsThreads) void k_rs_hist(const unsigned long long* __restrict__ keys, int n, int shift, int nwt, unsigned* __restrict__ hist) {
    __shared__ unsigned s_h[kRsThreads / 64][256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wt = blockIdx.x * (kRsThreads / 64) + wave;
#pragma unroll
    for (int q = 0; q < 4; ++q) s_h[wave][lane + 64 * q] = 0u;
    __builtin_amdgcn_wave_barrier();
    if (wt < nwt) {
        const long long base = (long long)wt * kRsWaveTile;
#pragma unroll
        for (int j = 0; j < kRsItems; ++j) {
            const long long i = base + j * 64 + lane;
            if (i < n) atomicAdd(&s_h[wave][(unsigned)(keys[i] >> shift) & 255u], 1u);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#pragma unroll
        for (int q = 0; q < 4; ++q) hist[(size_t)(lane + 64 * q) * nwt + wt] = s_h[wave][lane + 64 * q];
    }
}
// block d: exclusive scan of row d of the table in place; total[d] = the row's sum
__global__ __launch_bounds__(1024) void k_rs_scan_rows(unsigned* __restrict__ hist, int nwt, unsigned* __restrict__ total) {
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    unsigned* const row = hist + (size_t)blockIdx.x * nwt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int c0 = 0; c0 < nwt; c0 += 1024) {
        const int i = c0 + tid;
        const unsigned v = i < nwt ? row[i] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned wbase = 0;
        for (int w = 0; w < wave; ++w) wbase += s_w[w];
        const unsigned carry = s_carry;
        if (i < nwt) row[i] = carry + wbase + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wbase + inc;
        __syncthreads();
    }
    if (tid == 0) total[blockIdx.x] = s_carry;
}
// keys_in / ids_in (ids_in == nullptr: the identity) -> their stable order by the digit
__global__ __launch_bounds__(kRsThreads) void k_rs_scatter(const unsigned long long* __restrict__ keys_in, const unsigned* __restrict__ ids_in, int n, int shift, int nwt,
                                                           const unsigned* __restrict__ hist, const unsigned* __restrict__ total,
                                                           unsigned long long* __restrict__ keys_out, unsigned* __restrict__ ids_out) {
    __shared__ unsigned s_c[kRsThreads / 64][256];
    __shared__ unsigned s_base[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    {   // where digit d starts in the output: exclusive scan of the 256 row totals (one per thread)
        const unsigned v = total[tid];
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
        __shared__ unsigned s_w[kRsThreads / 64];
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned wb = 0;
        for (int w = 0; w < wave; ++w) wb += s_w[w];
        s_base[tid] = wb + inc - v;
        __syncthreads();
    }
    const int wt = blockIdx.x * (kRsThreads / 64) + wave;
    if (wt >= nwt) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) { const int d = lane + 64 * q; s_c[wave][d] = s_base[d] + hist[(size_t)d * nwt + wt]; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    const long long base = (long long)wt * kRsWaveTile;
    const unsigned long long lt = (1ull << lane) - 1ull;
    for (int j = 0; j < kRsItems; ++j) {
        const long long i = base + j * 64 + lane;
        const bool valid = i < n;
        const unsigned long long key = valid ? keys_in[i] : 0ull;
        const unsigned id = valid ? (ids_in ? ids_in[i] : (unsigned)i) : 0u;
        const unsigned d = (unsigned)(key >> shift) & 255u;
        unsigned long long peers = __ballot(valid);
#pragma unroll
        for (int b = 0; b < 8; ++b) { const unsigned long long m = __ballot((d >> b) & 1u); peers &= ((d >> b) & 1u) ? m : ~m; }
        const unsigned rank = (unsigned)__popcll(peers & lt);
        const unsigned start = valid ? s_c[wave][d] : 0u;           // (every peer reads the group's counter before its first lane advances it)
        __builtin_amdgcn_wave_barrier();
        if (valid && rank == 0u) s_c[wave][d] = start + (unsigned)__popcll(peers);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (valid) { keys_out[start + rank] = key; ids_out[start + rank] = id; }
    }
}

// ---- inclusive scan of n unsigned values: tiles of 4096, one block for the tile sums, add
constexpr int kScTile = 4096;
__global__ __launch_bounds__(1024) void k_sc_tiles(const unsigned* __restrict__ in, unsigned* __restrict__ out, int n, unsigned* __restrict__ tile_sum) {
    __shared__ unsigned s_w[16];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long i0 = (long long)blockIdx.x * kScTile + 4ll * tid;
    unsigned v[4], s = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) { v[q] = i0 + q < n ? in[i0 + q] : 0u; s += v[q]; v[q] = s; }
    unsigned inc = s;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
    if (lane == 63) s_w[wave] = inc;
    __syncthreads();
    unsigned wb = 0;
    for (int w = 0; w < wave; ++w) wb += s_w[w];
    const unsigned ex = wb + inc - s;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i0 + q < n) out[i0 + q] = ex + v[q];
    if (tid == 1023) tile_sum[blockIdx.x] = wb + inc;
}
__global__ __launch_bounds__(1024) void k_sc_sums(unsigned* __restrict__ tile_sum, int nt) {       // exclusive scan in place (one block)
    __shared__ unsigned s_w[16];
    __shared__ unsigned s_carry;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_carry = 0u;
    __syncthreads();
    for (int c0 = 0; c0 < nt; c0 += 1024) {
        const int i = c0 + tid;
        const unsigned v = i < nt ? tile_sum[i] : 0u;
        unsigned inc = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
        if (lane == 63) s_w[wave] = inc;
        __syncthreads();
        unsigned wb = 0;
        for (int w = 0; w < wave; ++w) wb += s_w[w];
        const unsigned carry = s_carry;
        if (i < nt) tile_sum[i] = carry + wb + inc - v;
        __syncthreads();
        if (tid == 1023) s_carry = carry + wb + inc;
        __syncthreads();
    }
}
__global__ __launch_bounds__(1024) void k_sc_add(unsigned* __restrict__ out, int n, const unsigned* __restrict__ tile_sum) {
    const unsigned add = tile_sum[blockIdx.x];
    const long long i0 = (long long)blockIdx.x * kScTile + 4ll * threadIdx.x;
#pragma unroll
    for (int q = 0; q < 4; ++q) if (i0 + q < n) out[i0 + q] += add;
}

}  // namespace pcu
