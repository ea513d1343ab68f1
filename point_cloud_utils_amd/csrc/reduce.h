// csrc/reduce.h -- epilogues over the per-query nearest-neighbour results.
//
//   Hausdorff  (src/point_cloud_distance.cpp:221-225): arg-max of the per-source distance array with Eigen's
//              maxCoeff rule -- strict '>' while visiting rows in order, i.e. the FIRST row attaining the maximum.
//   Chamfer    (point_cloud_utils/__init__.py:112-115): mean over queries of || nn(q) - q ||_p.
// Both are two-stage reductions inside ONE launch (per-block partials; the block that finishes last folds them), fp64
// accumulation for sums.
#pragma once
#include "pcu_types.h"
#include "grid.h"

namespace pcu {

constexpr int kRedBlocks = 1024;
#ifndef PCU_RED_BLOCKS
#define PCU_RED_BLOCKS 128
#endif
constexpr int kRedBlocksFused = PCU_RED_BLOCKS;     // per direction in the single-launch epilogues: every block takes one same-address ticket

template <typename T>
__device__ __forceinline__ void argmax_combine(T& v, long long& i, T v2, long long i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// Hausdorff epilogue in one launch (both directions; nb = 0 for an absent second direction): per-block arg-max
// partials, the last block to finish folds them, writes {value, i, j} per direction and copies the call's result
// block to pinned host memory (see k_pnorm_pair).
template <typename T>
struct ArgmaxSide { const T* d; const Pt4<T>* qsorted; const long long* corr; int n; int nb; };

template <typename T>
__device__ __forceinline__ void block_argmax(T& v, long long& idx) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        T v2 = __shfl_xor(v, o, 64); long long i2 = __shfl_xor(idx, o, 64);
        argmax_combine(v, idx, v2, i2);
    }
    __shared__ T sv[kBlock / 64]; __shared__ long long si[kBlock / 64];
    __syncthreads();
    if ((threadIdx.x & 63) == 0) { sv[threadIdx.x >> 6] = v; si[threadIdx.x >> 6] = idx; }
    __syncthreads();
    if (threadIdx.x == 0)
        for (int w = 1; w < kBlock / 64; ++w) argmax_combine(v, idx, sv[w], si[w]);      // valid in thread 0
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_argmax_pair(const ArgmaxSide<T> s0, const ArgmaxSide<T> s1, T* pv, long long* pi,
                                                        T* out_v, long long* out_ij, unsigned* ticket, const int* result_block, int* host_block, unsigned seq) {
    const bool second = (int)blockIdx.x >= s0.nb;
    const ArgmaxSide<T>& sd = second ? s1 : s0;
    const int bid = second ? (int)blockIdx.x - s0.nb : (int)blockIdx.x;
    T v = -Limits<T>::max_v; long long idx = 0x7fffffffffffffffll;
    for (int i = bid * kBlock + threadIdx.x; i < sd.n; i += sd.nb * kBlock)
        argmax_combine(v, idx, sd.d[i], ((long long)sd.qsorted[i].idx << 32) | (long long)i);
    block_argmax(v, idx);
    __shared__ bool s_last;
    if (threadIdx.x == 0) {
        publish(&pv[blockIdx.x], v); publish(&pi[blockIdx.x], idx);
        wait_stores();
        s_last = take_ticket(ticket, gridDim.x);
    }
    __syncthreads();
    if (!s_last) return;
    for (int side = 0; side < 2; ++side) {
        const int off = side ? s0.nb : 0, nb = side ? s1.nb : s0.nb;
        if (nb == 0) continue;
        T a = -Limits<T>::max_v; long long ai = 0x7fffffffffffffffll;
        for (int i = threadIdx.x; i < nb; i += kBlock) argmax_combine(a, ai, peek(&pv[off + i]), peek(&pi[off + i]));
        block_argmax(a, ai);
        if (threadIdx.x == 0) {
            out_v[side] = a; out_ij[2 * side] = ai >> 32;
            out_ij[2 * side + 1] = (side ? s1.corr : s0.corr)[ai & 0xffffffffll];
        }
    }
    if (threadIdx.x == 0) { *ticket = 0u; wait_stores(); }
    __syncthreads();
    if (host_block && threadIdx.x < 64) {       // one wave: 63 data words, system-scope fence, then the sequence word the host spins on
        if (threadIdx.x < 63) host_block[threadIdx.x] = peek(&result_block[threadIdx.x]);
        __threadfence_system();
        if (threadIdx.x == 63) { __hip_atomic_store(&host_block[63], (int)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

// ---- fused epilogues (search.h: k_search1_flat / k_search_wave with SearchArgs::fuse != 0) ---------------------------------
// Chamfer (p = 2, no indices) and Hausdorff need one scalar (pair) per direction, not the per-query rows. In fused mode the
// lane-per-query pass reduces its certified lanes' distances to ONE partial per block (fp64 sum, or arg-max) instead of
// writing (d, idx) rows, the wave-per-query pass adds its few queries, and the block of that launch which finishes last folds
// everything and hands the call's result block to the host: no rows written (-24 B per query of HBM writes and the re-read),
// no separate epilogue launch.
//
// The wave-per-query pass takes its queries from device-side lists whose ORDER depends on the timing of atomics, so its
// share of a floating-point sum would not be reproducible run to run. It is therefore accumulated EXACTLY: every value is
// split into 32-bit limbs of a wide fixed-point number (bit 0 = 2^-1074) and each limb is added with an integer atomic into
// its own 64-bit word (2^32 additions cannot overflow a word; carries are propagated once, by the fold). Integer addition
// commutes, so the result does not depend on the order, and the fold's conversion back to double is deterministic.
constexpr int kAccLimbs = 66;              // (2045 + 53 + 32) / 32 + 1: any finite double fits
__device__ __forceinline__ void exact_add(unsigned long long* limbs, double* special, double v) {
    const unsigned long long b = (unsigned long long)__double_as_longlong(v);
    const unsigned be = (unsigned)(b >> 52) & 0x7ffu;
    if (be == 0x7ffu || (long long)b < 0) { atomicAdd(special, v); return; }         // inf / NaN (and, never here, negatives): their sum is order-independent as a class
    const unsigned long long m = (b & 0x000fffffffffffffull) | (be ? 0x0010000000000000ull : 0ull);
    if (m == 0) return;
    const unsigned sh = be ? be - 1u : 0u;          // v = m * 2^(sh - 1074)
    const unsigned L = sh >> 5, off = sh & 31u;
    const unsigned long long lo64 = m << off, hi = off ? m >> (64u - off) : 0ull;      // the 85-bit value m << off
    const unsigned long long w0 = lo64 & 0xffffffffull, w1 = lo64 >> 32, w2 = hi;
    if (w0) atomicAdd(&limbs[L], w0);
    if (w1) atomicAdd(&limbs[L + 1], w1);
    if (w2) atomicAdd(&limbs[L + 2], w2);
}
// One thread: the accumulated value as a double (carries propagated in place, then summed from the top limb down).
__device__ __noinline__ double exact_value(unsigned long long* limbs, const double* special) {
    unsigned long long carry = 0;
#pragma unroll 1
    for (int i = 0; i < kAccLimbs; ++i) {
        const unsigned long long v = __hip_atomic_load(&limbs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) + carry;
        __hip_atomic_store(&limbs[i], v & 0xffffffffull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        carry = v >> 32;
    }
    double r = 0;
#pragma unroll 1
    for (int i = kAccLimbs - 1; i >= 0; --i) {
        const unsigned long long v = __hip_atomic_load(&limbs[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (v) r += ldexp((double)v, 32 * i - 1074);
    }
    return r + __hip_atomic_load(special, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

enum { FUSE_NONE = 0, FUSE_SUM = 1, FUSE_ARGMAX = 2 };
// What the wave-per-query launch of a fused call needs to finish the call (by value, in its kernel arguments).
template <typename T>
struct FuseTail {
    int mode;                                    // FUSE_*
    const double* flat_sum[2]; const T* flat_v[2]; const long long* flat_k[2]; int nflat[2];   // the lane pass's per-block partials, per direction
    double* wsum; T* wv; long long* wk;          // this launch's per-block partials, [direction][gridDim.x] (FUSE_ARGMAX)
    unsigned long long* limbs; double* special;  // [direction][kAccLimbs], [direction]: exact sum of this launch's distances (FUSE_SUM)
    unsigned* ticket;
    double* out_sums; T* out_v; long long* out_ij; int* out_tie;      // fields of the call's result block
    const int* result_block; int* host_block; unsigned seq;
};

// Hausdorff, row-based path: is the arg-max source row (ij[0], in the call's result block) one of the direction's queries
// with a genuine tie? Only then does the returned j depend on the reference's tie order (pcu_hip.hip, hausdorff_end).
template <typename T>
__global__ __launch_bounds__(kBlock) void k_tie_hit(const int* __restrict__ tt, int n, const Pt4<T>* __restrict__ qsorted,
                                                    const long long* __restrict__ ij, int* flag) {
    const long long row = ij[0];
    for (int i = blockIdx.x * kBlock + threadIdx.x; i < n; i += gridDim.x * kBlock)
        if ((long long)qsorted[tt[i]].idx == row) *flag = 1;
}

__device__ __forceinline__ double block_sum(double s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ss[kBlock / 64];
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = s;
    __syncthreads();
    double r = 0;
    if (threadIdx.x == 0) for (int w = 0; w < kBlock / 64; ++w) r += ss[w];
    return r;   // valid in thread 0
}

// p-norm codes (numpy.linalg.norm vector ord): 2, 1, +inf, -inf, 0, other
enum { P_TWO = 0, P_ONE = 1, P_INF = 2, P_NINF = 3, P_ZERO = 4, P_GEN = 5 };

// Per-query norm || tgt[corr[pos]] - qsorted[pos] ||_p, summed with fp64 accumulation (queries in cell order). For p == 2 the
// already computed (non-squared) nn distances d[pos] are summed instead: they are the same numbers,
// sqrt(((dx*dx)+(dy*dy))+(dz*dz)) (numpy's norm(ord=2, axis=-1) squares, add-reduces in axis order and takes sqrt, in the
// input dtype).
// Chamfer epilogue in one launch: blocks [0, nb0) reduce direction 0, blocks [nb0, nb0 + nb1) direction 1; the block
// that finishes last (ticket) folds both directions' partials and copies the call's 256-byte result block (counters of
// both searches + these sums) into pinned host memory, so the call needs no separate final-sum launches and no
// device-to-host copy kernel behind them.
template <typename T>
struct PnormSide { const Pt4<T>* qsorted; const T* tgt; const long long* corr; const T* d; int n; int nb; };

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pnorm_pair(const PnormSide<T> s0, const PnormSide<T> s1, int pcode, double p, double* partial,
                                                       double* out_sums, unsigned* ticket, const int* result_block, int* host_block, unsigned seq) {
    const bool second = (int)blockIdx.x >= s0.nb;
    const PnormSide<T>& sd = second ? s1 : s0;
    const int bid = second ? (int)blockIdx.x - s0.nb : (int)blockIdx.x;
    double s = 0;
    const int n_vec = pcode == P_TWO ? (sd.n & ~3) : 0;       // p = 2: the distances themselves, four per 16/32-byte load
    for (int i = 4 * (bid * kBlock + (int)threadIdx.x); i < n_vec; i += 4 * sd.nb * kBlock) {
        const T v0 = sd.d[i], v1 = sd.d[i + 1], v2 = sd.d[i + 2], v3 = sd.d[i + 3];
        s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (int i = n_vec + bid * kBlock + (int)threadIdx.x; i < sd.n; i += sd.nb * kBlock) {
        T v;
        if (pcode == P_TWO) {
            v = sd.d[i];
        } else {
            const long long c = sd.corr[i];
            const Pt4<T> q = sd.qsorted[i];
            const T a = sd.tgt[3 * c] - q.x, b = sd.tgt[3 * c + 1] - q.y, e = sd.tgt[3 * c + 2] - q.z;
            const T aa = a < 0 ? -a : a, ab = b < 0 ? -b : b, ae = e < 0 ? -e : e;
            if (pcode == P_ONE) v = (aa + ab) + ae;
            else if (pcode == P_INF) { v = aa > ab ? aa : ab; v = v > ae ? v : ae; }
            else if (pcode == P_NINF) { v = aa < ab ? aa : ab; v = v < ae ? v : ae; }
            else if (pcode == P_ZERO) v = (T)((a != 0) + (b != 0) + (e != 0));
            else v = (T)pow((double)(T)((T)pow((double)aa, p) + (T)pow((double)ab, p)) + (double)(T)pow((double)ae, p), 1.0 / p);
        }
        s += (double)v;
    }
    const double r = block_sum(s);
    __shared__ bool s_last;
    if (threadIdx.x == 0) {
        publish(&partial[blockIdx.x], r);
        wait_stores();
        s_last = take_ticket(ticket, gridDim.x);
    }
    __syncthreads();
    if (!s_last) return;
    double a0 = 0, a1 = 0;
    for (int i = threadIdx.x; i < s0.nb; i += kBlock) a0 += peek(&partial[i]);
    for (int i = threadIdx.x; i < s1.nb; i += kBlock) a1 += peek(&partial[s0.nb + i]);
    __syncthreads();                         // block_sum's shared array is reused
    const double r0 = block_sum(a0);
    __syncthreads();
    const double r1 = block_sum(a1);
    if (threadIdx.x == 0) { out_sums[0] = r0; out_sums[1] = r1; *ticket = 0u; wait_stores(); }
    __syncthreads();
    if (host_block && threadIdx.x < 64) {       // one wave: 63 data words, system-scope fence, then the sequence word the host spins on
        if (threadIdx.x < 63) host_block[threadIdx.x] = peek(&result_block[threadIdx.x]);
        __threadfence_system();
        if (threadIdx.x == 63) { __hip_atomic_store(&host_block[63], (int)seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM); }
    }
}

}  // namespace pcu
