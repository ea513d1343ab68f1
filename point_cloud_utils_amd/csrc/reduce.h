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
struct ArgmaxSide { const T* d; const Pt4<T>* qsorted; const long long* corr; int n; int nb; };      // qsorted == nullptr: d / corr are in the caller's ROW order

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
        argmax_combine(v, idx, sd.d[i], ((long long)(sd.qsorted ? (int)sd.qsorted[i].idx : i) << 32) | (long long)i);
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
            // (no row took part -- every distance NaN, e.g. the rows of a pass that gave up were never written -- : no index to read)
            const long long pos = ai & 0xffffffffll;
            out_v[side] = a; out_ij[2 * side] = ai >> 32;
            out_ij[2 * side + 1] = pos < (long long)(side ? s1.n : s0.n) ? (side ? s1.corr : s0.corr)[pos] : -1ll;
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
// its own 64-bit word (2^32 additions cannot overflow a word). Integer addition commutes, so the limbs do not depend on the
// order, and the fold's conversion back to double (exact_term / exact_total) is a fixed sequence of operations on them.
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
// The accumulated value as a double: term l = limb l scaled to its weight (thread l of the folding block), the terms summed by
// block_sum's fixed tree. A limb holds up to 2^32 additions of 32-bit values, so its conversion to double may round (2^-53
// relative): the result is deterministic -- which is the point -- and accurate far beyond the fp64 sums it is added to.
__device__ __forceinline__ double exact_term(const unsigned long long* limbs, const double* special, int l) {
    if (l >= kAccLimbs) return __hip_atomic_load(special, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const unsigned long long v = __hip_atomic_load(&limbs[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return v ? ldexp((double)v, 32 * l - 1074) : 0.0;
}
enum { FUSE_NONE = 0, FUSE_SUM = 1, FUSE_ARGMAX = 2, FUSE_MAXVAL = 3 };      // (MAXVAL: a variant of the lane kernel only -- see FuseTail::maxval; everything else says ARGMAX)
constexpr long long kKeyUnresolved = 1ll << 30;      // FUSE_MAXVAL keys: low word = this bit | the query's position in its cloud's cell order (clouds hold < 2^27 rows)
constexpr int kTailThreads = 1024;
// Arguments of k_fuse_tail, the one-block launch that ends a fused call.
template <typename T>
struct FuseTail {
    int mode;                                    // FUSE_*
    int njobs;                                   // directions
    const double* flat_sum[2]; const T* flat_v[2]; const long long* flat_k[2]; int nflat[2];   // the lane pass's per-block partials, per direction
    const T* wave_v[2]; const long long* wave_k[2]; int nwaves;                                // the wave pass's per-wave arg-max partials
    unsigned long long* limbs[2]; double* special[2];                                          // the wave pass's exact sums
    const int* result_block; int* host_block; unsigned seq;
    int w_sums, w_vals, w_ij, w_tie;             // word offsets of sums[2] / vals / ij[4] / tie[2] inside the result block
    // Round 6, Hausdorff (FUSE_ARGMAX with maxval != 0): the lane pass ran its VALUE-ONLY program (search.h: FUSE_MAXVAL -- the fused sum's scan with
    // adoption, about half the cost of the winner-tracking one) and its partials name the arg-max QUERY, not its neighbour: key = source row << 32 |
    // kKeyUnresolved | query position. The one query that wins is resolved here, by this block: it scans the 5 x 5 x 5 cells around the query (the lane
    // pass certified its value inside the 27, or inside these by its radius-2 rescue) in the distance's own arithmetic, takes the minimum, and
    // reports the dataset row -- or, if two records share the minimum, the tie flag, which sends the call to the row-based path as before.
    int maxval, squared;
    long long* prof;                             // diagnostics (PCU_HIP_PROF_TAIL): stage times of thread 0, summed over calls; 100 MHz ticks; [7] = calls
    const GridParams<T>* r_gp[2]; const unsigned* r_cs[2]; const T* r_xyz[2]; const int* r_idx[2]; const T* q_xyz[2];
};

// The nearest dataset record of a query among the 5 x 5 x 5 cells around it: its row, and whether a second record lies at exactly the same squared
// distance. The block serves BOTH directions at once -- threads [0, NT/2) direction 0, the rest direction 1, 16 threads per row of the box -- because
// the resolution is a chain of four dependent round trips (query + grid -> row bounds -> records -> row id), not work: one after the other they
// cost 12 us, side by side 5. need[d]: direction d has a winner to resolve at position qpos[d]. Results valid in every thread.
template <typename T, int NT>
__device__ __forceinline__ void tail_resolve2(const FuseTail<T>& ft, const bool (&need)[2], const unsigned (&qpos)[2], long long (&j_out)[2], int (&tie_out)[2], T (&min_d2)[2]) {
    constexpr int H = NT / 2, HW = H / 64;
    static_assert(H / 16 >= 25, "a row of the box per 16 threads");
    __shared__ T s_d[NT / 64]; __shared__ unsigned s_pos[NT / 64], s_cnt[NT / 64]; __shared__ int s_row[NT / 64];
    const int tid = threadIdx.x, d = tid >= H ? 1 : 0, lt = tid - d * H;
    T best = Limits<T>::max_v; unsigned pos = 0xffffffffu, cnt = 0; int row = 0x7fffffff;      // (row: the dataset row of `pos`, fetched with the record -- not a round trip of its own at the end)
    auto take = [&](T d2, unsigned p2, unsigned c2, int r2) {
        if (d2 < best) { best = d2; pos = p2; cnt = c2; row = r2; }
        else if (d2 == best) { cnt += c2; if (p2 < pos) { pos = p2; row = r2; } }
    };
    if (need[d]) {
        const GridParams<T>& g = *ft.r_gp[d];
        const T* const qp = ft.q_xyz[d] + 3 * (size_t)qpos[d];
        const T qx = qp[0], qy = qp[1], qz = qp[2];
        const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
        const int cx = grid_cell(g, 0, qx), cy = grid_cell(g, 1, qy), cz = grid_cell(g, 2, qz);
        const int bx0 = max(cx - 2, 0), bx1 = min(cx + 2, Gx - 1), by0 = max(cy - 2, 0), by1 = min(cy + 2, Gy - 1), bz0 = max(cz - 2, 0), bz1 = min(cz + 2, Gz - 1);
        const int ny = by1 - by0 + 1, nrows = ny * (bz1 - bz0 + 1);
        const int r = lt >> 4, sl = lt & 15;
        if (r < nrows) {
            const unsigned lo = (unsigned)row_run_lo(Gx, grid_row(Gy, by0 + r % ny, bz0 + r / ny), bx0, bx1);
            const unsigned s = ft.r_cs[d][lo], e = ft.r_cs[d][lo + (unsigned)(bx1 - bx0 + 1)];
            const T* const xyz = ft.r_xyz[d]; const int* const rid = ft.r_idx[d];
            for (unsigned i = s + (unsigned)sl; i < e; i += 16u) {
                const int ri = rid[i];
                const T dx = qx - xyz[3 * (size_t)i], dy = qy - xyz[3 * (size_t)i + 1], dz = qz - xyz[3 * (size_t)i + 2];
                take(((dx * dx) + (dy * dy)) + (dz * dz), i, 1u, ri);         // (the distance's own arithmetic: nanoflann.hpp:496-507)
            }
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1)            // (a wave belongs to one direction: H is a multiple of 64)
        take(__shfl_xor(best, o, 64), (unsigned)__shfl_xor((int)pos, o, 64), (unsigned)__shfl_xor((int)cnt, o, 64), __shfl_xor(row, o, 64));
    __syncthreads();
    if ((tid & 63) == 0) { s_d[tid >> 6] = best; s_pos[tid >> 6] = pos; s_cnt[tid >> 6] = cnt; s_row[tid >> 6] = row; }
    __syncthreads();
#pragma unroll
    for (int dd_ = 0; dd_ < 2; ++dd_) {
        best = s_d[dd_ * HW]; pos = s_pos[dd_ * HW]; cnt = s_cnt[dd_ * HW]; row = s_row[dd_ * HW];
        for (int w = 1; w < HW; ++w) take(s_d[dd_ * HW + w], s_pos[dd_ * HW + w], s_cnt[dd_ * HW + w], s_row[dd_ * HW + w]);
        min_d2[dd_] = best;
        j_out[dd_] = (need[dd_] && pos != 0xffffffffu) ? (long long)row : 0x7fffffffll;
        tie_out[dd_] = (cnt > 1u || pos == 0xffffffffu) ? 1 : 0;
    }
}

// ONE block folds a fused call: the lane pass's per-block partials (k_search1_flat), the wave pass's share (exact limbs or
// per-wave arg-max partials), both directions; the results and the search counters of the call's result block go to pinned
// host memory, sequence word last (the host spins on it, see k_pnorm_pair). Every thread requests all its inputs before the
// first reduction -- the launch is a chain of memory round trips, not work -- and all sums run in a fixed order (thread-strided
// partial sums, then a fixed tree), so the value is reproducible run to run.
// The fold's sums always run in the order of kTailLanes = 1024 virtual threads -- thread-strided partial sums, a butterfly inside every 64
// of them, the 16 wave sums added in order. k_fuse_tail runs it with 1024 real threads; a block of NT < 1024 threads gives the same bits
// (thread t stands for the virtual threads t, t + NT, ...). Round 3 measured the NT = 256 form as the last block of the wave pass (one launch
// less): 21.4 us against 9.3 + 7.0 us for the two launches -- the fold is a chain of dependent round trips that 4x fewer threads walk 4x
// longer -- so only the 1024-thread kernel is built.
constexpr int kTailLanes = 1024;
template <int NT>
__device__ __forceinline__ double tail_sum(const double (&v)[kTailLanes / NT], double* s_buf) {      // valid in every thread
    constexpr int V = kTailLanes / NT;
    double w[V];
#pragma unroll
    for (int q = 0; q < V; ++q) {
        w[q] = v[q];
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) w[q] += __shfl_xor(w[q], o, 64);
    }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
#pragma unroll
        for (int q = 0; q < V; ++q) s_buf[q * (NT / 64) + (threadIdx.x >> 6)] = w[q];      // virtual wave = q * (NT / 64) + wave
    }
    __syncthreads();
    double r = s_buf[0];
    for (int i = 1; i < kTailLanes / 64; ++i) r += s_buf[i];
    return r;
}
struct VK { double v; long long k; };           // (value widened to double: exact for float, identity for double)
__device__ __forceinline__ VK comb_max(VK a, VK b) { return (b.v > a.v || (b.v == a.v && b.k < a.k)) ? b : a; }
__device__ __forceinline__ VK shfl_vk(VK a, int o) { VK r; r.v = __shfl_xor(a.v, o, 64); r.k = __shfl_xor(a.k, o, 64); return r; }

// The fold as a block-level routine of NT threads (the sums do not depend on NT: tail_sum).
template <typename T, int NT>
__device__ __forceinline__ void fuse_tail_body(const FuseTail<T>& ft) {
    __shared__ double s_d[kTailLanes / 64]; __shared__ double s_mv[NT / 64]; __shared__ long long s_mk[NT / 64];
    __shared__ int s_res[64]; __shared__ unsigned long long s_mask;
    constexpr int kTailThreads = NT, V = kTailLanes / NT;
    const int tid = threadIdx.x;
    long long t_prev = ft.prof ? wall_clock64() : 0;
#define TAIL_PROF(slot) do { if (ft.prof && tid == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&ft.prof[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
    const int rbw = tid < 63 ? ft.result_block[tid] : 0;       // (requested with the partials: a round trip of its own when it was asked for after them)
    double acc[2][V];
#pragma unroll
    for (int q = 0; q < V; ++q) { acc[0][q] = 0; acc[1][q] = 0; }
    VK best[2] = {{-DBL_MAX, 0x7fffffffffffffffll}, {-DBL_MAX, 0x7fffffffffffffffll}};
#pragma unroll
    for (int jb = 0; jb < 2; ++jb) {           // (constant trip count + unroll: the per-direction values stay in registers)
        if (jb >= ft.njobs) continue;
        if (ft.mode == FUSE_SUM) {
            // virtual thread vt = tid + q * NT: its strided partials (in order), then its limb term. The loads of a batch of kBatch trips
            // of all V virtual threads are requested together and only then added, in the fixed order: a plain `acc += p[i]` loop is a
            // chain of dependent L2 round trips (8 per thread at 1M-vs-1M, which WAS the 7 us of this fold).
            constexpr int kBatch = 8;
            const double* const fs = ft.flat_sum[jb]; const int nf = ft.nflat[jb];
            for (int b0 = 0; b0 < nf; b0 += kBatch * kTailLanes) {
                double v[V][kBatch];
#pragma unroll
                for (int q = 0; q < V; ++q)
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) { const int i = b0 + u * kTailLanes + tid + q * NT; v[q][u] = i < nf ? fs[i] : 0.0; }
#pragma unroll
                for (int q = 0; q < V; ++q)
#pragma unroll
                    for (int u = 0; u < kBatch; ++u) acc[jb][q] += v[q][u];          // (+ 0.0 past the end: the sum of distances is unchanged)
            }
#pragma unroll
            for (int q = 0; q < V; ++q) { const int vt = tid + q * NT; if (vt <= kAccLimbs) acc[jb][q] += exact_term(ft.limbs[jb], ft.special[jb], vt); }
        } else {
            // (the arg-max is order-independent -- value, then the smaller key --: batches of independent loads, see above)
            constexpr int kBatch = 8;
            const T* const fv = ft.flat_v[jb]; const long long* const fk = ft.flat_k[jb]; const int nf = ft.nflat[jb];
            for (int b0 = tid; b0 < nf; b0 += kBatch * kTailThreads) {
                T v[kBatch]; long long k[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) { const int i = min(b0 + u * kTailThreads, nf - 1); v[u] = fv[i]; k[u] = fk[i]; }      // (past the end: the last entry again)
#pragma unroll
                for (int u = 0; u < kBatch; ++u) { const VK c = {(double)v[u], k[u]}; best[jb] = comb_max(best[jb], c); }
            }
            const T* const wv = ft.wave_v[jb]; const long long* const wk = ft.wave_k[jb];
            for (int b0 = tid; b0 < ft.nwaves; b0 += kBatch * kTailThreads) {
                T v[kBatch]; long long k[kBatch];
#pragma unroll
                for (int u = 0; u < kBatch; ++u) {
                    const int i = min(b0 + u * kTailThreads, ft.nwaves - 1);
                    v[u] = wv[i]; k[u] = wk[i];
                }
#pragma unroll
                for (int u = 0; u < kBatch; ++u) { const VK c = {(double)v[u], k[u]}; best[jb] = comb_max(best[jb], c); }
            }
        }
    }
    TAIL_PROF(0);                                   // partials in, combined per thread
    if (tid == 0) s_mask = 0ull;
    if (ft.mode == FUSE_SUM && V == 1) {
        // tail_sum's order (butterfly inside a wave, the 16 wave sums added in order) for both directions behind ONE pair of barriers
        __shared__ double s_ws[2][kTailLanes / 64];
        double w[2] = {acc[0][0], acc[1][0]};
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) w[jb] += __shfl_xor(w[jb], o, 64);
        }
        __syncthreads();
        if ((tid & 63) == 0) { s_ws[0][tid >> 6] = w[0]; s_ws[1][tid >> 6] = w[1]; }
        __syncthreads();
        if (tid == 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                if (jb >= ft.njobs) continue;
                double r = s_ws[jb][0];
                for (int i = 1; i < kTailLanes / 64; ++i) r += s_ws[jb][i];
                *reinterpret_cast<double*>(&s_res[ft.w_sums + 2 * jb]) = r; s_mask |= 3ull << (ft.w_sums + 2 * jb);
            }
        }
    } else if (ft.mode == FUSE_SUM) {
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
            if (jb >= ft.njobs) continue;
            const double r = tail_sum<NT>(acc[jb], s_d);
            if (tid == 0) { *reinterpret_cast<double*>(&s_res[ft.w_sums + 2 * jb]) = r; s_mask |= 3ull << (ft.w_sums + 2 * jb); }
        }
    } else {
        // block winners of both directions in ONE pair of barriers (stage timers, PCU_HIP_PROF_TAIL: per direction a wave butterfly, two barriers and a
        // serial walk over the 16 wave winners in LDS took 7.5 us of the tail's 15.7): wave butterflies for both, one exchange through LDS, then wave 0's
        // lanes 0..15 / 16..31 fold the 16 wave winners of direction 0 / 1 by a 4-step butterfly and publish them
        VK win[2] = {best[0], best[1]};
        __shared__ double s_wv[2][NT / 64]; __shared__ long long s_wk[2][NT / 64]; __shared__ double s_fv[2]; __shared__ long long s_fk[2];
        static_assert(NT / 64 == 16, "16 wave winners per direction");
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) win[jb] = comb_max(win[jb], shfl_vk(win[jb], o));
        }
        if ((tid & 63) == 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) { s_wv[jb][tid >> 6] = win[jb].v; s_wk[jb][tid >> 6] = win[jb].k; }
        }
        __syncthreads();
        if (tid < 32) {
            const int jb = tid >> 4, w = tid & 15;
            VK v = {s_wv[jb][w], s_wk[jb][w]};
#pragma unroll
            for (int o = 8; o > 0; o >>= 1) v = comb_max(v, shfl_vk(v, o));
            if (w == 0) { s_fv[jb] = v.v; s_fk[jb] = v.k; }
        }
        __syncthreads();
#pragma unroll
        for (int jb = 0; jb < 2; ++jb) { win[jb].v = s_fv[jb]; win[jb].k = s_fk[jb]; }
        TAIL_PROF(1);                               // block winners
        if (ft.maxval) {            // the winners named by a value-only lane pass: their neighbours, both directions side by side
            bool need[2]; unsigned qp[2]; long long j[2]; int tie[2]; T d2[2];
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) { need[jb] = jb < ft.njobs && (win[jb].k & kKeyUnresolved) && win[jb].v > -(double)Limits<T>::max_v; qp[jb] = (unsigned)(win[jb].k & 0x07ffffffll); }
            if (need[0] || need[1]) {
                tail_resolve2<T, NT>(ft, need, qp, j, tie, d2);
#pragma unroll
                for (int jb = 0; jb < 2; ++jb) {
                    if (!need[jb]) continue;
                    // (the value the lane pass certified is this minimum, or something is wrong: then the flag sends the call to the row-based path)
                    const T val = ft.squared ? d2[jb] : (T)sqrt(d2[jb]);
                    if ((double)val != win[jb].v) tie[jb] = 1;
                    win[jb].k = (win[jb].k & ~0xffffffffll) | (j[jb] & 0x7fffffffll) | ((long long)tie[jb] << 31);
                }
            }
        }
        TAIL_PROF(2);                               // winners' neighbours resolved
        if (tid == 0) {
#pragma unroll
            for (int jb = 0; jb < 2; ++jb) {
                if (jb >= ft.njobs) continue;
                const VK v = win[jb];
                const int wv = ft.w_vals + jb * (int)(sizeof(T) / 4), wi = ft.w_ij + 4 * jb, wt = ft.w_tie + jb;
                *reinterpret_cast<T*>(&s_res[wv]) = (T)v.v; s_mask |= (sizeof(T) == 8 ? 3ull : 1ull) << wv;
                *reinterpret_cast<long long*>(&s_res[wi]) = v.k >> 32; *reinterpret_cast<long long*>(&s_res[wi + 2]) = v.k & 0x7fffffffll; s_mask |= 15ull << wi;
                s_res[wt] = (int)((v.k >> 31) & 1ll); s_mask |= 1ull << wt;
            }
        }
    }
    __syncthreads();
    TAIL_PROF(3);                                   // sums / result words ready
    if (ft.prof && tid == 0) atomicAdd((unsigned long long*)&ft.prof[7], 1ull);
#undef TAIL_PROF
    if (tid < 64) {
        if (tid < 63) ft.host_block[tid] = ((s_mask >> tid) & 1ull) ? s_res[tid] : rbw;
        __threadfence_system();
        if (tid == 63) __hip_atomic_store(&ft.host_block[63], (int)ft.seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
template <typename T>
__global__ __launch_bounds__(1024) void k_fuse_tail(const FuseTail<T> ft) { fuse_tail_body<T, 1024>(ft); }

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
struct PnormSide { const Pt4<T>* qsorted;     // the queries in the order of corr / d: their cell order -- or nullptr: corr / d are in the caller's ROW order, the queries are qpts
                   const T* qpts;
                   const T* tgt; const long long* corr; const T* d; int n; int nb;
                   const int* giveup;         // the direction's skew / unplaced-buckets flags: no rows exist (yet) when one is set
                   long long n_tgt; };        // rows of tgt: a correspondence outside [0, n_tgt) is an unwritten row (a straggler whose
                                              // result comes with the host-driven passes; this launch is then repeated) and is not followed

template <typename T>
__global__ __launch_bounds__(kBlock) void k_pnorm_pair(const PnormSide<T> s0, const PnormSide<T> s1, int pcode, double p, double* partial,
                                                       double* out_sums, unsigned* ticket, const int* result_block, int* host_block, unsigned seq) {
    const bool second = (int)blockIdx.x >= s0.nb;
    const PnormSide<T>& sd = second ? s1 : s0;
    const int bid = second ? (int)blockIdx.x - s0.nb : (int)blockIdx.x;
    double s = 0;
    const bool skip = sd.giveup && (sd.giveup[0] | sd.giveup[3]);      // ([3]: kLargeFlag of search.h)
    const int n_all = skip ? 0 : sd.n;
    // One query's norm from its correspondence, numpy.linalg.norm(tgt[corr] - q, ord, axis=-1) operation by operation in T. A query the
    // search could not match (non-finite coordinates: src/point_cloud_distance.cpp:90-93 writes -1) is paired as the reference's Python
    // tail pairs it -- `x[corrs]` with corrs == -1 is numpy's LAST row (__init__.py:112-113) -- and its difference vector carries the
    // infinities / NaNs into the mean exactly as there.
    auto from_corr = [&](int i) -> T {
        long long c = sd.corr[i];
        if (c == -1ll) c = sd.n_tgt - 1;
        if ((unsigned long long)c >= (unsigned long long)sd.n_tgt) c = 0;          // (see PnormSide::n_tgt)
        Pt4<T> q;
        if (sd.qsorted) q = sd.qsorted[i]; else { q.x = sd.qpts[3 * (size_t)i]; q.y = sd.qpts[3 * (size_t)i + 1]; q.z = sd.qpts[3 * (size_t)i + 2]; }
        const T a = sd.tgt[3 * c] - q.x, b = sd.tgt[3 * c + 1] - q.y, e = sd.tgt[3 * c + 2] - q.z;
        const T aa = a < 0 ? -a : a, ab = b < 0 ? -b : b, ae = e < 0 ? -e : e;
        const bool any_nan = a != a || b != b || e != e;
        T v;
        if (pcode == P_TWO) v = sqrt(((a * a) + (b * b)) + (e * e));
        else if (pcode == P_ONE) v = (aa + ab) + ae;
        else if (pcode == P_INF) { v = aa > ab ? aa : ab; v = v > ae ? v : ae; if (any_nan) v = a + b + e; }        // (numpy's max / min propagate NaN)
        else if (pcode == P_NINF) { v = aa < ab ? aa : ab; v = v < ae ? v : ae; if (any_nan) v = a + b + e; }
        else if (pcode == P_ZERO) v = (T)((a != 0) + (b != 0) + (e != 0));
        else v = (T)pow((double)(T)((T)pow((double)aa, p) + (T)pow((double)ab, p)) + (double)(T)pow((double)ae, p), 1.0 / p);
        return v;
    };
    const int n_vec = pcode == P_TWO ? (n_all & ~3) : 0;       // p = 2: the distances themselves, four per 16/32-byte load
    for (int i = 4 * (bid * kBlock + (int)threadIdx.x); i < n_vec; i += 4 * sd.nb * kBlock) {
        T v0 = sd.d[i], v1 = sd.d[i + 1], v2 = sd.d[i + 2], v3 = sd.d[i + 3];
        if (v0 < (T)0 || v1 < (T)0 || v2 < (T)0 || v3 < (T)0) {         // -1.0: an unmatched query (rare)
            if (v0 < (T)0) v0 = from_corr(i);
            if (v1 < (T)0) v1 = from_corr(i + 1);
            if (v2 < (T)0) v2 = from_corr(i + 2);
            if (v3 < (T)0) v3 = from_corr(i + 3);
        }
        s += ((double)v0 + (double)v1) + ((double)v2 + (double)v3);
    }
    for (int i = n_vec + bid * kBlock + (int)threadIdx.x; i < n_all; i += sd.nb * kBlock) {
        T v;
        if (pcode == P_TWO) { v = sd.d[i]; if (v < (T)0) v = from_corr(i); }
        else v = from_corr(i);
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
