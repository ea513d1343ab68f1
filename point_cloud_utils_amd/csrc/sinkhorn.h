// csrc/sinkhorn.h -- dense pairwise distances and Sinkhorn iterations: the one dense N x M workload of the
// path's neighbourhood. Designated bound: HBM (an iteration needs the (nb, m, n) cost matrix for both updates: 2 m n s bytes when
// read twice, m n s with k_sink_iter); measured, an iteration is bound by its arithmetic -- per element three IEEE divisions by eps
// and two exp(), kept as the reference computes them -- at ~1.7 TB/s of algorithmic traffic, so halving the reads (k_sink_iter)
// bought 2 %, and keeping the next row's loads in flight across the block reductions nothing more.
//
// Replaces point_cloud_utils/_sinkhorn.py (pure numpy in the reference):
//   pairwise_distances :4-33    M[b,i,j] = || a[b,i,:] - b[b,j,:] ||_p   (numpy.linalg.norm(..., axis=-1, ord=p))
//   sinkhorn           :36-130  log-domain Sinkhorn: u <- eps (log a - LSE_j((-M + v_j) / eps)), v <- eps (log b - LSE_i((-M + u_i) / eps)),
//                               stop when both max_b sum |du|, max_b sum |dv| < stop_thresh; P = exp((-M + u_i + v_j) / eps)
//   earth_movers_distance :133-156  (P * M).sum()
// Arithmetic is done in the input dtype with the reference's operation order per element ((-M + v) / eps, exp(x - max), ...);
// what differs is the ORDER OF THE SUMS (numpy: pairwise blocks; here: fixed trees over a block, and a streaming
// log-sum-exp for the column pass so that the matrix is read once per pass). Tolerance: 1e-5 relative for float32, 1e-11 for
// float64 on u, v, P (tests/test_gpu_sinkhorn.py). All reductions run in a fixed order: results are reproducible run to run.
#pragma once
#include "pcu_types.h"
#include "grid.h"
#include "reduce.h"

namespace pcu {

// ---- pairwise distances ----------------------------------------------------------------------------------------------------
// one thread per output element; a block covers 64 consecutive j of 4 consecutive i (coalesced stores, the a rows and b rows it
// reads are shared through L1)
template <typename T>
__global__ __launch_bounds__(256) void k_pairwise(const T* __restrict__ a, const T* __restrict__ b, int m, int n, int d, int pcode, double p, T* __restrict__ out, int col_blocks) {
    // (rows x column blocks are folded into gridDim.x -- 2^31-1 blocks -- because gridDim.y / .z end at 65535: 100k points against 100
    // centroids is a legitimate call)
    const int bx = (int)(blockIdx.x % (unsigned)col_blocks), by = (int)(blockIdx.x / (unsigned)col_blocks);
    const int j = bx * 64 + (threadIdx.x & 63), i = by * 4 + (threadIdx.x >> 6), bt = blockIdx.y;
    if (i >= m || j >= n) return;
    const T* ai = a + ((size_t)bt * m + i) * d; const T* bj = b + ((size_t)bt * n + j) * d;
    T acc = pcode == P_NINF ? (T)INFINITY : (T)0;
    for (int c = 0; c < d; ++c) {
        const T x = ai[c] - bj[c];
        const T ax = x < 0 ? -x : x;
        if (pcode == P_TWO) acc += x * x;                       // sqrt(add.reduce(x * x)) -- numpy's 2-norm of a real vector
        else if (pcode == P_ONE) acc += ax;
        else if (pcode == P_INF) acc = ax > acc ? ax : acc;
        else if (pcode == P_NINF) acc = ax < acc ? ax : acc;
        else if (pcode == P_ZERO) acc += (T)(x != 0);
        else acc += (T)pow((double)ax, p);
    }
    if (pcode == P_TWO) acc = sqrt(acc);
    else if (pcode == P_GEN) acc = (T)pow((double)acc, 1.0 / p);
    out[((size_t)bt * m + i) * n + j] = acc;
}

// ---- Sinkhorn ---------------------------------------------------------------------------------------------------------------
template <typename T>
struct SinkArgs {
    const T* a; const T* b; const T* M;     // (nb,m), (nb,n), (nb,m,n)
    T* u; T* v; T* du; T* dv;               // potentials and |change| of the current iteration
    int nb, m, n;
    T eps;
    int* done;                              // device flag: set once both errors are below the threshold; later launches exit at once
};

template <typename T> __device__ __forceinline__ T blk_max(T v, T* s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { const T w = __shfl_xor(v, o, 64); v = w > v ? w : v; }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = s[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r = s[w] > r ? s[w] : r;
    return r;
}
template <typename T> __device__ __forceinline__ T blk_sum(T v, T* s) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s[threadIdx.x >> 6] = v;
    __syncthreads();
    T r = s[0];
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) r += s[w];
    return r;
}

// u update (:113-114): one block per (batch, row). Two sweeps over the row (max, then sum of exp(x - max)) as
// stabilized_log_sum_exp does (:106-110); the second sweep re-reads the row from cache.
template <typename T>
__global__ __launch_bounds__(256) void k_sink_rows(const SinkArgs<T> s) {
    if (*s.done) return;
    __shared__ T sm[4];
    const int i = blockIdx.x, bt = blockIdx.y;
    const T* Mrow = s.M + ((size_t)bt * s.m + i) * s.n; const T* v = s.v + (size_t)bt * s.n;
    T mx = -(T)INFINITY;
    for (int j = threadIdx.x; j < s.n; j += 256) { const T x = (-Mrow[j] + v[j]) / s.eps; mx = x > mx ? x : mx; }
    mx = blk_max(mx, sm);
    T acc = 0;
    for (int j = threadIdx.x; j < s.n; j += 256) { const T x = (-Mrow[j] + v[j]) / s.eps; acc += exp(x - mx); }
    acc = blk_sum(acc, sm);
    if (threadIdx.x == 0) {
        const T lse = log(acc) + mx;
        const size_t o = (size_t)bt * s.m + i;
        const T un = s.eps * (log(s.a[o]) - lse);
        const T d = s.u[o] - un;
        s.du[o] = d < 0 ? -d : d; s.u[o] = un;
    }
}
// v update (:116-117): a block owns 32 consecutive columns of one SLAB of rows (grid.z slabs, so that even a single batch of a few
// thousand columns fills the chip); its 32 thread rows stride over the slab's rows. The column sums use a streaming
// log-sum-exp (running maximum, rescaled sum), so M is read once; the 32 partial (max, sum) pairs of a column are merged in LDS
// into the slab's partial, and k_sink_cols_finish merges the slabs in a fixed order.
template <typename T>
__global__ __launch_bounds__(1024) void k_sink_cols(const SinkArgs<T> s, T* __restrict__ part_mx, T* __restrict__ part_sum, int rows_per_slab) {
    if (*s.done) return;
    __shared__ T s_mx[32][33], s_sum[32][33];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5, j = blockIdx.x * 32 + c, bt = blockIdx.y, slab = blockIdx.z;
    const T* Mb = s.M + (size_t)bt * s.m * s.n; const T* u = s.u + (size_t)bt * s.m;
    const int i0 = slab * rows_per_slab, i1 = min(s.m, i0 + rows_per_slab);
    T mx = -(T)INFINITY, acc = 0;
    if (j < s.n)
        for (int i = i0 + r; i < i1; i += 32) {
            const T x = (-Mb[(size_t)i * s.n + j] + u[i]) / s.eps;
            // (x = -inf -- a forbidden assignment M = +inf, or u = -inf from a zero weight -- adds exp(-inf) = 0, as in the reference's
            // max-subtracted sum; exp(x - mx) would be exp(NaN) while the running maximum is still -inf)
            if (x > mx) { acc = acc * exp(mx - x) + (T)1; mx = x; } else if (x != -(T)INFINITY) acc += exp(x - mx);
        }
    s_mx[r][c] = mx; s_sum[r][c] = acc;
    __syncthreads();
    if (r == 0 && j < s.n) {
        T M2 = s_mx[0][c];
        for (int k = 1; k < 32; ++k) M2 = s_mx[k][c] > M2 ? s_mx[k][c] : M2;
        T tot = 0;
        for (int k = 0; k < 32; ++k) tot += s_mx[k][c] == -(T)INFINITY ? (T)0 : s_sum[k][c] * exp(s_mx[k][c] - M2);
        const size_t o = ((size_t)slab * s.nb + bt) * s.n + j;
        part_mx[o] = M2; part_sum[o] = tot;
    }
}
// One launch per iteration for n <= 256 * CPT columns: the u update AND the column sums of the v update from ONE read of M.
// A block owns `rows_per_block` consecutive rows of one batch; thread t owns the columns t, t + 256, ... (CPT of them). Per row:
// the thread loads its CPT elements (coalesced), the block reduces max and sum of exp((-M + v) / eps) as k_sink_rows does and
// gets the row's new u -- which is all the v update needs from that row (the reference updates v with the NEW u, :116-117) --
// so the same registers go straight into the thread's running (max, sum) of exp((-M + u_new) / eps) for its columns. The
// block's column partials are then merged over the blocks by k_sink_cols_finish. HBM traffic per iteration: m n s bytes for M
// (half of the two-pass scheme) + 2 x the partials (n x blocks-per-batch x 2 s).
// (The block reductions use a raw LDS barrier -- s_waitcnt lgkmcnt(0) + s_barrier, no fence on global memory -- so that the NEXT
// row's loads, issued at the top of the trip, stay in flight across them: __syncthreads() would wait for every outstanding
// vector-memory operation, and a block is a chain of row trips.)
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
template <typename T, int CPT>
__global__ __launch_bounds__(256) void k_sink_iter(const SinkArgs<T> s, T* __restrict__ part_mx, T* __restrict__ part_sum, int rows_per_block) {
    if (*s.done) return;
    __shared__ T s_red[2][2][4];           // [row parity][max / sum][wave]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, slab = blockIdx.x, bt = blockIdx.y;
    const int i0 = slab * rows_per_block, i1 = min(s.m, i0 + rows_per_block);
    const T* Mb = s.M + (size_t)bt * s.m * s.n; const T* v = s.v + (size_t)bt * s.n;
    T vj[CPT], cmx[CPT], cacc[CPT], mv[CPT];
#pragma unroll
    for (int k = 0; k < CPT; ++k) { const int j = tid + 256 * k; vj[k] = j < s.n ? v[j] : (T)0; cmx[k] = -(T)INFINITY; cacc[k] = 0; }
    if (i0 < i1) {
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const int j = tid + 256 * k; mv[k] = j < s.n ? Mb[(size_t)i0 * s.n + j] : (T)INFINITY; }
    }
    for (int i = i0; i < i1; ++i) {
        T nx[CPT];                          // next row, requested now
        const int in = min(i + 1, i1 - 1);
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const int j = tid + 256 * k; nx[k] = j < s.n ? Mb[(size_t)in * s.n + j] : (T)INFINITY; }
        const size_t o = (size_t)bt * s.m + i;
        const T la = log(s.a[o]), uo = s.u[o];
        T (&red)[2][4] = s_red[i & 1];
        T mx = -(T)INFINITY;
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const T x = (-mv[k] + vj[k]) / s.eps; mx = x > mx ? x : mx; }        // (a column past n: x = -inf)
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) { const T w = __shfl_xor(mx, of, 64); mx = w > mx ? w : mx; }
        if (lane == 0) red[0][wave] = mx;
        lds_barrier();
        mx = red[0][0];
#pragma unroll
        for (int w = 1; w < 4; ++w) mx = red[0][w] > mx ? red[0][w] : mx;
        T acc = 0;
#pragma unroll
        for (int k = 0; k < CPT; ++k) { const T x = (-mv[k] + vj[k]) / s.eps; acc += exp(x - mx); }
#pragma unroll
        for (int of = 32; of > 0; of >>= 1) acc += __shfl_xor(acc, of, 64);
        if (lane == 0) red[1][wave] = acc;
        lds_barrier();
        acc = ((red[1][0] + red[1][1]) + red[1][2]) + red[1][3];
        const T un = s.eps * (la - (log(acc) + mx));
        if (tid == 0) { const T d = uo - un; s.du[o] = d < 0 ? -d : d; s.u[o] = un; }
#pragma unroll
        for (int k = 0; k < CPT; ++k) {
            const T x = (-mv[k] + un) / s.eps;
            if (x > cmx[k]) { cacc[k] = cacc[k] * exp(cmx[k] - x) + (T)1; cmx[k] = x; } else if (x != -(T)INFINITY) cacc[k] += exp(x - cmx[k]);     // (see k_sink_cols)
            mv[k] = nx[k];
        }
    }
#pragma unroll
    for (int k = 0; k < CPT; ++k) {
        const int j = tid + 256 * k;
        if (j < s.n) { const size_t o = ((size_t)slab * s.nb + bt) * s.n + j; part_mx[o] = cmx[k]; part_sum[o] = cacc[k]; }
    }
}
template <typename T>
__global__ __launch_bounds__(256) void k_sink_cols_finish(const SinkArgs<T> s, const T* __restrict__ part_mx, const T* __restrict__ part_sum, int n_slabs) {
    if (*s.done) return;
    const int j = blockIdx.x * 256 + threadIdx.x, bt = blockIdx.y;
    if (j >= s.n) return;
    const size_t o = (size_t)bt * s.n + j, stride = (size_t)s.nb * s.n;
    T M2 = part_mx[o];
    for (int k = 1; k < n_slabs; ++k) { const T v = part_mx[o + k * stride]; M2 = v > M2 ? v : M2; }
    T tot = 0;
    for (int k = 0; k < n_slabs; ++k) { const T v = part_mx[o + k * stride]; tot += v == -(T)INFINITY ? (T)0 : part_sum[o + k * stride] * exp(v - M2); }
    const T lse = log(tot) + M2;
    const T vn = s.eps * (log(s.b[o]) - lse);
    const T d = s.v[o] - vn;
    s.dv[o] = d < 0 ? -d : d; s.v[o] = vn;
}
// The same merge for many slabs (k_sink_iter: one per block of rows): a block owns 32 columns, its 32 thread rows stride over the
// slabs with a streaming merge, then merge in LDS as k_sink_cols does.
template <typename T>
__global__ __launch_bounds__(1024) void k_sink_cols_merge(const SinkArgs<T> s, const T* __restrict__ part_mx, const T* __restrict__ part_sum, int n_slabs) {
    if (*s.done) return;
    __shared__ T s_mx[32][33], s_sum[32][33];
    const int c = threadIdx.x & 31, r = threadIdx.x >> 5, j = blockIdx.x * 32 + c, bt = blockIdx.y;
    const size_t o = (size_t)bt * s.n + j, stride = (size_t)s.nb * s.n;
    T mx = -(T)INFINITY, acc = 0;
    if (j < s.n)
        for (int k = r; k < n_slabs; k += 32) {
            const T pm = part_mx[o + k * stride], ps = part_sum[o + k * stride];
            if (pm == -(T)INFINITY) continue;
            if (pm > mx) { acc = acc * exp(mx - pm) + ps; mx = pm; } else acc += ps * exp(pm - mx);
        }
    s_mx[r][c] = mx; s_sum[r][c] = acc;
    __syncthreads();
    if (r == 0 && j < s.n) {
        T M2 = s_mx[0][c];
        for (int k = 1; k < 32; ++k) M2 = s_mx[k][c] > M2 ? s_mx[k][c] : M2;
        T tot = 0;
        for (int k = 0; k < 32; ++k) tot += s_mx[k][c] == -(T)INFINITY ? (T)0 : s_sum[k][c] * exp(s_mx[k][c] - M2);
        const T lse = log(tot) + M2;
        const T vn = s.eps * (log(s.b[o]) - lse);
        const T d = s.v[o] - vn;
        s.dv[o] = d < 0 ? -d : d; s.v[o] = vn;
    }
}
// err_u = max_b sum_i |du| , err_v likewise (:119-120); done when both are below the threshold (:122-123). One block.
template <typename T>
__global__ __launch_bounds__(1024) void k_sink_check(const SinkArgs<T> s, T stop_thresh, int* iters) {
    if (*s.done) return;
    __shared__ T sm[16];
    T eu = 0, ev = 0;
    for (int bt = 0; bt < s.nb; ++bt) {
        T a = 0, b = 0;
        for (int i = threadIdx.x; i < s.m; i += 1024) a += s.du[(size_t)bt * s.m + i];
        for (int j = threadIdx.x; j < s.n; j += 1024) b += s.dv[(size_t)bt * s.n + j];
        a = blk_sum(a, sm); b = blk_sum(b, sm);
        eu = a > eu ? a : eu; ev = b > ev ? b : ev;
    }
    if (threadIdx.x == 0) { *iters += 1; if (eu < stop_thresh && ev < stop_thresh) *s.done = 1; }
}
// P = exp((-M + u_i + v_j) / eps) (:125-127)
template <typename T>
__global__ __launch_bounds__(256) void k_sink_plan(const SinkArgs<T> s, T* __restrict__ P, int col_blocks) {
    const int bx = (int)(blockIdx.x % (unsigned)col_blocks), i = (int)(blockIdx.x / (unsigned)col_blocks);      // (folded like k_pairwise's grid)
    const int j = bx * 256 + threadIdx.x, bt = blockIdx.y;
    if (j >= s.n) return;
    const size_t o = ((size_t)bt * s.m + i) * s.n + j;
    P[o] = exp(((-s.M[o] + s.u[(size_t)bt * s.m + i]) + s.v[(size_t)bt * s.n + j]) / s.eps);
}
// sum of x * y in double (earth_movers_distance's (P * M).sum(), :156): per-block partials, folded by the host
template <typename T>
__global__ __launch_bounds__(256) void k_dot_partial(const T* __restrict__ x, const T* __restrict__ y, size_t count, double* __restrict__ partial) {
    __shared__ double sm[4];
    double acc = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < count; i += (size_t)gridDim.x * 256) acc += (double)(T)(x[i] * y[i]);
    acc = blk_sum(acc, sm);
    if (threadIdx.x == 0) partial[blockIdx.x] = acc;
}

}  // namespace pcu
