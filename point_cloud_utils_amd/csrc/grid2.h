// csrc/grid2.h -- the one-pass index build of whole-call indexes, second form (round 4). Same role and same result as grid.h's
// k_bbox_partial -> k_bucket_onepass -> k_bucket_sort chain (the kd-tree build of the reference, src/point_cloud_distance.cpp:41-42,
// nanoflann.hpp:1363-1375, replaced by a counting sort into snake cell order), re-cut around what that chain spent its time on
// (profiles/r03_pmc.txt: 62.5 us and 218 MB of HBM traffic for 24 MB of input; profiles/r04_build_ab.txt has the steps):
//
//   no bbox pass      The grid only decides which candidates a query looks at -- border cells hold whatever lies outside its range (cell_coord
//                     clamps, the searches know) -- so it does not need the exact bounding box: every scatter block lays it out itself
//                     from a stratified SAMPLE of 1024 points (range = the sample's box clipped at mean +- 3 sigma, as before; identical
//                     arithmetic on identical inputs: identical grids), while its own points are in flight. The 24 MB read of
//                     k_bbox_partial (10.7 us) and its launch are gone. The exact box and the non-finite classification -- needed by the
//                     certification bounds, the kd-tree root and the input checks -- are taken from the coordinates the scatter blocks
//                     hold anyway: one partial per block, folded by an extra block of the sort launch.
//   k_bucket_onepass3 8192 points per block as before (LDS histogram over buckets, one returning atomic per (block, non-empty bucket)),
//                     but the block's records are staged in LDS by bucket and written run by run, a wave per run, every run rounded up
//                     to 8 records = one 128-byte line with HOLE records (row id -1): every line of `tmp` leaves the CU whole.
//                     (Per-thread 16-byte stores in input order reached HBM as partial-line writes, 70 MB for 32 MB of records -- and
//                     still 72 MB with aligned runs: what the L2 combines is a store instruction, not a line's history.)
//   k_bucket_sort2    one block per bucket: LDS histogram over its cells (holes skipped), scan -> cell_start, the records -- kept in
//                     registers since their single read -- go straight to their sorted slots of the LDS stage (no parking + in-place
//                     permutation) and leave as coalesced copies: the coordinates-only stream, the 32-bit row ids, and the Pt4 records
//                     only if the call has a kernel that reads them (`want_pt4`; the fused k = 1 calls do not: 28 -> 16 bytes written
//                     per point).
//   no memset         The bucket fill words live in the CONTEXT, twice: a build uses one set, which its predecessor left zeroed, and its
//                     sort launch zeroes the other one for its successor (calls of a context do not overlap). The call's result block is
//                     zeroed by the same extra blocks.
//
// Any valid cell order gives the same search results; nothing here can change a result.
#pragma once
#include "grid.h"

namespace pcu {

constexpr int kPrepSamples = 1024;          // (stratified: one point from each of 1024 equal slices of the cloud, position hashed; every scatter
                                            // block reads the same sample, so its size is L2 traffic: 4096 scattered points per block cost 6 us)
constexpr int kXPartStride = 8;             // per scatter block: [0..2] finite min, [3..5] finite max, [6] non-finite flags (bbox_body's bits)
constexpr int kStagedMaxBuckets = 2048;     // the scatter's bucket tables (12 bytes each) sit beside its 128 KB stage
constexpr int kFillWords = 2 * kStagedMaxBuckets + 8;      // one set of fill words: [side][bucket], then [2 * kStagedMaxBuckets + side] = slot-overflow flag

// A wave-uniform value pinned into scalar registers (the compiler cannot prove uniformity of values read back from LDS or from global memory
// that the kernel also writes, and then keeps them in vector registers or re-reads them at every use).
__device__ __forceinline__ int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ float uniform(float v) { return __int_as_float(__builtin_amdgcn_readfirstlane(__float_as_int(v))); }
__device__ __forceinline__ double uniform(double v) {
    const long long b = __double_as_longlong(v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)b), hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __longlong_as_double((long long)(((unsigned long long)hi << 32) | lo));
}
// What cell_linear needs of a grid, in scalar registers.
template <typename T>
struct CellMap {
    T org[3], inv_h; int G[3];
    __device__ __forceinline__ explicit CellMap(const GridParams<T>& g) {
#pragma unroll
        for (int j = 0; j < 3; ++j) { org[j] = uniform(g.org[j]); G[j] = uniform(g.G[j]); }
        inv_h = uniform(g.inv_h);
    }
    __device__ __forceinline__ unsigned cell(T x, T y, T z) const {
        const int cx = cell_coord(x, org[0], inv_h, G[0]), cy = cell_coord(y, org[1], inv_h, G[1]), cz = cell_coord(z, org[2], inv_h, G[2]);
        return (unsigned)row_run_lo(G[0], grid_row(G[1], cy, cz), cx, cx);
    }
};

__device__ __forceinline__ unsigned hash32(unsigned x) { x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16; return x; }

// The geometry of a grid as the context keeps it from one call to the next (round 6): the layout of call i is computed from call i's sample by
// the sort launch's extra block -- off the critical path -- and USED by call i + 1, whose scatter blocks then start keying as soon as their
// points arrive instead of first reading a sample and laying the grid out (8 of a block's 25 us). Any layout gives the same search results; one
// that no longer fits the data (range moved by more than 5 % of the extent, cell edge by more than 4 %) is refused by the same extra block
// (GridParams::has_large bit 8: the searches give up, the host restarts the call with a fresh layout) BEFORE the balance heuristics can mistake
// a stale grid for uneven data.
template <typename T> struct GridGeo { T org[3], h, inv_h, slack[3], rlo[3], rhi[3]; int G[3], ncells; };
constexpr int kGeoStale = 8;        // GridParams::has_large bit (search.h: index_not_ready)

template <typename T>
struct Build2Side {
    const T* pts; int n; GridParams<T>* gp; int shift;
    double occupancy; int max_cells; double h_want;       // grid layout (grid.h: grid_layout)
    unsigned long long* fill;       // per bucket: low word = records reserved in its slot (runs padded to 8), high word = points among them
    unsigned long long* ovf;        // != 0: some bucket overflowed its slot (the index is unusable; -> GridParams::has_large = 2)
    Pt4<T>* tmp; unsigned cap;      // bucket b owns tmp[b * cap, (b + 1) * cap)
    T* xpartial; int n_xpart;       // exact bbox / non-finite partials, one per scatter block
    unsigned* cell_start; Pt4<T>* sorted; unsigned* pos_of; int want_pt4;
    unsigned* n_large;              // (the two-pass build's count of over-full buckets: none here, but k_bucket_large may be launched on this index)
    unsigned long long* zero_next; int n_zero_next;       // the other set of fill words, zeroed for the context's next build (side 0 only)
    unsigned* zero2; int n_zero2;   // the call's result block (side 0 only)
    long long* prof;                // diagnostics (PCU_HIP_PROF_BUILD2): per-stage time of every block's thread 0, summed; 100 MHz ticks
    // SHARED GRID (round 6; two-sided calls between clouds of comparable size): both clouds of the call are laid over ONE grid -- same origin,
    // cell edge and cell counts -- so that a query's cell in its own cloud's order IS its cell in the dataset's grid and a block of consecutive
    // queries needs a compact box of dataset rows (search_brick.h stages that box in LDS). The layout then comes from 512 samples of EACH cloud,
    // read in the same order by the blocks of both sides (identical arithmetic on identical inputs: identical grids), for n_layout = the larger
    // cloud's size. spts1 == nullptr: the cloud's own grid from its own 1024 samples, as before.
    const T* spts0; int sn0; const T* spts1; int sn1; int n_layout;
    const GridGeo<T>* geo_in;       // nullable: lay the grid out as the context's previous call did (see GridGeo)
    GridGeo<T>* geo_out;            // nullable: where the sort launch's extra block leaves THIS call's layout for the next one
};

// A column of per-thread values (NT threads) folded by one wave: lane l takes the values of threads l, l + 64, ...; the result is valid in lane 63.
// kind 0: min, 1: max, 2: sum, 3: bitwise or (values as raw bits of T's unsigned twin are not needed: flags are stored as exact small T values).
template <typename T, int NT>
__device__ __forceinline__ T fold_column(const T* col, int kind, int lane) {
    T r = col[lane];
    for (int i = 1; i < NT / 64; ++i) {
        const T o = col[lane + 64 * i];
        r = kind == 0 ? (o < r ? o : r) : (kind == 1 ? (o > r ? o : r) : r + o);
    }
    return kind == 0 ? wave_min63(r) : (kind == 1 ? wave_max63(r) : wave_sum63(r));
}

// The stratified sample of a build (one point per thread of a 1024-thread block) and the grid layout it gives. Shared by the scatter blocks
// (which lay the grid out themselves when the context has no layout to hand down) and the sort launch's extra block (which computes the layout
// the NEXT call will use). Identical arithmetic on identical inputs: identical grids, whoever computes them.
template <typename T> struct SampleOf { T v[3]; T pv[3]; bool on; };        // (raw loads: nothing of them is looked at before the caller has issued its own loads)
template <typename T>
__device__ __forceinline__ SampleOf<T> sample_request(const Build2Side<T>& a, const int n, const int tid) {
    struct __attribute__((packed, aligned(4))) P3 { T v[3]; };
    const P3* const pts3 = reinterpret_cast<const P3*>(a.pts);
    const bool shared = a.spts1 != nullptr;       // (both clouds hold >= kPrepSamples points then: the host's rule)
    const bool all = !shared && n < kPrepSamples; // a cloud smaller than the sample: every point
    const int S = all ? n : kPrepSamples;
    static_assert(kPrepSamples == kBkThreads && kPrepSamples == kSortThreads, "one sample per thread");
    P3 sv, pv;
    if (!shared) {
        const int j = min(tid, S - 1);            // sample j: one point of the slice [j n / S, (j + 1) n / S), position hashed
        size_t i = (size_t)j;
        if (!all) {
            static_assert(kPrepSamples == 1024, "slice bounds by a shift");
            const unsigned long long b0 = ((unsigned long long)j * (unsigned long long)n) >> 10, b1 = ((unsigned long long)(j + 1) * (unsigned long long)n) >> 10;
            i = (size_t)b0 + (size_t)(((unsigned long long)hash32((unsigned)j) * (unsigned long long)(unsigned)(b1 - b0)) >> 32);
        }
        sv = pts3[i]; pv = pts3[0];
    } else {                                      // threads 0..511: the call's first cloud, 512..1023: its second -- on BOTH sides
        const int half = tid >> 9, j = tid & 511;
        const P3* const src = reinterpret_cast<const P3*>(half ? a.spts1 : a.spts0);
        const int ns = half ? a.sn1 : a.sn0;
        const unsigned long long b0 = ((unsigned long long)j * (unsigned long long)ns) >> 9, b1 = ((unsigned long long)(j + 1) * (unsigned long long)ns) >> 9;
        const size_t i = (size_t)b0 + (size_t)(((unsigned long long)hash32((unsigned)tid) * (unsigned long long)(unsigned)(b1 - b0)) >> 32);
        sv = src[i]; pv = reinterpret_cast<const P3*>(a.spts0)[0];
    }
    SampleOf<T> r;
#pragma unroll
    for (int j = 0; j < 3; ++j) { r.v[j] = sv.v[j]; r.pv[j] = pv.v[j]; }
    r.on = tid < S;
    return r;
}
// (every thread of the 1024-thread block; s_col: 13 x 1024 scalars of scratch; two barriers inside, one at the end. gp_out: geometry fields +
// provisional gmin / gmax = the sample's box; rng6: the range the grid was laid over, valid in thread 0)
template <typename T>
__device__ __forceinline__ void layout_from_sample(const Build2Side<T>& a, const int n, const SampleOf<T>& smp, T* const s_col, T* const s_fin, GridParams<T>* const gp_out, T (&rng6)[6]) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const bool shared = a.spts1 != nullptr;
    {
        T piv[3];
#pragma unroll
        for (int j = 0; j < 3; ++j) piv[j] = ((smp.pv[j] < (T)0 ? -smp.pv[j] : smp.pv[j]) <= Limits<T>::max_v) ? smp.pv[j] : (T)0;
        T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
        T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
        T s1[3] = {(T)0, (T)0, (T)0}, s2[3] = {(T)0, (T)0, (T)0}, cnt = (T)0;
        if (smp.on) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const T x = smp.v[c];
                const bool fin = (x < (T)0 ? -x : x) <= Limits<T>::max_v;
                lo[c] = fin ? x : lo[c]; hi[c] = fin ? x : hi[c];
                const T dv = fin ? x - piv[c] : (T)0;
                s1[c] = dv; s2[c] = dv * dv;
            }
            cnt = (T)1;
        }
        // 13 columns of 1024 values, one wave each (16 waves x 13 DPP chains on ONE CU cost 3 us; this is 0.5)
        const T m[13] = {lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], cnt, s1[0], s1[1], s1[2], s2[0], s2[1], s2[2]};
#pragma unroll
        for (int q = 0; q < 13; ++q) s_col[q * kBkThreads + tid] = m[q];
        __syncthreads();
        if (wave < 13) { const T r = fold_column<T, kBkThreads>(s_col + wave * kBkThreads, wave < 3 ? 0 : (wave < 6 ? 1 : 2), lane); if (lane == 63) s_fin[wave] = r; }
        __syncthreads();
        if (tid == 0) {
            const double m0 = (double)s_fin[6];
            const double inv0 = m0 > 0 ? 1.0 / m0 : 0.0;
            T rlo[3], rhi[3];
            for (int j = 0; j < 3; ++j) {
                T l = s_fin[j], h = s_fin[3 + j];
                if (!(l <= h)) { l = 0; h = 0; }          // no finite value sampled in this column
                gp_out->gmin[j] = l; gp_out->gmax[j] = h;       // (provisional: the exact box arrives with k_bucket_sort2's extra blocks)
                rlo[j] = l; rhi[j] = h;
                if (m0 > 0) {
                    const double m1 = (double)s_fin[7 + j] * inv0, var = (double)s_fin[10 + j] * inv0 - m1 * m1, sd = var > 0 ? (double)sqrtf((float)var) : 0.0, mu = (double)piv[j] + m1;
                    const T p = (T)(mu - kCoreSigmas * sd), q = (T)(mu + kCoreSigmas * sd);
                    if (p <= q) {
                        if (p > rlo[j] && p < h) rlo[j] = p;
                        if (q < rhi[j] && q > rlo[j]) rhi[j] = q;
                    }
                }
                rng6[j] = rlo[j]; rng6[3 + j] = rhi[j];
            }
            grid_layout<T>(gp_out, rlo, rhi, shared ? a.n_layout : n, a.occupancy, a.max_cells, a.h_want);
        }
        __syncthreads();
    }
}

template <typename T> struct StagedPts { static constexpr int n = sizeof(T) == 4 ? 8 : 4; };       // points per thread: 32-byte f64 records halve the block
// Dynamic LDS of a scatter block of kBkThreads x pts points with tables for nbcap buckets: the stage + three words per bucket; never less than
// the 13 statistics columns of the layout stage (which lie over stage and tables: they are done before the tables are zeroed).
template <typename T>
static size_t onepass3_lds_bytes(int pts = StagedPts<T>::n, int nbcap = kStagedMaxBuckets) {
    return std::max((size_t)kBkThreads * pts * sizeof(Pt4<T>) + (size_t)nbcap * 12, (size_t)13 * kBkThreads * sizeof(T));
}

// (the side's arguments are read through an index into the kernel-argument segment: selecting between two by-value structs by reference makes the
// compiler copy the chosen one to scratch -- 472 bytes per lane and every field a scratch load, which doubled the layout stage when the struct grew)
template <typename T> struct Build2Args { Build2Side<T> a[2]; };
// PTS points per thread (round 6: chosen by the host so that a launch has about a block per CU or more -- 8192-point blocks left three quarters of
// the GPU empty on config 4's 262 144-point clouds -- and, below 8, so that two blocks fit a CU's LDS and overlap their phases); nbcap: bucket
// slots of the three tables (>= the grid's bucket count, a multiple of 64).
template <typename T, int PTS>
__global__ __launch_bounds__(kBkThreads) void k_bucket_onepass3(const Build2Args<T> p, int nb0, int nbcap) {
    constexpr int BLOCK_PTS = kBkThreads * PTS;
    extern __shared__ __attribute__((aligned(32))) unsigned char s_dyn[];
    Pt4<T>* const s_stage = reinterpret_cast<Pt4<T>*>(s_dyn);
    T* const s_col = reinterpret_cast<T*>(s_dyn);                    // [13][kBkThreads] columns of per-thread statistics: the stage is not in use yet
    unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_dyn + (size_t)BLOCK_PTS * sizeof(Pt4<T>));
    unsigned* const s_lbase = s_cnt + nbcap;
    unsigned* const s_gbase = s_lbase + nbcap;
    __shared__ GridParams<T> s_gp;
    __shared__ T s_fin[13];
    __shared__ unsigned s_nf;                    // non-finite flags met by the block (bbox_body's bits)
    if (threadIdx.x == 0) s_nf = 0u;
    static_assert(6 * kBkThreads * sizeof(T) <= (size_t)BLOCK_PTS * sizeof(Pt4<T>), "the six columns of the block's exact partial fit the stage (the 13 of the layout may lie over the tables: onepass3_lds_bytes)");
    const bool second = (int)blockIdx.x >= nb0;
    const Build2Side<T>& a = p.a[second ? 1 : 0];
    const int bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    const int n = a.n, shift = a.shift;
    const unsigned cap = a.cap;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int base = bid * BLOCK_PTS;
    long long* const prof = a.prof;
    long long t_prev = prof ? wall_clock64() : 0;
#define P3_PROF(slot) do { if (prof && tid == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&prof[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
    struct __attribute__((packed, aligned(4))) P3 { T v[3]; };      // a point = ONE 12 / 24-byte load (three scalar loads cost the L1 path three requests)
    const P3* const pts3 = reinterpret_cast<const P3*>(a.pts);
    // ---- grid layout: the context's previous one (GridGeo), or from the sample (every block the same). The sample is requested BEFORE the
    // block's own points: loads return in order, so the layout waits for the sample only and runs while the points are still in flight.
    const bool cached = a.geo_in != nullptr;
    SampleOf<T> smp;
    if (!cached) smp = sample_request<T>(a, n, tid);
    T px[PTS], py[PTS], pz[PTS];
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const P3 p = pts3[min(base + j * kBkThreads + tid, n - 1)];
        px[j] = p.v[0]; py[j] = p.v[1]; pz[j] = p.v[2];
    }
    if (cached) {
        if (tid == 0) {                     // (a wave-uniform address: scalar loads, which do not queue behind the vector loads just issued)
            const GridGeo<T> c = *a.geo_in;
            for (int j = 0; j < 3; ++j) { s_gp.gmin[j] = c.rlo[j]; s_gp.gmax[j] = c.rhi[j]; s_gp.org[j] = c.org[j]; s_gp.slack[j] = c.slack[j]; s_gp.G[j] = c.G[j]; }
            s_gp.h = c.h; s_gp.inv_h = c.inv_h; s_gp.ncells = c.ncells; s_gp.closed = 0;
        }
        __syncthreads();
    } else {
        T rng6[6];
        layout_from_sample<T>(a, n, smp, s_col, s_fin, &s_gp, rng6);
    }
    if (tid == 0 && bid == 0) {             // the side's first block publishes the geometry for the kernels that follow
        GridParams<T>& o = *a.gp;
        for (int j = 0; j < 3; ++j) { o.gmin[j] = s_gp.gmin[j]; o.gmax[j] = s_gp.gmax[j]; o.slack[j] = s_gp.slack[j]; o.G[j] = s_gp.G[j]; o.org[j] = s_gp.org[j]; }
        o.h = s_gp.h; o.inv_h = s_gp.inv_h; o.ncells = s_gp.ncells; o.closed = 0; o.nonfinite = 0; o.sumsq = 0ull;
    }
    const CellMap<T> cm(s_gp);
    const int NB = (uniform(s_gp.ncells) + (1 << shift) - 1) >> shift;           // (<= kStagedMaxBuckets: the host chose this kernel)
    P3_PROF(0);                                                      // sample loads, statistics, layout
    for (int i = tid; i < NB; i += kBkThreads) s_cnt[i] = 0;
    __syncthreads();
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P3_PROF(1); }       // the block's points have arrived
    unsigned bk[PTS], rk[PTS];
    {
        T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
        T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
        unsigned nf = 0;
#pragma unroll
        for (int j = 0; j < PTS; ++j) {
            const bool valid = base + j * kBkThreads + tid < n;
            bk[j] = cm.cell(px[j], py[j], pz[j]) >> shift;
            rk[j] = count_rank(s_cnt, bk[j], valid);
            const T v[3] = {px[j], py[j], pz[j]};
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool fin = (v[c] < (T)0 ? -v[c] : v[c]) <= Limits<T>::max_v;
                lo[c] = (valid && fin && v[c] < lo[c]) ? v[c] : lo[c];
                hi[c] = (valid && fin && v[c] > hi[c]) ? v[c] : hi[c];
                if (valid && !fin) nf |= v[c] != v[c] ? 1u : (v[c] > (T)0 ? (2u << c) : (16u << c));
            }
        }
        // the block's exact partial: 6 columns folded by a wave each; the flags by a ballot per bit
#pragma unroll
        for (int c = 0; c < 3; ++c) { s_col[c * kBkThreads + tid] = lo[c]; s_col[(3 + c) * kBkThreads + tid] = hi[c]; }
        const unsigned wf = wave_or63(nf);
        if (wf && lane == 63) atomicOr(&s_nf, wf);
    }
    __syncthreads();
    P3_PROF(2);                                                      // keys, LDS ranks, statistics
    if (wave < 6) {
        const T r = fold_column<T, kBkThreads>(s_col + wave * kBkThreads, wave < 3 ? 0 : 1, lane);
        if (lane == 63) a.xpartial[(size_t)bid * kXPartStride + wave] = r;
    } else if (tid == 6 * 64) a.xpartial[(size_t)bid * kXPartStride + 6] = (T)s_nf;
    // per bucket: where the block's run starts in the stage (exclusive scan of the counts, two buckets per thread) and in the bucket's slot
    {
        static_assert(kStagedMaxBuckets == 2 * kBkThreads, "two buckets per thread");
        const int i0 = 2 * tid;
        const unsigned c0 = i0 < NB ? s_cnt[i0] : 0u, c1 = i0 + 1 < NB ? s_cnt[i0 + 1] : 0u;
        unsigned total;
        const unsigned ex = block_exclusive_scan_nt<kBkThreads>(c0 + c1, &total);
        if (i0 < NB) s_lbase[i0] = ex;
        if (i0 + 1 < NB) s_lbase[i0 + 1] = ex + c0;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = i0 + q; const unsigned c = q ? c1 : c0;
            if (i < NB && c) {
                const unsigned pad = (c + 7u) & ~7u;
                const unsigned at = (unsigned)atomicAdd(&a.fill[i], ((unsigned long long)c << 32) | (unsigned long long)pad);
                if (at + pad > cap) { *a.ovf = 1ull; s_gbase[i] = 0xffffffffu; }
                else s_gbase[i] = (unsigned)i * cap + at;
            }
        }
    }
    __syncthreads();                // (the folds above have read the columns: the stage may be written)
    P3_PROF(3);                                                      // partial folds, bucket scan, slot reservations (global atomics)
    typedef typename RawRec<T>::type Raw;
#pragma unroll
    for (int j = 0; j < PTS; ++j) {
        const int i = base + j * kBkThreads + tid;
        if (i < n) { Pt4<T> p; p.x = px[j]; p.y = py[j]; p.z = pz[j]; p.idx = i; s_stage[s_lbase[bk[j]] + rk[j]] = p; }
    }
    __syncthreads();
    P3_PROF(4);                                                      // staging
    // run by run, a wave each: lane l of a wave holds the description of bucket wave + 16 (64 r + l) in round r and hands it out by shuffles
    Pt4<T> hole; hole.x = hole.y = hole.z = (T)0; hole.idx = -1;
    for (int r0 = 0; r0 * 1024 + wave < NB; ++r0) {
        const int mine = wave + 16 * (64 * r0 + lane);
        const unsigned mc = mine < NB ? s_cnt[mine] : 0u, ml = mine < NB ? s_lbase[mine] : 0u, mg = mine < NB ? s_gbase[mine] : 0u;
        const int nk = min(64, (NB - wave - 1024 * r0 + 15) / 16);
        for (int k = 0; k < nk; ++k) {
            const unsigned c = (unsigned)__shfl((int)mc, k, 64);
            if (c == 0) continue;
            const unsigned lb = (unsigned)__shfl((int)ml, k, 64), gb = (unsigned)__shfl((int)mg, k, 64);
            if (gb == 0xffffffffu) continue;
            const unsigned pad = (c + 7u) & ~7u;
            for (unsigned l = (unsigned)lane; l < pad; l += 64u) {
                Pt4<T> rec = hole;
                if (l < c) rec = s_stage[lb + l];
                a.tmp[gb + l] = rec;
            }
        }
    }
    P3_PROF(5);                                                      // run copies issued
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); P3_PROF(6); if (tid == 0) atomicAdd((unsigned long long*)&prof[15], 1ull); }
#undef P3_PROF
}

// The launch's extra block of a cloud: the exact bounding box and the input classification from its scatter blocks' partials -> GridParams;
// the slot-overflow flag -> GridParams::has_large; sentinel records; and the housekeeping of the "no memset" scheme (head of this file).
template <typename T>
__device__ __forceinline__ void fold_xpartials(const Build2Side<T>& a) {
    __shared__ T s_lo[kSortThreads / 64][3], s_hi[kSortThreads / 64][3];
    __shared__ unsigned s_nf[kSortThreads / 64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // ---- the layout THIS call's sample gives, for the context's next call (GridGeo); and, if this call was laid out by its predecessor's, is
    // that layout still close to it?
    __shared__ int s_stale;
    if (tid == 0) s_stale = 0;
    if (a.geo_out) {
        extern __shared__ __attribute__((aligned(32))) unsigned char s_dyn[];          // (the bucket blocks' stage: this block has no bucket)
        T* const s_col = reinterpret_cast<T*>(s_dyn);
        __shared__ T s_fin[13];
        __shared__ GridParams<T> s_fresh;
        T rng6[6];
        const SampleOf<T> smp = sample_request<T>(a, a.n, tid);
        layout_from_sample<T>(a, a.n, smp, s_col, s_fin, &s_fresh, rng6);
        if (tid == 0) {
            GridGeo<T> f;
            for (int j = 0; j < 3; ++j) { f.org[j] = s_fresh.org[j]; f.slack[j] = s_fresh.slack[j]; f.G[j] = s_fresh.G[j]; f.rlo[j] = rng6[j]; f.rhi[j] = rng6[3 + j]; }
            f.h = s_fresh.h; f.inv_h = s_fresh.inv_h; f.ncells = s_fresh.ncells;
            if (a.geo_in) {
                const GridGeo<T> u = *a.geo_in;
                bool stale = false;
                for (int j = 0; j < 3; ++j) {
                    const T eu = u.rhi[j] - u.rlo[j], ef = f.rhi[j] - f.rlo[j], ext = eu > ef ? eu : ef, tol = (T)0.05 * ext;
                    const T dl = f.rlo[j] - u.rlo[j], dh = f.rhi[j] - u.rhi[j];
                    if (!((dl < (T)0 ? -dl : dl) <= tol) || !((dh < (T)0 ? -dh : dh) <= tol)) stale = true;
                }
                const T hr = f.h / u.h;
                if (!(hr > (T)0.96 && hr < (T)1.04)) stale = true;
                s_stale = stale ? 1 : 0;
            }
            *a.geo_out = f;
        }
        __syncthreads();
    }
    for (int i = tid; i < a.n_zero_next; i += kSortThreads) a.zero_next[i] = 0ull;
    for (int i = tid; i < a.n_zero2; i += kSortThreads) a.zero2[i] = 0u;
    T lo[3] = {Limits<T>::max_v, Limits<T>::max_v, Limits<T>::max_v};
    T hi[3] = {-Limits<T>::max_v, -Limits<T>::max_v, -Limits<T>::max_v};
    unsigned nf = 0;
    for (int b = tid; b < a.n_xpart; b += kSortThreads) {
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const T p = a.xpartial[(size_t)b * kXPartStride + j], q = a.xpartial[(size_t)b * kXPartStride + 3 + j];
            lo[j] = p < lo[j] ? p : lo[j]; hi[j] = q > hi[j] ? q : hi[j];
        }
        nf |= (unsigned)a.xpartial[(size_t)b * kXPartStride + 6];
    }
#pragma unroll
    for (int j = 0; j < 3; ++j) {
        const T p = wave_min63(lo[j]), q = wave_max63(hi[j]);
        if (lane == 63) { s_lo[wave][j] = p; s_hi[wave][j] = q; }
    }
    { const unsigned m = wave_or63(nf); if (lane == 63) s_nf[wave] = m; }
    __syncthreads();
    if (tid != 0) return;
    GridParams<T>* gp = a.gp;
    unsigned m = 0;
    for (int w = 0; w < kSortThreads / 64; ++w) m |= s_nf[w];
    const unsigned pinf = (m >> 1) & 7u, ninf = (m >> 4) & 7u;
    gp->nonfinite = ((m & 1u) ? kNfNaN : 0) | ((pinf & ninf) ? kNfBothInf : 0) | ((pinf | ninf) ? kNfAnyInf : 0) | (int)(m << 8);      // (as make_grid_body)
    for (int j = 0; j < 3; ++j) {
        T p = s_lo[0][j], q = s_hi[0][j];
        for (int w = 1; w < kSortThreads / 64; ++w) { p = s_lo[w][j] < p ? s_lo[w][j] : p; q = s_hi[w][j] > q ? s_hi[w][j] : q; }
        if (!(p <= q)) { p = 0; q = 0; }          // no finite value in this column
        gp->gmin[j] = p; gp->gmax[j] = q;
    }
    gp->has_large = (*a.ovf ? 2 : 0) | (s_stale ? kGeoStale : 0);
    if (a.n_large) *a.n_large = 0u;
    put_sentinels(a.sorted, a.n);
}

constexpr int kRegIters = (kStageRecs + kSortThreads - 1) / kSortThreads;       // 5 at 4352 / 1024: trips over a slot whose records a thread keeps in registers
// FAST (the normal case): the slot's records (holes included) make at most kRegIters trips and its points fit the LDS stage -- the records stay
// in registers from their single read to their placement in the stage. !FAST (an over-full bucket): the slot is read twice and the records are
// placed directly. (The kernel is bound by instruction issue, not by latency: with the register count forced down to 64 -- two resident
// 1024-thread blocks per CU -- every block ran twice as long, and the spills made it 42 us instead of 26; profiles/r04_build_ab.txt.)
template <typename T, bool FAST>
__device__ __forceinline__ void sort2_body(const int b, const Build2Side<T>& a, const int cnt_cap, const unsigned s, const unsigned pf, const unsigned nvalid,
                                           unsigned* const s_cnt, Pt4<T>* const s_stage, unsigned* const s_w, unsigned long long* const s_q, long long t_prev) {
    GridParams<T>* const gp = a.gp;
    const CellMap<T> cm(*gp);
    const int ncells = uniform(gp->ncells);
    const int shift = a.shift, CB = 1 << shift;
    const int NB = (ncells + CB - 1) >> shift;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long* const prof = a.prof;
#define S2_PROF(slot) do { if (prof && tid == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&prof[slot], (unsigned long long)(t_now - t_prev)); t_prev = t_now; } } while (0)
    const unsigned cap = a.cap;
    const unsigned c0 = (unsigned)b << shift;
    const int ncl = min(CB, ncells - (int)c0);
    const unsigned e = s + nvalid;
    const Pt4<T>* const slot = a.tmp + (size_t)b * cap;
    const unsigned lastp = pf ? pf - 1u : 0u;
    typedef typename RawRec<T>::type Raw;
    constexpr int kIters = FAST ? kRegIters : kSortIters;
    constexpr int kSortBatch = 4;
    Raw recs[FAST ? kRegIters : 1];
    unsigned rr[kIters];               // (cell in bucket) << 16 | rank in cell; 0xffffffff: hole / past the end
    if (FAST) {
        const Raw* const rslot = reinterpret_cast<const Raw*>(slot);
#pragma unroll
        for (int u = 0; u < kRegIters; ++u) recs[u] = rslot[min((unsigned)(u * kSortThreads + tid), lastp)];
#pragma unroll
        for (int u = 0; u < kRegIters; ++u) {
            const unsigned p = (unsigned)(u * kSortThreads + tid);
            const Pt4<T>& rec = reinterpret_cast<const Pt4<T>&>(recs[u]);
            rr[u] = 0xffffffffu;
            if (p < pf && rec.idx >= 0) {
                const unsigned c = cm.cell(rec.x, rec.y, rec.z) - c0;
                rr[u] = (c << 16) | atomicAdd(&s_cnt[c], 1u);
            }
        }
    } else {
#pragma unroll
        for (int it0 = 0; it0 < kIters; it0 += kSortBatch) {
            Pt4<T> rec[kSortBatch];
            const bool batch_on = (unsigned)(it0 * kSortThreads) < pf;       // uniform in the block
            if (batch_on) {
#pragma unroll
                for (int u = 0; u < kSortBatch; ++u) rec[u] = slot[min((unsigned)((it0 + u) * kSortThreads + tid), lastp)];
            }
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) {
                const unsigned p = (unsigned)((it0 + u) * kSortThreads + tid);
                rr[it0 + u] = 0xffffffffu;
                if (batch_on && p < pf && rec[u].idx >= 0) {
                    const unsigned c = cm.cell(rec[u].x, rec[u].y, rec[u].z) - c0;
                    rr[it0 + u] = (c << 16) | atomicAdd(&s_cnt[c], 1u);
                }
            }
        }
    }
    __syncthreads();
    S2_PROF(1);                                    // records in, cells, LDS ranks
    // exclusive scan of the CB counters (thread-contiguous items), cell_start, balance metric
    const int per = CB >= kSortThreads ? CB / kSortThreads : 1;
    const int i0 = tid * per;
    {
        unsigned v[kBkMaxCellsPerBucket / kSortThreads], sum = 0; unsigned long long sq = 0;
#pragma unroll
        for (int q = 0; q < kBkMaxCellsPerBucket / kSortThreads; ++q) {
            v[q] = (q < per && i0 + q < CB) ? s_cnt[i0 + q] : 0u;
            sum += v[q]; sq += (unsigned long long)v[q] * v[q];
        }
        unsigned inc = sum;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o, 64); if (lane >= o) inc += t; }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) sq += __shfl_xor(sq, o, 64);
        if (lane == 63) s_w[wave] = inc;
        if (lane == 0) s_q[wave] = sq;
        __syncthreads();
        if (tid == 0) {
            unsigned run = 0; unsigned long long Q = 0;
            for (int w = 0; w < kSortThreads / 64; ++w) { const unsigned t = s_w[w]; s_w[w] = run; run += t; Q += s_q[w]; }
            if (Q) atomicAdd(&gp->sumsq, Q);
            if (b == NB - 1) a.cell_start[ncells] = e;
        }
        __syncthreads();
        unsigned ex = inc - sum + s_w[wave];
#pragma unroll
        for (int q = 0; q < kBkMaxCellsPerBucket / kSortThreads; ++q) {
            if (q < per && i0 + q < CB) {
                s_cnt[i0 + q] = ex;
                if (i0 + q < ncl) a.cell_start[c0 + i0 + q] = s + ex;
                ex += v[q];
            }
        }
    }
    __syncthreads();
    S2_PROF(2);                                    // scan, cell_start
    const int n_pts = a.n;
    T* const xyz = xyz_of(a.sorted, n_pts);
    int* const idx32 = idx32_of(a.sorted, n_pts);
    if (!FAST) {
#pragma unroll
        for (int it0 = 0; it0 < kIters; it0 += kSortBatch) {
            if (!((unsigned)(it0 * kSortThreads) < pf)) break;                  // uniform in the block
            Pt4<T> rec[kSortBatch];
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) rec[u] = slot[min((unsigned)((it0 + u) * kSortThreads + tid), lastp)];
#pragma unroll
            for (int u = 0; u < kSortBatch; ++u) {
                const unsigned r = rr[it0 + u];
                if (r != 0xffffffffu) {
                    const unsigned pos = s + s_cnt[r >> 16] + (r & 0xffffu);
                    if (a.want_pt4) a.sorted[pos] = rec[u];
                    put_xyz(a.sorted, n_pts, pos, rec[u]);
                    if (a.pos_of) a.pos_of[rec[u].idx] = pos;
                }
            }
        }
        return;
    }
    // The stage holds the two output streams as they will lie in memory: the coordinates packed (3 scalars per record, no padding), the row ids
    // behind them. The packed stream starts `phase` scalars into the stage so that 16-byte chunks of the stage are 16-byte chunks of the output:
    // the copy-out is one ds_read_b128 + one 16-byte store per 4 (2) scalars instead of index arithmetic per scalar.
    constexpr unsigned V = 16 / sizeof(T);                           // scalars per 16 bytes
    T* const st = reinterpret_cast<T*>(s_stage);
    int* const sidx = reinterpret_cast<int*>(st + (V + 3 * kStageRecs + V - 1) / V * V);
    static_assert(((V + 3 * kStageRecs + V - 1) / V * V) * sizeof(T) + (size_t)kStageRecs * 4 <= (size_t)kStageRecs * sizeof(Pt4<T>) + 64, "both streams fit the stage (bucket_sort_lds_bytes)");
    const unsigned phase = (unsigned)((3ull * (unsigned long long)s) & (V - 1));
    {
#pragma unroll
        for (int u = 0; u < kRegIters; ++u) {
            const unsigned r = rr[u];
            if (r != 0xffffffffu) {
                const Pt4<T>& rec = reinterpret_cast<const Pt4<T>&>(recs[u]);
                const unsigned pos = s_cnt[r >> 16] + (r & 0xffffu);
                T* const o = st + phase + 3u * pos;
                o[0] = rec.x; o[1] = rec.y; o[2] = rec.z; sidx[pos] = (int)rec.idx;
            }
        }
    }
    __syncthreads();
    S2_PROF(3);                                    // placement in the stage
    for (unsigned i = tid; i < nvalid; i += kSortThreads) {
        const int id = sidx[i];
        idx32[s + i] = id;
        if (a.want_pt4) { Pt4<T> r; const T* const o = st + phase + 3u * i; r.x = o[0]; r.y = o[1]; r.z = o[2]; r.idx = id; a.sorted[s + i] = r; }
        if (a.pos_of) a.pos_of[id] = s + i;
    }
    {
        // stage scalar L <-> output scalar 3 s - phase + L; the packed stream is stage[phase, end)
        const unsigned end = phase + 3u * nvalid;
        T* const out = xyz + (3 * (size_t)s - phase);                // (16-byte aligned: 3 s - phase is a multiple of V)
        typedef typename RawRec<float>::type Raw16;                  // 16 bytes
        const unsigned v0 = (phase + V - 1) / V, v1 = end / V;       // whole chunks [v0, v1)
        for (unsigned c = v0 + tid; c < v1; c += kSortThreads)
            reinterpret_cast<Raw16*>(out)[c] = reinterpret_cast<const Raw16*>(st)[c];
        if (tid < V) {                                               // the partial chunks at both ends, scalar by scalar
            const unsigned La = phase + tid;                         // head: [phase, min(v0 V, end))
            if (La < min(v0 * V, end)) out[La] = st[La];
            const unsigned Lb = max(v1 * V, v0 * V) + tid;           // tail: [max(v1, v0) V, end)
            if (v1 >= v0 && Lb < end) out[Lb] = st[Lb];
        }
    }
    S2_PROF(4);                                    // copies issued
    if (prof) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); S2_PROF(5); if (tid == 0) atomicAdd((unsigned long long*)&prof[7], 1ull); }
#undef S2_PROF
}

// One bucket: its output offset and fill, then the path that fits it.
template <typename T>
__device__ __forceinline__ void sort2_bucket(const int b, const Build2Side<T>& a, const int cnt_cap) {
    // dynamic LDS: [cnt_cap counters][kStageRecs records]
    extern __shared__ __attribute__((aligned(32))) unsigned char s_dyn[];
    unsigned* const s_cnt = reinterpret_cast<unsigned*>(s_dyn);
    Pt4<T>* const s_stage = reinterpret_cast<Pt4<T>*>(s_dyn + (size_t)cnt_cap * 4);
    __shared__ unsigned s_w[kSortThreads / 64 + 1];
    __shared__ unsigned long long s_q[kSortThreads / 64];
    if (*a.ovf) return;                           // a slot overflowed: the index is not usable (the host rebuilds with the two-pass pipeline)
    const int CB = 1 << a.shift;
    const int NB = (uniform(a.gp->ncells) + CB - 1) >> a.shift;
    if (b >= NB) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    long long t_prev = a.prof ? wall_clock64() : 0;
    // output offset = the points of the buckets before this one
    unsigned s;
    {
        unsigned part = 0;
        for (int i = tid; i < b; i += kSortThreads) part += (unsigned)(a.fill[i] >> 32);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) part += __shfl_xor(part, o, 64);
        if (lane == 0) s_w[wave] = part;
        for (int i = tid; i < CB; i += kSortThreads) s_cnt[i] = 0u;
        __syncthreads();
        s = 0;
        for (int w = 0; w < kSortThreads / 64; ++w) s += s_w[w];
        __syncthreads();
    }
    if (a.prof && tid == 0) { const long long t_now = wall_clock64(); atomicAdd((unsigned long long*)&a.prof[0], (unsigned long long)(t_now - t_prev)); t_prev = t_now; }      // head
    const unsigned long long fw = a.fill[b];
    const unsigned pf = min((unsigned)fw, a.cap), nvalid = (unsigned)(fw >> 32);
    static_assert(kSortThreads * kSortIters >= kLargeBucket, "one trip per record of a full slot");
    if (pf <= (unsigned)(kRegIters * kSortThreads) && nvalid <= (unsigned)kStageRecs) sort2_body<T, true>(b, a, cnt_cap, s, pf, nvalid, s_cnt, s_stage, s_w, s_q, t_prev);      // (uniform in the block)
    else sort2_body<T, false>(b, a, cnt_cap, s, pf, nvalid, s_cnt, s_stage, s_w, s_q, t_prev);
}

// (the side's arguments through an index into the kernel-argument segment, see k_bucket_onepass3)
template <typename T>
__global__ __launch_bounds__(kSortThreads) void k_bucket_sort2(const Build2Args<T> p, int nb0, int nb1, int cnt_cap, int nfold) {
    // The launch's FIRST blocks, one per cloud, are the extra blocks (fold_xpartials): since round 6 they also compute the layout the context's next
    // call will use (a cold sample read + the serial layout: longer than a bucket), and as the launch's last blocks they ended it 5 us late.
    const int bid = (int)blockIdx.x;
    if (bid < nfold) { fold_xpartials<T>(p.a[bid]); return; }
    const int b = bid - nfold;
    const int side = b >= nb0 ? 1 : 0;
    sort2_bucket<T>(side ? b - nb0 : b, p.a[side], cnt_cap);
}

// The Pt4 records of a LEAN index from its two streams (for the kernels that read whole records: k > 1 lane passes, row-based epilogues,
// the tie-order resolver), sentinels included.
template <typename T>
struct Pt4Side { Pt4<T>* sorted; int n; };
template <typename T>
__global__ __launch_bounds__(kBlock) void k_make_pt4(const Pt4Side<T> a0, const Pt4Side<T> a1, int nb0) {
    const bool second = (int)blockIdx.x >= nb0;
    const Pt4Side<T>& a = second ? a1 : a0;
    const int nblk = second ? (int)gridDim.x - nb0 : nb0, bid = second ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    const T* const xyz = xyz_of(a.sorted, a.n);
    const int* const idx32 = idx32_of(a.sorted, a.n);
    for (int i = bid * kBlock + (int)threadIdx.x; i < a.n + 8; i += nblk * kBlock) {
        Pt4<T> r; r.x = xyz[3 * (size_t)i]; r.y = xyz[3 * (size_t)i + 1]; r.z = xyz[3 * (size_t)i + 2]; r.idx = idx32[i];
        a.sorted[i] = r;
    }
}

}  // namespace pcu
