// csrc/search_brick.h -- k = 1 main pass over a SHARED grid (round 6): the dataset box of a block of queries staged in LDS.
//
// Replaces, for two-sided fused calls between clouds of comparable size (Chamfer p = 2 without indices: the headline), the per-lane gathers of
// k_search1_flat -- i.e. nanoflann's searchLevel / leaf scan, external/nanoflann/nanoflann.hpp:1544-1624, as every search kernel here does.
// k_search1_flat is bound by the CU's texture-address path: 38.8 vector-memory instructions per wave at ~20 cycles each, TA busy 74 %
// (profiles/r05_pmc.txt) -- one query, nine row tables and ~25 candidate groups per lane, all per-lane gathers. Round 4's block-level LDS tile
// lost (profiles/r04_flat_tile_ab.txt) for three reasons this design removes:
//   * the two clouds' grids were not aligned (a strip of query cells touched 4 x 4 dataset rows per query row; 21 staged rows per block).
//     With ONE grid over both clouds (grid2.h: Build2Side::spts1) a query's cell in its own cloud's order IS its cell in the dataset's grid:
//     a block's NT consecutive queries are ~1.6 consecutive ROWS of cells, whose candidates are the whole rows (y - 1 .. y + 1) x (z - 1 .. z + 1)
//     around them: 12-15 rows, each ONE contiguous run of the dataset's coordinate stream (~158 records, 1.9 KB);
//   * the stage was filled through registers as 16-byte records by dependent per-row trips. Here a row is two global -> LDS DMA instructions
//     (global_load_lds_dwordx4: 1 KB per wave-instruction, no VGPRs, no ds_write), the 12-byte records stay packed, and the rows' slices of
//     cell_start come the same way: ~9 vector-memory instructions per wave in all, issued back to back behind ONE dependent step (the rows'
//     start / end words);
//   * the scan was a different, longer program. Here every lane runs k_search1_flat's own scan -- centre row, rounding-safe cell and row cuts,
//     the surviving runs in a per-lane list, one pipelined loop, certification against the 27-cell box, radius 2 by the query's own wave --
//     on LDS addresses: same candidates, same bounds, same minimum.
// The minimum d2 of a certified query does not depend on where its candidates were read from, so the fused sum is the same set of numbers;
// only their grouping into per-block fp64 partials follows this kernel's block size.
// A block whose box does not fit the stage (a block that straddles two z slabs of the grid, rows of very uneven length) scans its 27 cells
// straight from global memory, lane by lane (brick_scan_global: the plain form of the same scan) and counts itself in SearchArgs::n_fallback;
// a context whose calls mostly fall back (surfaces, clusters: rows are short and uneven) goes back to k_search1_flat (pcu_hip.hip: brick_ok).
#pragma once
#include "search.h"

namespace pcu {

#ifndef PCU_BRICK_NT
#define PCU_BRICK_NT 256
#endif
#ifndef PCU_BRICK_MINW
#define PCU_BRICK_MINW 4                    // waves per SIMD the register budget allows (128 VGPRs: nothing spills)
#endif
#ifndef PCU_BRICK_LDS
#define PCU_BRICK_LDS 39168                 // bytes of the stage: with the kernel's other shared arrays 4 blocks of 256 threads fill a CU's 160 KB
#endif
constexpr int kBrickNT = PCU_BRICK_NT;
constexpr int kBrickLds = PCU_BRICK_LDS;
constexpr int kBrickRows = 64;              // staged rows per block at most (one lane of wave 0 lays out each)
constexpr int kBrickRowsBytes = kBrickRows * 8, kBrickSent = kBrickRowsBytes, kBrickRegA = kBrickRowsBytes + 64;      // layout of the stage, see below
constexpr int kBrickListEntries = 8;

struct BrickRow { unsigned radj, tabw; };   // LDS byte address of record i of the row = 12 i + radj; tabw: LDS word index of the row's cell_start slice, position 0

// 16 bytes per lane from global memory straight into LDS: lane l's chunk lands at lds + 16 l (lds: wave-uniform).
__device__ __forceinline__ void glds16(const void* gsrc, void* lds_wave_base) {
    __builtin_amdgcn_global_load_lds(reinterpret_cast<const __attribute__((address_space(1))) void*>(reinterpret_cast<uintptr_t>(gsrc)),
                                     reinterpret_cast<__attribute__((address_space(3))) void*>((unsigned)reinterpret_cast<uintptr_t>(lds_wave_base)), 16, 0, 0);
}

// The plain form of the k = 1 scan for one lane, candidates from global memory: the nine rows of the query's 27 cells, centre row first, a row
// skipped when its slab is beyond the running minimum (the same rounding-safe bound as everywhere), groups of 4 records. Used by blocks that do
// not fit the stage; its minimum over the box is the staged scan's.
template <typename T>
__device__ __forceinline__ void brick_scan_global(const SearchArgs<T>& a, const GridParams<T>& g, const Pt4<T>& q, int ccx, int ccy, int ccz,
                                                  const T (&my2)[3], const T (&mz2)[3], unsigned cand_cap, T& best, bool& defer) {
    typedef GroupEval<T, true> GE;
    constexpr unsigned kRec = GE::kRec;
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const char* const base = reinterpret_cast<const char*>(a.ref_xyz);
    const int xa = max(ccx - 1, 0), xb = min(ccx + 1, Gx - 1);
    unsigned total = 0;
    for (int j = 0; j < 9 && !defer; ++j) {
        const int oy = kRowOy[j], oz = kRowOz[j];
        const int y = ccy + oy, z = ccz + oz;
        if (y < 0 || y >= Gy || z < 0 || z >= Gz) continue;
        const T ry = my2[oy == 0 ? 0 : (oy < 0 ? 1 : 2)], rz = mz2[oz == 0 ? 0 : (oz < 0 ? 1 : 2)];
        const T rlb = oy == 0 ? rz : (oz == 0 ? ry : ry + rz);
        if (best < rlb) continue;
        const unsigned lo = (unsigned)row_run_lo(Gx, grid_row(Gy, y, z), xa, xb);
        const unsigned s = a.cell_start[lo], e = a.cell_start[lo + (unsigned)(xb - xa + 1)];
        total += e - s;
        if (total > cand_cap) { defer = true; break; }
        for (unsigned off = s * kRec; off < e * kRec; off += 4u * kRec) {
            T d[4]; GE::dists(GE::load(base, off), q, d);
            const T m = min4(d[0], d[1], d[2], d[3]);
            best = m < best ? m : best;
        }
    }
}

template <typename T, int NT>
__device__ __forceinline__ void search1_brick_body(const SearchArgs<T>& a, const int nq, const int bid, const int nblk, bool& f_ok, T& f_v) {
    static_assert(sizeof(T) == 4, "the stage is sized for 12-byte records");
    typedef GroupEval<T, true> GE;
    constexpr unsigned kRec = GE::kRec;
    constexpr int kG = 4, NW = NT / 64;
    constexpr unsigned kStep = (unsigned)kG * kRec;
    // the stage: [0, 512) BrickRow[64] | [512, 576) one group of +inf records | region A: the rows' cell_start slices, LATER the lanes' run lists |
    // the rows' records, packed as they lie in the dataset's coordinate stream
    __shared__ __attribute__((aligned(16))) unsigned char s_pool[kBrickLds];
    __shared__ int s_geo[4];
    __shared__ int s_hdr[8];                    // ok, y0, z0, ny, nz, nrows, stride (words), chunks of the tables
    __shared__ unsigned s_g0[kBrickRows], s_nch[kBrickRows], s_dst[kBrickRows], s_w0[kBrickRows];
    const int per = nblk >> 3;
    const int vb = (bid & 7) * per + (bid >> 3);       // XCD-aware block order (nblk: a multiple of 8), as k_search1_flat
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (vb * NT >= nq) return;                         // (block-uniform)
    const GridParams<T>& g = *a.gp;
    if (const int hl = index_not_ready(a, g)) { if (vb == 0 && tid == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (vb == 0 && tid == 0) *a.skew_flag = (float)g.sumsq > a.skew_far ? 2 : 1; return; }
    const int t = vb * NT + tid;
    const bool valid = t < nq;
    const int qpos = valid ? t : nq - 1;               // (lanes past the end shadow the last query: every lane takes part in the barriers)
    Pt4<T> q;
    {
        struct __attribute__((packed, aligned(4))) Q3 { T v[3]; };
        const Q3 c = *reinterpret_cast<const Q3*>(a.q_xyz + 3 * (size_t)qpos);
        q.x = c.v[0]; q.y = c.v[1]; q.z = c.v[2]; q.idx = 0;
    }
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = cell_of_query(g, 0, q.x), ccy = cell_of_query(g, 1, q.y), ccz = cell_of_query(g, 2, q.z);
    const unsigned cand_cap = a.lane_max_cand < 65535u ? a.lane_max_cand : 65535u;
    // ---- the block's box of dataset rows, laid out by wave 0
    const int last = min(NT, nq - vb * NT) - 1;
    if (tid == 0) { s_geo[0] = ccy; s_geo[1] = ccz; }
    if (tid == last) { s_geo[2] = ccy; s_geo[3] = ccz; }
    if (tid < 12) reinterpret_cast<T*>(s_pool + kBrickSent)[tid] = (T)INFINITY;
    __syncthreads();
    if (wave == 0) {
        const int yA = s_geo[0], zA = s_geo[1], yB = s_geo[2], zB = s_geo[3];
        // queries are in cell order: the block covers the linear rows from (yA, zA) to (yB, zB); within a slab y runs one way, and a block that
        // crosses into the next slab turns around at the slab's end
        int ylo = min(yA, yB), yhi = max(yA, yB);
        if (zA != zB) { const int e = (zA & 1) ? 0 : Gy - 1; ylo = min(ylo, e); yhi = max(yhi, e); }
        const int y0 = max(ylo - 1, 0), y1 = min(yhi + 1, Gy - 1), z0 = max(zA - 1, 0), z1 = min(zB + 1, Gz - 1);
        const int ny = y1 - y0 + 1, nz = z1 - z0 + 1, nrows = ny * nz;
        bool ok = zB - zA <= 1 && zB >= zA && nrows <= kBrickRows;
        const int stride = ((Gx + 1 + 3) + 3) & ~3;                    // words per row slice: up to 3 words of alignment + Gx + 1 cell starts
        const int tab_bytes = nrows * stride * 4;
        const int reg_a = max(tab_bytes, NT * kBrickListEntries * 4);
        const int cand_base = kBrickRegA + reg_a;
        unsigned g0 = 0, nch = 0, w0 = 0;
        if (ok && lane < nrows) {
            const int y = y0 + lane % ny, z = z0 + lane / ny;
            w0 = (unsigned)grid_row(Gy, y, z) * (unsigned)Gx;
            const unsigned S = a.cell_start[w0], E = a.cell_start[w0 + (unsigned)Gx];
            g0 = (S * kRec) & ~15u;
            nch = (E * kRec + 3u * kRec - g0 + 15u) >> 4;              // (+ 3 records: a group that starts at the row's last record)
        }
        unsigned inc = nch;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned u = (unsigned)__shfl_up((int)inc, o, 64); if (lane >= o) inc += u; }
        const unsigned total = (unsigned)__shfl((int)inc, 63, 64);
        ok = ok && (unsigned)cand_base + 16u * total <= (unsigned)kBrickLds;
        if (lane < kBrickRows) {
            const unsigned dst = (unsigned)cand_base + 16u * (inc - nch);
            s_g0[lane] = g0; s_nch[lane] = nch; s_dst[lane] = dst; s_w0[lane] = w0 & ~3u;
            BrickRow br; br.radj = dst - g0; br.tabw = (unsigned)(kBrickRegA / 4) + (unsigned)(lane * stride) + (w0 & 3u);
            reinterpret_cast<BrickRow*>(s_pool)[lane] = br;
        }
        if (lane == 0) { s_hdr[0] = ok ? 1 : 0; s_hdr[1] = y0; s_hdr[2] = z0; s_hdr[3] = ny; s_hdr[4] = nz; s_hdr[5] = nrows; s_hdr[6] = stride; }
    }
    __syncthreads();
    const bool staged = s_hdr[0] != 0;
    // ---- per-query margins (the scan's bounds; the same arithmetic as k_search1_flat)
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    T mxl = q.x - face_below(g, 0, ccx); mxl = mxl > (T)0 ? mxl * shrink : (T)0;
    T mxh = face_above(g, 0, ccx) - q.x; mxh = mxh > (T)0 ? mxh * shrink : (T)0;
    const T mxl2 = mxl * mxl, mxh2 = mxh * mxh;
    T my2[3], mz2[3];
    {
        T m;
        my2[0] = (T)0; mz2[0] = (T)0;
        m = q.y - face_below(g, 1, ccy); m = m > (T)0 ? m * shrink : (T)0; my2[1] = m * m;
        m = face_above(g, 1, ccy) - q.y; m = m > (T)0 ? m * shrink : (T)0; my2[2] = m * m;
        m = q.z - face_below(g, 2, ccz); m = m > (T)0 ? m * shrink : (T)0; mz2[1] = m * m;
        m = face_above(g, 2, ccz) - q.z; m = m > (T)0 ? m * shrink : (T)0; mz2[2] = m * m;
    }
    T best = Limits<T>::max_v;
    bool defer = false;
    if (!staged) {
        if (tid == 0 && a.n_fallback) atomicAdd(a.n_fallback, 1);
        brick_scan_global(a, g, q, ccx, ccy, ccz, my2, mz2, cand_cap, best, defer);
    } else {
        const int y0 = s_hdr[1], z0 = s_hdr[2], ny = s_hdr[3], nrows = s_hdr[5], stride = s_hdr[6];
        // ---- stage: the rows' records (a wave takes every NW-th row), then the rows' cell_start slices
        {
            const char* const refb = reinterpret_cast<const char*>(a.ref_xyz);
            for (int r = wave; r < nrows; r += NW) {
                const unsigned g0 = (unsigned)__builtin_amdgcn_readfirstlane((int)s_g0[r]), nch = (unsigned)__builtin_amdgcn_readfirstlane((int)s_nch[r]);
                const unsigned dst = (unsigned)__builtin_amdgcn_readfirstlane((int)s_dst[r]);
                for (unsigned c0 = 0; c0 < nch; c0 += 64u)
                    if (c0 + (unsigned)lane < nch) glds16(refb + (size_t)g0 + 16u * (size_t)(c0 + (unsigned)lane), s_pool + dst + 16u * c0);
            }
            const int cpr = stride >> 2, nchunk = nrows * cpr;
            for (int f0 = wave * 64; f0 < nchunk; f0 += NW * 64) {
                const int f = f0 + lane;
                if (f < nchunk) {
                    const int r = f / cpr, i = f - r * cpr;
                    glds16(a.cell_start + s_w0[r] + 4u * (unsigned)i, s_pool + kBrickRegA + 16 * f0);
                }
            }
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __syncthreads();
        // ---- the nine row tables, from the staged slices (k_search1_flat: "the nine row tables"; a row's table = the four words {start of the
        // cell before the query's in the row's direction, start of its own, of the next one, end of that})
        const bool hasxl = ccx > 0, hasxh = ccx < Gx - 1;
        const bool odd0 = ((ccy ^ ccz) & 1) != 0;
        const bool hasA = (odd0 && hasxh) || (!odd0 && hasxl), hasB = (odd0 && hasxl) || (!odd0 && hasxh);
        const int pS = odd0 ? Gx - 1 - ccx : ccx, pR = odd0 ? ccx : Gx - 1 - ccx;     // the query's cell position in rows that run like the centre row / against it
        const bool okyM = ccy > 0, okyP = ccy < Gy - 1, okzM = ccz > 0, okzP = ccz < Gz - 1;
        const int r0 = (ccy - y0) + ny * (ccz - z0);
        const unsigned* const pw = reinterpret_cast<const unsigned*>(s_pool);
        bool okj[9]; unsigned tb[9][4]; unsigned radj[9];
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            const int oy = kRowOy[j], oz = kRowOz[j];
            okj[j] = (oy == 0 || (oy < 0 ? okyM : okyP)) && (oz == 0 || (oz < 0 ? okzM : okzP));
            const int rj = okj[j] ? r0 + oy + ny * oz : r0;
            const BrickRow br = reinterpret_cast<const BrickRow*>(s_pool)[rj];
            radj[j] = br.radj;
            const unsigned w = br.tabw + (unsigned)((((oy + oz) & 1) == 0 ? pS : pR) - 1);
            tb[j][0] = pw[w]; tb[j][1] = pw[w + 1]; tb[j][2] = pw[w + 2]; tb[j][3] = pw[w + 3];
        }
        __syncthreads();                                // (every lane holds its tables: region A becomes the run lists)
        unsigned* const list = reinterpret_cast<unsigned*>(s_pool + kBrickRegA) + tid;       // entry n of this lane: list[n * NT]
#define PCU_BRICK_EVAL(OFF)                                                                              \
        {                                                                                                \
            const typename GE::Raw raw_ = *reinterpret_cast<const typename GE::Raw*>(s_pool + (OFF));    \
            T d_[4]; GE::dists(raw_, q, d_);                                                             \
            const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);                                               \
            best = m_ < best ? m_ : best;                                                                \
        }
        // ---- centre row: whole run
        const unsigned c_s = hasA ? tb[0][0] : tb[0][1], c_e = hasB ? tb[0][3] : tb[0][2];
        const unsigned cnt0 = c_e - c_s;
        defer = cnt0 > cand_cap;
        {
            const unsigned o0 = c_s * kRec + radj[0];
            const unsigned o1 = defer ? o0 : c_e * kRec + radj[0];
            for (unsigned off = o0; off < o1; off += kStep) PCU_BRICK_EVAL(off)
        }
        // ---- the other rows: cut runs that survive the centre row's minimum -> this lane's list (entry: LDS word offset << 18 | records << 12 |
        // the row bound's exponent and top 4 mantissa bits, rounded down)
        const T mA2 = odd0 ? mxh2 : mxl2, mB2 = odd0 ? mxl2 : mxh2;
        unsigned total = cnt0;
        int n = 0;
#pragma unroll
        for (int j = 1; j < 9; ++j) {
            const int oy = kRowOy[j], oz = kRowOz[j];
            const T ry = my2[oy == 0 ? 0 : (oy < 0 ? 1 : 2)], rz = mz2[oz == 0 ? 0 : (oz < 0 ? 1 : 2)];
            const bool same = ((oy + oz) & 1) == 0;
            const T mF2 = same ? mA2 : mB2, mL2 = same ? mB2 : mA2;
            const bool hasF = same ? hasA : hasB, hasL = same ? hasB : hasA;
            const T rlb = oy == 0 ? rz : (oz == 0 ? ry : ry + rz);
            const T bF = oy == 0 ? mF2 + rz : (oz == 0 ? mF2 + ry : (mF2 + ry) + rz), bL = oy == 0 ? mL2 + rz : (oz == 0 ? mL2 + ry : (mL2 + ry) + rz);
            const bool cutF = !hasF || best < bF, cutL = !hasL || best < bL;
            const unsigned s_run = cutF ? tb[j][1] : tb[j][0], e_run = cutL ? tb[j][2] : tb[j][3];
            const unsigned cnt = e_run - s_run;
            const bool take = okj[j] && !defer && !(best < rlb) && e_run > s_run;
            if (take && cnt > 63u) defer = true;                       // (a run the 6-bit count cannot hold: the wave-per-query pass takes the query)
            list[n * NT] = (((s_run * kRec + radj[j]) >> 2) << 18) | ((cnt & 63u) << 12) | ((__float_as_uint((float)rlb) >> 19) & 0xfffu);
            n += take ? 1 : 0;
            total += take ? cnt : 0u;
        }
        if (total > cand_cap || defer) { defer = true; n = 0; }
        int r = 0;
        unsigned off = 0, end = 0;
        bool live = false;
        auto next_run = [&]() {
            live = false;
            while (r < n) {
                const unsigned e = list[r * NT];
                ++r;
                if (!(best < (T)__uint_as_float((e & 0xfffu) << 19))) { off = (e >> 18) << 2; end = off + ((e >> 12) & 63u) * kRec; live = true; break; }
            }
        };
        next_run();
        while (live) {
            const unsigned coff = off;
            off += kStep;
            if (off >= end) next_run();
            PCU_BRICK_EVAL(coff)
        }
#undef PCU_BRICK_EVAL
    }
    // ---- certification against the 27-cell box; radius 2 by the query's own wave for the few stragglers (k_search1_flat: "radius 2, inside the
    // launch", the fused sum's form: only the value)
    T lb;
    {
        const T smax = g.slack[0] > g.slack[1] ? (g.slack[0] > g.slack[2] ? g.slack[0] : g.slack[2]) : (g.slack[1] > g.slack[2] ? g.slack[1] : g.slack[2]);
        const T mag = ((fabs(g.org[0]) + fabs(g.org[1])) + fabs(g.org[2])) + (T)(Gx + Gy + Gz + 3) * g.h + smax;
        const T hq = (g.h - (T)2 * smax) - (T)16 * Limits<T>::eps * mag;
        const T hs = hq > (T)0 ? hq * shrink : (T)0;
        lb = hs * hs;
    }
    if (__ballot(valid && !defer && !(best < lb)) != 0ull) {
        const int cx0 = max(ccx - 1, 0), cx1 = min(ccx + 1, Gx - 1);
        const int cy0 = max(ccy - 1, 0), cy1 = min(ccy + 1, Gy - 1), cz0 = max(ccz - 1, 0), cz1 = min(ccz + 1, Gz - 1);
        lb = face_lower_bound_inner(g, q.x, q.y, q.z, cx0, cx1, cy0, cy1, cz0, cz1);
        unsigned long long todo = __ballot(valid && !defer && !(best < lb));
        if (todo && __popcll(todo) <= 4 && __ballot(true) == ~0ull) {
            const char* const base = reinterpret_cast<const char*>(a.ref_xyz);
            while (todo) {
                const int l = __ffsll((long long)todo) - 1;
                todo &= todo - 1;
                Pt4<T> sq;
                sq.x = __shfl(q.x, l, 64); sq.y = __shfl(q.y, l, 64); sq.z = __shfl(q.z, l, 64); sq.idx = 0;
                const int scx = grid_cell(g, 0, sq.x), scy = grid_cell(g, 1, sq.y), scz = grid_cell(g, 2, sq.z);
                const int bx0 = max(scx - 2, 0), bx1 = min(scx + 2, Gx - 1), by0 = max(scy - 2, 0), by1 = min(scy + 2, Gy - 1);
                const int bz0 = max(scz - 2, 0), bz1 = min(scz + 2, Gz - 1);
                const int ny_ = by1 - by0 + 1, nrows_ = ny_ * (bz1 - bz0 + 1);
                T wbest = Limits<T>::max_v;
                if ((lane >> 1) < nrows_) {
                    const int r_ = lane >> 1;
                    const unsigned lo = (unsigned)row_run_lo(Gx, grid_row(Gy, by0 + r_ % ny_, bz0 + r_ / ny_), bx0, bx1);
                    const unsigned rs_ = a.cell_start[lo], re_ = a.cell_start[lo + (unsigned)(bx1 - bx0 + 1)];
                    const unsigned mid = rs_ + (((re_ - rs_ + 1u) >> 1) + 3u) / 4u * 4u;
                    const unsigned s_ = ((lane & 1) ? min(mid, re_) : rs_) * kRec, e_ = ((lane & 1) ? re_ : min(mid, re_)) * kRec;
                    for (unsigned off_ = s_; off_ < e_; off_ += kStep) {
                        T d_[4]; GE::dists(GE::load(base, off_), sq, d_);
                        if (!(lane & 1)) {
#pragma unroll
                            for (int u = 1; u < 4; ++u) d_[u] = off_ + (unsigned)u * kRec >= e_ ? (T)INFINITY : d_[u];
                        }
                        const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);
                        wbest = m_ < wbest ? m_ : wbest;
                    }
                }
                T mn = wbest;
#pragma unroll
                for (int o = 32; o > 0; o >>= 1) { const T ot = __shfl_xor(mn, o, 64); mn = ot < mn ? ot : mn; }
                const T lb2 = face_lower_bound_inner(g, sq.x, sq.y, sq.z, bx0, bx1, by0, by1, bz0, bz1);
                if (lane == l) { best = mn; lb = lb2; }
            }
        }
    }
    if (defer) {                 // nothing (final) was scanned: the wave-per-query pass at the same radius takes over
        wave_append(valid, qpos, a.ties, a.n_ties);
        return;
    }
    const bool certified = valid && best < lb;
    const int us = wave_append(valid && !certified, qpos, a.unresolved, a.n_unresolved);
    if (us >= 0 && a.ubound) a.ubound[us] = best;
    f_ok = certified;
    f_v = a.squared ? best : sqrt(best);
}

template <typename T, int NT>
__global__ __launch_bounds__(NT, PCU_BRICK_MINW) void k_search1_brick(const SearchArgs2<T> p, int nb0) {
    const int side = (int)blockIdx.x >= nb0 ? 1 : 0;
    const int bid = side ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    bool ok = false; T v = (T)0;
    const int nq_side = side ? p.a[1].nq : p.a[0].nq;
    search1_brick_body<T, NT>(p.a[side], nq_side, bid, side ? (int)gridDim.x - nb0 : nb0, ok, v);
    // one fp64 partial per block; lanes in a fixed order: reproducible
    double s = ok ? (double)v : 0.0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    __shared__ double ss[NT / 64];
    if ((threadIdx.x & 63) == 0) ss[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) { double r = 0; for (int w = 0; w < NT / 64; ++w) r += ss[w]; p.a[side].f_sum[bid] = r; }
}

}  // namespace pcu
