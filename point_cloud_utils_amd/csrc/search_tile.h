// csrc/search_tile.h -- block-cooperative, LDS-staged variant of the k = 1 main pass (k_search1_flat, search.h): the north-star's
// "reference tiles staged through LDS for coalesced HBM reads", re-measured in round 4 against the tuned gather kernel
// (profiles/r04_flat_tile_ab.txt). Same scan, same arithmetic, same certification, same lists and epilogues -- only where a lane's
// candidates come from differs:
//
//   k_search1_flat   every lane gathers its own candidate records from the cell-ordered stream in global memory (12-byte records, three
//                    16-byte loads per group of 4): ~54 per-lane gather instructions per wave, ~1 KB each through the CU's L1 path
//                    (64 B/clk) -- the kernel's co-bottleneck with the vector ALU (profiles/r03_pmc.txt: TA busy 72 %, VALU busy 69 %).
//   k_search1_tile   the block's 256 queries are consecutive in their own cell order, i.e. a strip of ~128 cells of the query grid; the
//                    dataset rows (y, z) they can touch form a small box, and of every such row they need one contiguous x-range of cells =
//                    one contiguous run of the record stream. The block finds those runs (min / max over its lanes per row), copies them
//                    into LDS with coalesced loads (~7 records per query instead of ~50 gathered), stages the rows' slice of cell_start
//                    next to them, and then every lane runs the usual scan on LDS addresses (one ds_read_b128 per candidate: the staged
//                    records are 16 bytes, row id included).
// A block whose runs do not fit the stage (dense regions of an uneven cloud) hands its queries to the wave-per-query pass.
// Open indexes only, like k_search1_flat.
#pragma once
#include "search.h"

namespace pcu {

__device__ unsigned long long g_tile_dbg[8];
#ifndef PCU_TILE_CAP
#define PCU_TILE_CAP 2560
#endif
#ifndef PCU_TILE_CS
#define PCU_TILE_CS 2560
#endif
constexpr int kTileCap = PCU_TILE_CAP;  // staged records per block (+ 4 sentinels): 40 KB
constexpr int kTileRows = 64;           // dataset rows (y, z) a block may touch
constexpr int kTileCs = PCU_TILE_CS;    // staged cell_start entries per block: 10 KB

template <typename T> struct TileRec;   // a staged record: coordinates + row id
template <> struct alignas(16) TileRec<float> { float x, y, z; int idx; };
template <> struct alignas(32) TileRec<double> { double x, y, z; long long idx; };

template <typename T, int FUSE>
__device__ __forceinline__ void search1_tile_body(const SearchArgs<T>& a, const int nq_arg, const int bid, const int nblk, bool& f_ok, T& f_v, long long& f_key) {
    __shared__ TileRec<T> s_rec[kTileCap + 4];
    __shared__ unsigned s_cs[kTileCs];
    __shared__ uint2 s_rng[8][kBlock];
    __shared__ int s_xa[kTileRows], s_xb[kTileRows];
    __shared__ unsigned s_gs[kTileRows], s_ls[kTileRows], s_cso[kTileRows], s_lo[kTileRows];
    __shared__ int s_box[kBlock / 64][4];      // per wave: min / max of ccy, ccz
    __shared__ int s_meta[8];                  // Y0, ny, Z0, nz, R, T, fallback
    const int per = nblk >> 3;
    const int vb = (bid & 7) * per + (bid >> 3);       // XCD-aware block order, see k_search
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = a.qcount_dev ? *a.qcount_dev : nq_arg;
    if (vb * kBlock >= nq) return;                     // (block-uniform)
    const GridParams<T>& g = *a.gp;
    if (const int hl = index_not_ready(a, g)) { if (vb == 0 && tid == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (vb == 0 && tid == 0) *a.skew_flag = 1; return; }
    const int t = vb * kBlock + tid;
    const bool valid = t < nq;                         // (the whole block stays: barriers)
    const int qpos = valid ? (a.qlist ? a.qlist[t] : t) : 0;
    Pt4<T> q;
    {
        struct __attribute__((packed, aligned(4))) Q3 { T v[3]; };
        const Q3 c = *reinterpret_cast<const Q3*>(a.q_xyz + 3 * (size_t)qpos);
        q.x = c.v[0]; q.y = c.v[1]; q.z = c.v[2];
        q.idx = FUSE == FUSE_SUM ? 0 : a.q_idx[qpos];
    }
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = grid_cell(g, 0, q.x), ccy = grid_cell(g, 1, q.y), ccz = grid_cell(g, 2, q.z);
    const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, Gx - 1);
    const int len = x1 - x0 + 1;
    // ---- the block's box of dataset rows
    {
        int ylo = valid ? ccy : 0x7fffffff, yhi = valid ? ccy : -1, zlo = valid ? ccz : 0x7fffffff, zhi = valid ? ccz : -1;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            ylo = min(ylo, __shfl_xor(ylo, o, 64)); yhi = max(yhi, __shfl_xor(yhi, o, 64));
            zlo = min(zlo, __shfl_xor(zlo, o, 64)); zhi = max(zhi, __shfl_xor(zhi, o, 64));
        }
        if (lane == 0) { s_box[wave][0] = ylo; s_box[wave][1] = yhi; s_box[wave][2] = zlo; s_box[wave][3] = zhi; }
        if (tid < kTileRows) { s_xa[tid] = 0x7fffffff; s_xb[tid] = -1; }
        __syncthreads();
        if (tid == 0) {
            int a0 = s_box[0][0], a1 = s_box[0][1], b0 = s_box[0][2], b1 = s_box[0][3];
            for (int w = 1; w < kBlock / 64; ++w) { a0 = min(a0, s_box[w][0]); a1 = max(a1, s_box[w][1]); b0 = min(b0, s_box[w][2]); b1 = max(b1, s_box[w][3]); }
            const int Y0 = max(a0 - 1, 0), Y1 = min(a1 + 1, Gy - 1), Z0 = max(b0 - 1, 0), Z1 = min(b1 + 1, Gz - 1);
            s_meta[0] = Y0; s_meta[1] = Y1 - Y0 + 1; s_meta[2] = Z0; s_meta[3] = Z1 - Z0 + 1;
            s_meta[4] = (Y1 - Y0 + 1) * (Z1 - Z0 + 1);
        }
        __syncthreads();
    }
    const int Y0 = s_meta[0], ny = s_meta[1], Z0 = s_meta[2], nz = s_meta[3], R = s_meta[4];
    const bool rows_ok = R <= kTileRows;               // (block-uniform)
    // ---- the x-range the block needs of every row: per wave, lanes with the same (ccy, ccz) are folded first, then <= 18 LDS atomics per group
    if (rows_ok) {
        unsigned long long todo = __ballot(valid);
        while (todo) {
            const int first = __ffsll((long long)todo) - 1;
            const int fy = __shfl(ccy, first, 64), fz = __shfl(ccz, first, 64);
            const bool mine = valid && ccy == fy && ccz == fz;
            const unsigned long long grp = __ballot(mine) & todo;
            int lo = mine ? x0 : 0x7fffffff, hi = mine ? x1 : -1;
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) { lo = min(lo, __shfl_xor(lo, o, 64)); hi = max(hi, __shfl_xor(hi, o, 64)); }
            if (lane < 9) {
                const int cy = fy + kRowOy[lane], cz = fz + kRowOz[lane];
                if (cy >= 0 && cy < Gy && cz >= 0 && cz < Gz) {
                    const int r = (cy - Y0) + (cz - Z0) * ny;
                    atomicMin(&s_xa[r], lo); atomicMax(&s_xb[r], hi);
                }
            }
            todo &= ~grp;
        }
    }
    __syncthreads();
    // ---- one wave lays out the stage: per row its run of the stream, its place in the stage, its slice of the cell table
    if (wave == 0) {
        unsigned cnt = 0, ncs = 0, gs = 0, lo = 0;
        if (rows_ok && lane < R && s_xb[lane] >= s_xa[lane]) {
            const int cy = Y0 + lane % ny, cz = Z0 + lane / ny;
            const int xa = s_xa[lane], xb = s_xb[lane];
            lo = (unsigned)row_run_lo(Gx, grid_row(Gy, cy, cz), xa, xb);
            gs = a.cell_start[lo];
            cnt = a.cell_start[lo + (unsigned)(xb - xa + 1)] - gs;
            ncs = (unsigned)(xb - xa + 2);
        }
        unsigned ic = cnt, is = ncs;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) { const unsigned tc = __shfl_up(ic, o, 64), ts = __shfl_up(is, o, 64); if (lane >= o) { ic += tc; is += ts; } }
        if (lane < kTileRows) { s_gs[lane] = gs; s_ls[lane] = ic - cnt; s_cso[lane] = is - ncs; s_lo[lane] = lo; }
        const unsigned Ttot = (unsigned)__shfl((int)ic, 63, 64), Ctot = (unsigned)__shfl((int)is, 63, 64);
        if (lane == 0) { s_meta[5] = (int)Ttot; s_meta[6] = (!rows_ok || Ttot > (unsigned)kTileCap || Ctot > (unsigned)kTileCs) ? 1 : 0; }
    }
    __syncthreads();
    const int Ttot = s_meta[5];
    if (tid == 0) { atomicAdd(&g_tile_dbg[0], 1ull); atomicAdd(&g_tile_dbg[1], (unsigned long long)Ttot); atomicAdd(&g_tile_dbg[2], (unsigned long long)s_meta[6]); atomicAdd(&g_tile_dbg[3], (unsigned long long)R); atomicMax(&g_tile_dbg[4], (unsigned long long)Ttot); if (!rows_ok) atomicAdd(&g_tile_dbg[5], 1ull); }
    if (s_meta[6]) {
        // the runs do not fit the stage: the wave-per-query pass serves these queries (as a lane that defers does, see k_search1_flat)
        wave_append(valid, qpos, a.ties, a.n_ties);
        return;
    }
    // ---- staging: rows dealt to the waves; the cell table as stage positions, the records with their row ids
    for (int r = wave; r < R; r += kBlock / 64) {
        const int xa = s_xa[r], xb = s_xb[r];
        if (xb < xa) continue;
        const unsigned gs = s_gs[r], ls = s_ls[r], cso = s_cso[r], lo = s_lo[r];
        const unsigned ncs = (unsigned)(xb - xa + 2);
        for (unsigned i = (unsigned)lane; i < ncs; i += 64u) s_cs[cso + i] = ls + (a.cell_start[lo + i] - gs);
        const unsigned cnt = (r + 1 < R ? s_ls[r + 1] : (unsigned)Ttot) - ls;       // (rows without a run repeat their predecessor's end)
        struct __attribute__((packed, aligned(4))) Q3 { T v[3]; };
        for (unsigned k = (unsigned)lane; k < cnt; k += 64u) {
            const Q3 c = *reinterpret_cast<const Q3*>(a.ref_xyz + 3 * (size_t)(gs + k));
            TileRec<T> rec; rec.x = c.v[0]; rec.y = c.v[1]; rec.z = c.v[2]; rec.idx = a.ref_idx[gs + k];
            s_rec[ls + k] = rec;
        }
    }
    if (tid < 4) { TileRec<T> rec; rec.x = rec.y = rec.z = (T)INFINITY; rec.idx = 0x7fffffff; s_rec[Ttot + tid] = rec; }
    __syncthreads();
    if (!valid) return;
    // ---- the lane's scan, on stage positions (cf. search1_flat_body: the same steps in the same order)
    constexpr int kG = 4;
    const unsigned cand_cap = a.lane_max_cand < 65535u ? a.lane_max_cand : 65535u;
    T best = Limits<T>::max_v;
    unsigned boff = 0xffffffffu, toff = 0xffffffffu;
    bool tie = false, tie2 = false;
    struct Grp { TileRec<T> c[4]; };
    auto load = [&](unsigned pos) { Grp r; r.c[0] = s_rec[pos]; r.c[1] = s_rec[pos + 1]; r.c[2] = s_rec[pos + 2]; r.c[3] = s_rec[pos + 3]; return r; };
    auto dists = [&](const Grp& r, T (&d)[4]) {
#pragma unroll
        for (int u = 0; u < 4; ++u) { const T dx = q.x - r.c[u].x, dy = q.y - r.c[u].y, dz = q.z - r.c[u].z; d[u] = ((dx * dx) + (dy * dy)) + (dz * dz); }
    };
#define PCU_T1_EVAL(RAW, OFF)                                                                                \
    {                                                                                                        \
        T d_[4]; dists((RAW), d_);                                                                           \
        const T m_ = min4(d_[0], d_[1], d_[2], d_[3]);                                                       \
        const bool eq_ = m_ == best, lt_ = m_ < best;                                                        \
        tie2 = !lt_ && (tie2 || (tie && eq_));                                                               \
        tie = !lt_ && (tie || eq_);                                                                          \
        toff = eq_ ? (OFF) : toff;                                                                           \
        best = lt_ ? m_ : best;                                                                              \
        boff = lt_ ? (OFF) : boff;                                                                           \
    }
    // a row's table: stage positions of the run of the cells x0 .. x1 in stream order -- its start, the second cell's start, the last cell's
    // start and the end (selected by `len` here: an array indexed by it would live in scratch)
    struct RowTab { unsigned s, c1, ec, ef; };
    auto row_table = [&](int j, bool& ok, bool& odd) -> RowTab {
        const int cy = ccy + kRowOy[j], cz = ccz + kRowOz[j];
        ok = cy >= 0 && cy < Gy && cz >= 0 && cz < Gz;
        const int ry = ok ? cy : ccy, rz = ok ? cz : ccz;
        const int r = (ry - Y0) + (rz - Z0) * ny;
        odd = (unsigned)grid_row(Gy, ry, rz) & 1u;
        const unsigned i0 = s_cso[r] + (unsigned)(odd ? s_xb[r] - x1 : x0 - s_xa[r]);
        const unsigned v0 = s_cs[i0], v1 = s_cs[i0 + 1], v2 = len >= 2 ? s_cs[i0 + 2] : 0u, v3 = len >= 3 ? s_cs[i0 + 3] : 0u;
        RowTab t;
        t.s = v0; t.c1 = v1;
        t.ef = len == 3 ? v3 : (len == 2 ? v2 : v1);
        t.ec = len == 3 ? v2 : (len == 2 ? v1 : v0);
        return t;
    };
    bool okj[9], oddj[9];
    RowTab tb[9];
    tb[0] = row_table(0, okj[0], oddj[0]);
    const unsigned cnt0 = tb[0].ef - tb[0].s;
    bool defer = cnt0 > cand_cap;
    {
        const unsigned o1 = defer ? tb[0].s : tb[0].s + cnt0;
        for (unsigned off = tb[0].s; off < o1; off += (unsigned)kG) { const Grp raw = load(off); PCU_T1_EVAL(raw, off) }
    }
#pragma unroll
    for (int j = 1; j < 9; ++j) tb[j] = row_table(j, okj[j], oddj[j]);
    const T shrink = (T)1 - (T)4 * Limits<T>::eps;
    T mxl = q.x - face_below(g, 0, ccx); mxl = mxl > (T)0 ? mxl * shrink : (T)0;
    T mxh = face_above(g, 0, ccx) - q.x; mxh = mxh > (T)0 ? mxh * shrink : (T)0;
    const T mxl2 = mxl * mxl, mxh2 = mxh * mxh;
    const bool has_lo = x0 < ccx, has_hi = x1 > ccx;
    T my2[3], mz2[3];
    {
        T m;
        my2[0] = (T)0; mz2[0] = (T)0;
        m = q.y - face_below(g, 1, ccy); m = m > (T)0 ? m * shrink : (T)0; my2[1] = m * m;
        m = face_above(g, 1, ccy) - q.y; m = m > (T)0 ? m * shrink : (T)0; my2[2] = m * m;
        m = q.z - face_below(g, 2, ccz); m = m > (T)0 ? m * shrink : (T)0; mz2[1] = m * m;
        m = face_above(g, 2, ccz) - q.z; m = m > (T)0 ? m * shrink : (T)0; mz2[2] = m * m;
    }
    unsigned total = cnt0;
#pragma unroll
    for (int j = 1; j < 9; ++j) total += okj[j] ? tb[j].ef - tb[j].s : 0u;
    defer = defer || total > cand_cap;
    int n = 0;
#pragma unroll
    for (int j = 1; j < 9; ++j) {
        const T ry = my2[kRowOy[j] == 0 ? 0 : (kRowOy[j] < 0 ? 1 : 2)], rz = mz2[kRowOz[j] == 0 ? 0 : (kRowOz[j] < 0 ? 1 : 2)];
        const T rlb = ry + rz;
        const bool cut_lo = has_lo && best < ((mxl2 + ry) + rz), cut_hi = has_hi && best < ((mxh2 + ry) + rz);
        const bool cut_first = oddj[j] ? cut_hi : cut_lo, cut_last = oddj[j] ? cut_lo : cut_hi;
        const unsigned s_run = cut_first ? tb[j].c1 : tb[j].s;
        const unsigned e_run = cut_last ? tb[j].ec : tb[j].ef;
        if (okj[j] && !defer && !(best < rlb) && e_run > s_run) {
            s_rng[n][tid] = make_uint2(s_run, ((e_run - s_run) << 16) | lb_pack(rlb));
            ++n;
        }
    }
    int r = 0;
    unsigned off = 0, end = 0;
    bool live = false;
    auto next_run = [&]() {
        live = false;
        while (r < n) {
            const uint2 e = s_rng[r][tid];
            ++r;
            if (!(best < lb_unpack<T>(e.y & 0xffffu))) { off = e.x; end = e.x + (e.y >> 16); live = true; break; }
        }
    };
    next_run();
    const unsigned sent_off = (unsigned)Ttot;
    if (live) {
        Grp ga = load(off), gb;
        for (;;) {
            unsigned coff = off;
            off += (unsigned)kG;
            if (off >= end) next_run();
            gb = load(live ? off : sent_off);
            PCU_T1_EVAL(ga, coff)
            if (!live) break;
            coff = off;
            off += (unsigned)kG;
            if (off >= end) next_run();
            ga = load(live ? off : sent_off);
            PCU_T1_EVAL(gb, coff)
            if (!live) break;
        }
    }
#undef PCU_T1_EVAL
    T bd[1] = {best};
    int bi[1] = {0x7fffffff};
    auto which = [&](unsigned goff, int& hits, unsigned& rec_off) {
        T d_[4]; dists(load(goff), d_);
        hits = 0; rec_off = 0xffffffffu;
#pragma unroll
        for (int u = kG - 1; u >= 0; --u) { const bool eq = d_[u] == best; hits += eq ? 1 : 0; rec_off = eq ? goff + (unsigned)u : rec_off; }
    };
    if (FUSE != FUSE_SUM && boff != 0xffffffffu) {
        int hits; unsigned ro;
        which(boff, hits, ro);
        if (ro != 0xffffffffu) bi[0] = (int)s_rec[ro].idx;
        if (hits > 1) { tie = true; tie2 = true; }
        if (tie && !tie2) {
            int h2; unsigned ro2;
            which(toff, h2, ro2);
            if (h2 == 1 && ro2 == ro) tie = false;
        }
    }
    const int y0 = max(ccy - 1, 0), y1 = min(ccy + 1, Gy - 1);
    const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, Gz - 1);
    if (FUSE == FUSE_NONE) { finish_lane<T, 1>(a, g, q, qpos, x0, x1, y0, y1, z0, z1, bd, bi, tie, true, defer); return; }
    if (defer) { wave_append(true, qpos, a.ties, a.n_ties); return; }
    const T lb = face_lower_bound_inner(g, q.x, q.y, q.z, x0, x1, y0, y1, z0, z1);
    const bool certified = best < lb;
    const int us = wave_append(!certified, qpos, a.unresolved, a.n_unresolved);
    if (us >= 0 && a.ubound) a.ubound[us] = best;
    f_ok = certified;
    f_v = a.squared ? best : sqrt(best);
    f_key = ((long long)q.idx << 32) | (long long)((unsigned)bi[0] | (tie ? 0x80000000u : 0u));
}

template <typename T, int FUSE>
__global__ __launch_bounds__(kBlock) void k_search1_tile(const SearchArgs2<T> p, int nb0) {
    const int side = (int)blockIdx.x >= nb0 ? 1 : 0;
    const int bid = side ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    bool ok = false; T v = (T)0; long long key = 0x7fffffffffffffffll;
    const int nq_side = side ? p.a[1].nq : p.a[0].nq;
    search1_tile_body<T, FUSE>(p.a[side], nq_side, bid, side ? (int)gridDim.x - nb0 : nb0, ok, v, key);
    if (FUSE == FUSE_SUM) {
        const double r = block_sum(ok ? (double)v : 0.0);
        if (threadIdx.x == 0) p.a[side].f_sum[bid] = r;
    } else if (FUSE == FUSE_ARGMAX) {
        T bv = ok ? v : -Limits<T>::max_v; long long bk = ok ? key : 0x7fffffffffffffffll;
        block_argmax(bv, bk);
        if (threadIdx.x == 0) { p.a[side].f_max_v[bid] = bv; p.a[side].f_max_k[bid] = bk; }
    }
}

}  // namespace pcu
