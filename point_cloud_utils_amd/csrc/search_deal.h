// csrc/search_deal.h -- k = 1 main pass with the candidate groups of a WAVE dealt evenly to its lanes (fused Chamfer sum only).
//
// k_search1_flat (search.h) gives every lane its own query and lets it walk its own surviving runs: a lane evaluates ~6 groups of 4 candidates
// on average, but a wave runs until its slowest lane is done -- ~15 trips of the loop -- and pays for the divergence in scalar exec-mask
// bookkeeping (profiles/r03_pmc.txt: 696 scalar instructions per wave beside 1175 vector ones, 68 % active lanes). Here the lanes only
// DECIDE what their queries need -- centre row, cuts, surviving runs, exactly as in k_search1_flat -- and write the groups into one list per
// wave (LDS; slots handed out by a wave-wide prefix sum). The list is then consumed by all 64 lanes in lock step, item w by lane w mod 64:
// a lane evaluates a group against the query of the lane that listed it (coordinates from LDS) and merges the group's minimum into that
// query's best with an LDS atomic min on the float's bits (distances are >= 0: the bit patterns order like the values). ceil(W / 64) uniform
// trips instead of max-over-lanes, no per-lane control flow in the hot loop, the same candidates, the same arithmetic, the same minimum.
// The sum needs neither the winner's row nor tie flags; the other epilogues keep k_search1_flat.
#pragma once
#include "search.h"

namespace pcu {

constexpr int kDealCap = 768;          // work items (groups) per wave: twice the mean of a uniform cloud; lanes whose groups do not fit are deferred

template <typename T> struct BitsOf;
template <> struct BitsOf<float> { typedef unsigned type; static __device__ __forceinline__ unsigned of(float v) { return __float_as_uint(v); } static __device__ __forceinline__ float back(unsigned b) { return __uint_as_float(b); } };
template <> struct BitsOf<double> { typedef unsigned long long type; static __device__ __forceinline__ unsigned long long of(double v) { return (unsigned long long)__double_as_longlong(v); } static __device__ __forceinline__ double back(unsigned long long b) { return __longlong_as_double((long long)b); } };

template <typename T>
__device__ __forceinline__ void search1_deal_body(const SearchArgs<T>& a, const int nq_arg, const int bid, const int nblk, bool& f_ok, T& f_v) {
    typedef GroupEval<T, true> GE;
    typedef typename BitsOf<T>::type Bits;
    __shared__ unsigned s_off[kBlock / 64][kDealCap];
    __shared__ unsigned char s_ql[kBlock / 64][kDealCap];
    __shared__ Bits s_best[kBlock / 64][64];
    __shared__ T s_q[kBlock / 64][3][64];
    const int per = nblk >> 3;
    const int vb = (bid & 7) * per + (bid >> 3);       // XCD-aware block order, see k_search
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int nq = a.qcount_dev ? *a.qcount_dev : nq_arg;
    if (vb * kBlock >= nq) return;                     // (block-uniform)
    const GridParams<T>& g = *a.gp;
    if (const int hl = index_not_ready(a, g)) { if (vb == 0 && tid == 0) a.skew_flag[kLargeFlag] = hl; return; }
    if (a.skew_limit > 0.f && ((float)g.sumsq > a.skew_limit || (float)g.sumsq < a.skew_lo)) { if (vb == 0 && tid == 0) *a.skew_flag = 1; return; }
    const int t = vb * kBlock + tid;
    if ((t & ~63) >= nq) return;                       // (wave-uniform: a wave without queries)
    const bool valid = t < nq;
    const int qpos = valid ? (a.qlist ? a.qlist[t] : t) : 0;
    Pt4<T> q;
    {
        struct __attribute__((packed, aligned(4))) Q3 { T v[3]; };
        const Q3 c = *reinterpret_cast<const Q3*>(a.q_xyz + 3 * (size_t)qpos);
        q.x = c.v[0]; q.y = c.v[1]; q.z = c.v[2]; q.idx = 0;
    }
    s_q[wave][0][lane] = q.x; s_q[wave][1][lane] = q.y; s_q[wave][2][lane] = q.z;
    const int Gx = g.G[0], Gy = g.G[1], Gz = g.G[2];
    const int ccx = grid_cell(g, 0, q.x), ccy = grid_cell(g, 1, q.y), ccz = grid_cell(g, 2, q.z);
    const int x0 = max(ccx - 1, 0), x1 = min(ccx + 1, Gx - 1);
    const int len = x1 - x0 + 1;
    constexpr unsigned kRec = GE::kRec;
    constexpr int kG = K1Group<T>::n;
    const char* const base = reinterpret_cast<const char*>(a.ref_xyz);
    const unsigned cand_cap = a.lane_max_cand < 65535u ? a.lane_max_cand : 65535u;
    T best = Limits<T>::max_v;
    auto eval = [&](const typename GE::Raw& raw, const Pt4<T>& qq) -> T { T d_[4]; GE::dists(raw, qq, d_); return min4(d_[0], d_[1], d_[2], d_[3]); };
    auto row_table = [&](int j, bool& ok, bool& odd) {
        const int cy = ccy + kRowOy[j], cz = ccz + kRowOz[j];
        ok = cy >= 0 && cy < Gy && cz >= 0 && cz < Gz;
        const unsigned row = (unsigned)(ok ? cz : ccz) * (unsigned)Gy + (unsigned)(((ok ? cz : ccz) & 1) ? Gy - 1 - (ok ? cy : ccy) : (ok ? cy : ccy));
        odd = row & 1u;
        const unsigned lo = __umul24(row, (unsigned)Gx) + (unsigned)(odd ? Gx - 1 - x1 : x0);
        return *reinterpret_cast<const CellStart4*>(reinterpret_cast<const char*>(a.cell_start) + (size_t)(lo * 4u));
    };
    bool okj[9], oddj[9];
    CellStart4 tb[9];
#pragma unroll
    for (int j = 0; j < 9; ++j) tb[j] = row_table(j, okj[j], oddj[j]);
    // ---- centre row, by its own lane: the first estimate that prunes the other rows
    const unsigned cnt0 = (len == 3 ? tb[0].v[3] : (len == 2 ? tb[0].v[2] : tb[0].v[1])) - tb[0].v[0];
    bool defer = cnt0 > cand_cap;
    {
        const unsigned o0 = tb[0].v[0] * kRec;
        const unsigned o1 = (defer || !valid) ? o0 : o0 + cnt0 * kRec;
        for (unsigned off = o0; off < o1; off += (unsigned)kG * kRec) { const T m = eval(GE::load(base, off), q); best = m < best ? m : best; }
    }
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
    for (int j = 1; j < 9; ++j) total += okj[j] ? (len == 3 ? tb[j].v[3] : (len == 2 ? tb[j].v[2] : tb[j].v[1])) - tb[j].v[0] : 0u;
    defer = defer || total > cand_cap;
    // ---- surviving cut runs of the other eight rows (as in k_search1_flat), counted in groups
    unsigned rs[8], rg[8];
    unsigned n = 0;
#pragma unroll
    for (int j = 1; j < 9; ++j) {
        const T ry = my2[kRowOy[j] == 0 ? 0 : (kRowOy[j] < 0 ? 1 : 2)], rz = mz2[kRowOz[j] == 0 ? 0 : (kRowOz[j] < 0 ? 1 : 2)];
        const T rlb = ry + rz;
        const bool cut_lo = has_lo && best < ((mxl2 + ry) + rz), cut_hi = has_hi && best < ((mxh2 + ry) + rz);
        const bool cut_first = oddj[j] ? cut_hi : cut_lo, cut_last = oddj[j] ? cut_lo : cut_hi;
        const unsigned s_run = cut_first ? tb[j].v[1] : tb[j].v[0];
        const unsigned e_full = len == 3 ? tb[j].v[3] : (len == 2 ? tb[j].v[2] : tb[j].v[1]);
        const unsigned e_cut = len == 3 ? tb[j].v[2] : (len == 2 ? tb[j].v[1] : tb[j].v[0]);
        const unsigned e_run = cut_last ? e_cut : e_full;
        const bool on = valid && okj[j] && !defer && !(best < rlb) && e_run > s_run;
        rs[j - 1] = s_run * kRec;
        rg[j - 1] = on ? (e_run - s_run + (unsigned)kG - 1u) / (unsigned)kG : 0u;
        n += rg[j - 1];
    }
    // ---- the wave's list: slots by a prefix sum over the lanes; lanes whose groups do not fit (dense regions) hand their queries to the wave pass
    unsigned inc = n;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) { const unsigned u = __shfl_up(inc, o, 64); if (lane >= o) inc += u; }
    const bool fits = inc <= (unsigned)kDealCap;                       // (inc is monotone over the lanes: the lanes that fit are a prefix of the wave)
    const unsigned long long fm = __ballot(fits);
    const unsigned W = fm ? (unsigned)__shfl((int)inc, 63 - __clzll((long long)fm), 64) : 0u;        // items listed
    if (!fits) { defer = true; n = 0; }
    unsigned slot = inc - n;
    s_best[wave][lane] = BitsOf<T>::of(best);
    if (n) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            unsigned off = rs[j];
            for (unsigned gq = 0; gq < rg[j]; ++gq) { s_off[wave][slot] = off; s_ql[wave][slot] = (unsigned char)lane; ++slot; off += (unsigned)kG * kRec; }
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    // ---- all lanes, 64 items per trip; the next trip's groups are requested before the current ones are evaluated (two register sets, no copies)
    const unsigned sent_off = a.n_ref * kRec;
#define PCU_DEAL_ITEM(WB, OFF, QL) { const unsigned w_ = (WB) + (unsigned)lane; const bool in_ = w_ < W; OFF = in_ ? s_off[wave][in_ ? w_ : 0u] : sent_off; QL = in_ ? (int)s_ql[wave][in_ ? w_ : 0u] : -1; }
#define PCU_DEAL_EVAL(RAW, QL) { if ((QL) >= 0) { Pt4<T> qq; qq.x = s_q[wave][0][QL]; qq.y = s_q[wave][1][QL]; qq.z = s_q[wave][2][QL]; \
                                     atomicMin(&s_best[wave][QL], BitsOf<T>::of(eval((RAW), qq))); } }
    if (W) {
        unsigned wb = 0, off_a, off_b; int ql_a, ql_b;
        PCU_DEAL_ITEM(0u, off_a, ql_a)
        typename GE::Raw ga = GE::load(base, off_a), gb;
        for (;;) {
            wb += 64u;
            const bool more_b = wb < W;                                // (uniform)
            if (more_b) { PCU_DEAL_ITEM(wb, off_b, ql_b) gb = GE::load(base, off_b); }
            PCU_DEAL_EVAL(ga, ql_a)
            if (!more_b) break;
            wb += 64u;
            const bool more_a = wb < W;
            if (more_a) { PCU_DEAL_ITEM(wb, off_a, ql_a) ga = GE::load(base, off_a); }
            PCU_DEAL_EVAL(gb, ql_b)
            if (!more_a) break;
        }
    }
#undef PCU_DEAL_ITEM
#undef PCU_DEAL_EVAL
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    best = BitsOf<T>::back(s_best[wave][lane]);
    if (!valid) return;
    // ---- certification and the fused sum's share (cf. search1_flat_body)
    const int y0 = max(ccy - 1, 0), y1 = min(ccy + 1, Gy - 1);
    const int z0 = max(ccz - 1, 0), z1 = min(ccz + 1, Gz - 1);
    if (defer) { wave_append(true, qpos, a.ties, a.n_ties); return; }
    const T lb = face_lower_bound_inner(g, q.x, q.y, q.z, x0, x1, y0, y1, z0, z1);
    const bool certified = best < lb;
    const int us = wave_append(!certified, qpos, a.unresolved, a.n_unresolved);
    if (us >= 0 && a.ubound) a.ubound[us] = best;
    f_ok = certified;
    f_v = a.squared ? best : sqrt(best);
}

template <typename T>
__global__ __launch_bounds__(kBlock) void k_search1_deal(const SearchArgs2<T> p, int nb0) {
    const int side = (int)blockIdx.x >= nb0 ? 1 : 0;
    const int bid = side ? (int)blockIdx.x - nb0 : (int)blockIdx.x;
    bool ok = false; T v = (T)0;
    const int nq_side = side ? p.a[1].nq : p.a[0].nq;
    search1_deal_body<T>(p.a[side], nq_side, bid, side ? (int)gridDim.x - nb0 : nb0, ok, v);
    const double r = block_sum(ok ? (double)v : 0.0);
    if (threadIdx.x == 0) p.a[side].f_sum[bid] = r;
}

}  // namespace pcu
