// csrc/voxel_host.h -- host orchestration of the SURVEY.md 8f-4 operators (included by pcu_hip.hip after the context / arena
// helpers): Morton codes (morton.h), voxel-grid downsampling and duplicate removal (voxel.h).
#pragma once

template <typename U>
static int stage_any(Arena& ar, const U* p, size_t count, bool on_dev, hipStream_t s, const U** out) {
    if (on_dev) { *out = p; return 0; }
    U* d = nullptr;
    if (aalloc(ar, &d, count)) return -1;
    HIP_TRY(hipMemcpyAsync(d, p, count * sizeof(U), hipMemcpyHostToDevice, s));
    *out = d;
    return 0;
}
static hipStream_t pick_stream(pcu_hip_ctx* c, unsigned flags, void* stream) {
    return (stream || (flags & PCU_HIP_STREAM_GIVEN)) ? (hipStream_t)stream : c->own_stream;
}

// ---------------------------------------------------------------------------------------------------- Morton codes
// kind: 0 encode (in: I (n,3) -> out u64 (n)), 1 decode (in: C (n) -> out i32 (n,3)), 2 add, 3 subtract (in, in2: C (n) -> u64 (n))
template <typename In, typename In2>
static int morton_map_impl(pcu_hip_ctx* c, int kind, const In* in, const In2* in2, int64_t n, void* out, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (n <= 0) return fail(PCU_HIP_ERR_INVALID, kind == 0 ? "pts must be an array of shape [n, 3] but got an empty array"
                                                 : kind == 1 ? "codes must be an array of shape [n] but got an empty array"
                                                             : "codes_1 must be an array of shape [n,] but got an empty array");
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    const size_t in_count = kind == 0 ? (size_t)n * 3 : (size_t)n, out_bytes = kind == 1 ? (size_t)n * 12 : (size_t)n * 8;
    if (ctx_begin(c, on_dev ? 4096 : align_up(in_count * sizeof(In), 256) + align_up((size_t)n * sizeof(In2), 256) + align_up(out_bytes, 256) + 4096)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const In* d_in; const In2* d_in2 = in2;
        if ((rc = stage_any(ar, in, in_count, on_dev, s, &d_in))) break;
        if (kind >= 2 && (rc = stage_any(ar, in2, (size_t)n, on_dev, s, &d_in2))) break;
        void* d_out = out;
        if (!on_dev) { char* t = nullptr; if ((rc = aalloc(ar, &t, out_bytes))) break; d_out = t; }
        const int blocks = (int)std::min<int64_t>((n + 255) / 256, 65536);
        if (kind == 0) hipLaunchKernelGGL((k_morton_encode<In>), dim3(blocks), dim3(256), 0, s, d_in, (long long)n, (uint64_t*)d_out);
        else if (kind == 1) hipLaunchKernelGGL((k_morton_decode<In>), dim3(blocks), dim3(256), 0, s, d_in, (long long)n, (int32_t*)d_out);
        else hipLaunchKernelGGL((k_morton_addsub<In, In2>), dim3(blocks), dim3(256), 0, s, d_in, d_in2, (long long)n, kind == 3 ? 1 : 0, (uint64_t*)d_out);
        HIP_TRY(hipGetLastError());
        if (!on_dev) HIP_TRY(hipMemcpyAsync(out, d_out, out_bytes, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename C>
static int morton_knn_impl(pcu_hip_ctx* c, const C* codes, int64_t n, const C* qcodes, int64_t m, int k, int sort_dist, int64_t* out_nn, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (k <= 0) return fail(PCU_HIP_ERR_INVALID, "k must be greater than 0");
    if (n <= 0 || m <= 0) return fail(PCU_HIP_ERR_INVALID, "codes must be an array of shape [n] but got an empty array");
    if (k > n) k = (int)n;                                   // k = std::min(k, (int)codes.rows()), src/morton.cpp:362
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    if (ctx_begin(c, on_dev ? 4096 : align_up((size_t)n * sizeof(C), 256) + align_up((size_t)m * sizeof(C), 256) + align_up((size_t)m * k * 8, 256) + 4096)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const C *d_codes, *d_q;
        if ((rc = stage_any(ar, codes, (size_t)n, on_dev, s, &d_codes)) || (rc = stage_any(ar, qcodes, (size_t)m, on_dev, s, &d_q))) break;
        long long* d_nn = (long long*)out_nn;
        if (!on_dev && (rc = aalloc(ar, &d_nn, (size_t)m * k))) break;
        hipLaunchKernelGGL((k_morton_knn<C>), dim3((unsigned)((m + 255) / 256)), dim3(256), 0, s, d_codes, (long long)n, d_q, (long long)m, k, sort_dist, d_nn);
        HIP_TRY(hipGetLastError());
        if (!on_dev) HIP_TRY(hipMemcpyAsync(out_nn, d_nn, (size_t)m * k * 8, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ---------------------------------------------------------------------------------------------------- sort by a key triple
// inclusive scan of n unsigned values (radix.h: tiles, one block over the tile sums, add)
static int own_inclusive_scan(Arena& ar, hipStream_t s, const unsigned* in, unsigned* out, size_t n) {
    const int nt = (int)((n + kScTile - 1) / kScTile);
    unsigned* sums = nullptr;
    if (aalloc(ar, &sums, (size_t)nt + 1)) return -1;
    hipLaunchKernelGGL(k_sc_tiles, dim3(nt), dim3(1024), 0, s, in, out, (int)n, sums);
    if (nt > 1) {
        hipLaunchKernelGGL(k_sc_sums, dim3(1), dim3(1024), 0, s, sums, nt);
        hipLaunchKernelGGL(k_sc_add, dim3(nt), dim3(1024), 0, s, out, (int)n, sums);
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
// Stable LSD radix sort of (key, id) pairs by the low `bits` bits of the keys (radix.h); ids == nullptr on entry: the identity. On return
// *keys / *ids point at the buffers that hold the result (the pairs ping-pong between the given buffers).
static int own_radix_sort(Arena& ar, hipStream_t s, unsigned long long** keys, unsigned long long** keys_alt, unsigned** ids, unsigned** ids_alt, bool ids_identity, int n, int bits) {
    const int nwt = (n + kRsWaveTile - 1) / kRsWaveTile;
    unsigned *hist = nullptr, *total = nullptr;
    if (aalloc(ar, &hist, (size_t)256 * nwt) || aalloc(ar, &total, 256)) return -1;
    const int nblk = (nwt + kRsThreads / 64 - 1) / (kRsThreads / 64);
    for (int shift = 0; shift < bits; shift += 8) {
        hipLaunchKernelGGL(k_rs_hist, dim3(nblk), dim3(kRsThreads), 0, s, *keys, n, shift, nwt, hist);
        hipLaunchKernelGGL(k_rs_scan_rows, dim3(256), dim3(1024), 0, s, hist, nwt, total);
        hipLaunchKernelGGL(k_rs_scatter, dim3(nblk), dim3(kRsThreads), 0, s, *keys, ids_identity ? (const unsigned*)nullptr : *ids, n, shift, nwt, hist, total, *keys_alt, *ids_alt);
        std::swap(*keys, *keys_alt); std::swap(*ids, *ids_alt);
        ids_identity = false;
    }
    HIP_TRY(hipGetLastError());
    return 0;
}
static int bits_of(unsigned long long v) { int b = 0; while (v) { ++b; v >>= 1; } return b; }
// perm_out: point ids ordered lexicographically by (k0, k1, k2), equal triples in input order. The components are packed, each less its
// minimum, into one key of as many bits as their ranges need and sorted by the repo's own stable radix passes (radix.h); triples wider than
// 64 bits in total are sorted component by component, least significant first. One host read-back (the components' ranges).
template <typename K>
static int sort_by_triple(Arena& ar, hipStream_t s, const K* k0, const K* k1, const K* k2, unsigned* ids, int n, unsigned** perm_out, const unsigned long long** sorted_keys = nullptr) {
    if (sorted_keys) *sorted_keys = nullptr;
    KeyRange* rng = nullptr;
    unsigned long long *ka = nullptr, *kb = nullptr; unsigned* alt = nullptr;
    if (aalloc(ar, &rng, 1) || aalloc(ar, &ka, (size_t)n) || aalloc(ar, &kb, (size_t)n) || aalloc(ar, &alt, (size_t)n)) return -1;
    HIP_TRY(hipMemsetAsync(rng->lo, 0xff, sizeof rng->lo, s));
    HIP_TRY(hipMemsetAsync(rng->hi, 0, sizeof rng->hi, s));
    const int nb = (n + kBlock - 1) / kBlock;
    hipLaunchKernelGGL((k_key_range<K>), dim3(std::min(nb, 256)), dim3(kBlock), 0, s, k0, k1, k2, n, rng);
    KeyRange h;
    HIP_TRY(hipMemcpyAsync(&h, rng, sizeof h, hipMemcpyDeviceToHost, s));
    HIP_WAIT(s);
    const int w[3] = {bits_of(h.hi[0] - h.lo[0]), bits_of(h.hi[1] - h.lo[1]), bits_of(h.hi[2] - h.lo[2])};
    unsigned *cur = ids, *nxt = alt;
    if (w[0] + w[1] + w[2] <= 64) {
        hipLaunchKernelGGL((k_key_pack<K>), dim3(nb), dim3(kBlock), 0, s, k0, k1, k2, n, rng, w[1], w[2], ka);
        if (own_radix_sort(ar, s, &ka, &kb, &cur, &nxt, /*ids_identity=*/true, n, w[0] + w[1] + w[2])) return -1;      // (0 bits: every triple equal, ids stay the identity)
        if (sorted_keys) *sorted_keys = ka;
    } else {
        const K* comps[3] = {k2, k1, k0};
        bool identity = true;
        for (int pass = 0; pass < 3; ++pass) {
            hipLaunchKernelGGL((k_key_gather<K>), dim3(nb), dim3(kBlock), 0, s, comps[pass], identity ? (const unsigned*)nullptr : cur, n, rng, 2 - pass, ka);
            if (own_radix_sort(ar, s, &ka, &kb, &cur, &nxt, identity, n, w[2 - pass])) return -1;
            if (w[2 - pass] > 0) identity = false;
        }
    }
    HIP_TRY(hipGetLastError());
    *perm_out = cur;
    return 0;
}
// head flags + inclusive scan of a sorted order -> (flag, scan); the number of runs is scan[n-1]
template <typename K>
static int runs_of(Arena& ar, hipStream_t s, const K* k0, const K* k1, const K* k2, const unsigned* perm, int n, unsigned** flag_out, unsigned** scan_out, const unsigned long long* sorted_keys = nullptr) {
    unsigned *flag = nullptr, *scan = nullptr;
    if (aalloc(ar, &flag, (size_t)n) || aalloc(ar, &scan, (size_t)n + 1)) return -1;
    if (sorted_keys) hipLaunchKernelGGL(k_run_heads_sorted, dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, sorted_keys, n, flag);
    else hipLaunchKernelGGL((k_run_heads<K>), dim3((n + kBlock - 1) / kBlock), dim3(kBlock), 0, s, k0, k1, k2, perm, n, flag);
    if (own_inclusive_scan(ar, s, flag, scan, (size_t)n)) return -1;
    *flag_out = flag; *scan_out = scan;
    return 0;
}

// ---------------------------------------------------------------------------------------------------- voxel-grid downsampling
// downsample_point_cloud_voxel_grid_internal (src/sample_point_cloud.cpp:336-368). out_v (n,3) / out_a (n,cols): caller-sized for the
// worst case; *out_count = number of voxels written.
template <typename T, typename A>
static int voxel_downsample_impl(pcu_hip_ctx* c, const T* pts, int64_t n, const A* attrib, int64_t attrib_rows, int cols, const double* vsize, const double* vmin,
                                 const double* vmax, int min_pts, T* out_v, A* out_a, int64_t* out_count, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (n <= 0) { *out_count = 0; return 0; }
    if (n > 0x7ffffff0ll) return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^31-16 rows are not supported");
    const T vs[3] = {(T)vsize[0], (T)vsize[1], (T)vsize[2]}, mn[3] = {(T)vmin[0], (T)vmin[1], (T)vmin[2]}, mx[3] = {(T)vmax[0], (T)vmax[1], (T)vmax[2]};
    for (int i = 0; i < 3; ++i) {                            // :174-181
        if (vs[i] <= 0.0) return fail(PCU_HIP_ERR_INVALID, "Voxel size is negative");
        if (vs[i] * (T)std::numeric_limits<int>::max() < (T)(mx[i] - mn[i])) return fail(PCU_HIP_ERR_INVALID, "Voxel size is too small");
    }
    const bool has_attr = attrib && attrib_rows != 0 && cols != 0;
    if (has_attr && attrib_rows != n)
        return fail(PCU_HIP_ERR_INVALID, "Invalid number of attributes (%lld). Must match number of input vertices (%lld) or be 0.", (long long)attrib_rows, (long long)n);
    if (!has_attr) cols = 0;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    const size_t N = (size_t)n;
    size_t need = 20 * align_up(N * 4, 256) + (1 << 20) + (on_dev ? 0 : 2 * (align_up(N * 3 * sizeof(T), 256) + align_up(N * (size_t)cols * sizeof(A), 256)));
    if (ctx_begin(c, need + 65536)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T* d_pts; const A* d_attr = attrib;
        if ((rc = stage_any(ar, pts, N * 3, on_dev, s, &d_pts))) break;
        if (has_attr && (rc = stage_any(ar, attrib, N * (size_t)cols, on_dev, s, &d_attr))) break;
        T* d_out_v = out_v; A* d_out_a = out_a;
        if (!on_dev) { if ((rc = aalloc(ar, &d_out_v, N * 3))) break; if (has_attr && (rc = aalloc(ar, &d_out_a, N * (size_t)cols))) break; }
        int *k0 = nullptr, *k1 = nullptr, *k2 = nullptr; unsigned* ids = nullptr;
        if ((rc = aalloc(ar, &k0, N)) || (rc = aalloc(ar, &k1, N)) || (rc = aalloc(ar, &k2, N)) || (rc = aalloc(ar, &ids, N))) break;
        const int nb = (int)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((k_voxel_keys<T>), dim3(nb), dim3(kBlock), 0, s, d_pts, (int)n, vs[0], vs[1], vs[2], mn[0], mn[1], mn[2], k0, k1, k2, ids);
        unsigned *perm = nullptr, *flag = nullptr, *scan = nullptr;
        const unsigned long long* skeys = nullptr;
        if ((rc = sort_by_triple<int>(ar, s, k0, k1, k2, ids, (int)n, &perm, &skeys))) break;
        if ((rc = runs_of<int>(ar, s, k0, k1, k2, perm, (int)n, &flag, &scan, skeys))) break;
        unsigned *start = nullptr, *keep = nullptr, *keep_scan = nullptr;
        if ((rc = aalloc(ar, &start, N + 1)) || (rc = aalloc(ar, &keep, N)) || (rc = aalloc(ar, &keep_scan, N))) break;
        hipLaunchKernelGGL(k_run_starts, dim3(nb), dim3(kBlock), 0, s, flag, scan, (int)n, start);
        const unsigned* n_runs_dev = scan + (n - 1);
        HIP_TRY(hipMemsetAsync(keep, 0, N * 4, s));
        hipLaunchKernelGGL(k_run_keep, dim3(nb), dim3(kBlock), 0, s, start, n_runs_dev, min_pts, keep);
        if ((rc = own_inclusive_scan(ar, s, keep, keep_scan, N))) break;
        hipLaunchKernelGGL((k_voxel_means<T, A>), dim3(nb), dim3(kBlock), 0, s, d_pts, d_attr, cols, perm, start, n_runs_dev, keep, keep_scan, d_out_v, d_out_a);
        HIP_TRY(hipGetLastError());
        unsigned n_out = 0;
        HIP_TRY(hipMemcpyAsync(&n_out, keep_scan + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        *out_count = (int64_t)n_out;
        if (!on_dev && n_out) {
            HIP_TRY(hipMemcpyAsync(out_v, d_out_v, (size_t)n_out * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
            if (has_attr) HIP_TRY(hipMemcpyAsync(out_a, d_out_a, (size_t)n_out * cols * sizeof(A), hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
        }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ---------------------------------------------------------------------------------------------------- duplicate removal
// deduplicate_point_cloud (src/remove_duplicates.cpp:108-129). out_pts (n,3), out_svi (n) worst case; out_svj (n); *out_count = unique rows.
template <typename T>
static int dedup_impl(pcu_hip_ctx* c, const T* pts, int64_t n, double epsilon, T* out_pts, int32_t* out_svi, int32_t* out_svj, int64_t* out_count, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (n <= 0) { *out_count = 0; return 0; }
    if (n > 0x7ffffff0ll) return fail(PCU_HIP_ERR_INVALID, "point clouds with more than 2^31-16 rows are not supported");
    typedef typename EncT<T>::type K;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    const size_t N = (size_t)n;
    if (ctx_begin(c, 8 * align_up(N * sizeof(K), 256) + 8 * align_up(N * 4, 256) + (1 << 20) + (on_dev ? 0 : 3 * align_up(N * 3 * sizeof(T), 256)) + 65536)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T* d_pts;
        if ((rc = stage_any(ar, pts, N * 3, on_dev, s, &d_pts))) break;
        T* d_out = out_pts; int *d_svi = out_svi, *d_svj = out_svj;
        if (!on_dev) { if ((rc = aalloc(ar, &d_out, N * 3)) || (rc = aalloc(ar, &d_svi, N)) || (rc = aalloc(ar, &d_svj, N))) break; }
        K *k0 = nullptr, *k1 = nullptr, *k2 = nullptr; unsigned* ids = nullptr;
        if ((rc = aalloc(ar, &k0, N)) || (rc = aalloc(ar, &k1, N)) || (rc = aalloc(ar, &k2, N)) || (rc = aalloc(ar, &ids, N))) break;
        const int nb = (int)((n + kBlock - 1) / kBlock);
        hipLaunchKernelGGL((k_round_keys<T>), dim3(nb), dim3(kBlock), 0, s, d_pts, (int)n, (T)epsilon, k0, k1, k2, ids);
        unsigned *perm = nullptr, *flag = nullptr, *scan = nullptr;
        const unsigned long long* skeys = nullptr;
        if ((rc = sort_by_triple<K>(ar, s, k0, k1, k2, ids, (int)n, &perm, &skeys))) break;
        if ((rc = runs_of<K>(ar, s, k0, k1, k2, perm, (int)n, &flag, &scan, skeys))) break;
        hipLaunchKernelGGL((k_dedup_write<T>), dim3(nb), dim3(kBlock), 0, s, d_pts, perm, flag, scan, (int)n, d_out, d_svi, d_svj);
        HIP_TRY(hipGetLastError());
        unsigned n_out = 0;
        HIP_TRY(hipMemcpyAsync(&n_out, scan + (n - 1), 4, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        *out_count = (int64_t)n_out;
        if (!on_dev) {
            HIP_TRY(hipMemcpyAsync(out_pts, d_out, (size_t)n_out * 3 * sizeof(T), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_svi, d_svi, (size_t)n_out * 4, hipMemcpyDeviceToHost, s));
            HIP_TRY(hipMemcpyAsync(out_svj, d_svj, N * 4, hipMemcpyDeviceToHost, s));
            HIP_WAIT(s);
        }
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}

// ---------------------------------------------------------------------------------------------------- pairwise distances / Sinkhorn (8f-3)
template <typename T>
static int pairwise_impl(pcu_hip_ctx* c, const T* a, const T* b, int64_t nb, int64_t m, int64_t n, int64_t d, double p_norm, T* out, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (nb <= 0 || m <= 0 || n <= 0 || d <= 0) return 0;            // empty result
    if (nb > 65535) return fail(PCU_HIP_ERR_INVALID, "more than 65535 batches are not supported");
    if (m > 0x7fffffffll || n > 0x7fffffffll || d > 0x7fffffffll) return fail(PCU_HIP_ERR_INVALID, "dimension too large");
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    const size_t na = (size_t)nb * m * d, nbb = (size_t)nb * n * d, no = (size_t)nb * m * n;
    if (ctx_begin(c, on_dev ? 4096 : align_up(na * sizeof(T), 256) + align_up(nbb * sizeof(T), 256) + align_up(no * sizeof(T), 256) + 4096)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T *da, *db;
        if ((rc = stage_any(ar, a, na, on_dev, s, &da)) || (rc = stage_any(ar, b, nbb, on_dev, s, &db))) break;
        T* dout = out;
        if (!on_dev && (rc = aalloc(ar, &dout, no))) break;
        const int pc = std::isnan(p_norm) ? P_TWO : pcode_of(p_norm);                 // ord=None: the 2-norm
        const long long pw_cols = (n + 63) / 64, pw_blocks = pw_cols * ((m + 3) / 4);
        if (pw_blocks > 0x7fffffffll || nb > 65535) { rc = fail(PCU_HIP_ERR_INVALID, "pairwise_distances: problem too large (more than 2^31-1 tiles of 4 x 64 entries per batch, or more than 65535 batches)"); break; }
        hipLaunchKernelGGL((k_pairwise<T>), dim3((unsigned)pw_blocks, (unsigned)nb), dim3(256), 0, s, da, db, (int)m, (int)n, (int)d, pc, p_norm, dout, (int)pw_cols);
        HIP_TRY(hipGetLastError());
        if (!on_dev) HIP_TRY(hipMemcpyAsync(out, dout, no * sizeof(T), hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename T>
static int sinkhorn_impl(pcu_hip_ctx* c, const T* a, const T* b, const T* M, int64_t nb, int64_t m, int64_t n, double eps, int max_iters, double stop_thresh,
                         T* out_P, int* out_iters, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    if (nb <= 0 || m <= 0 || n <= 0) { if (out_iters) *out_iters = 0; return 0; }
    if (nb > 65535 || m > 0x7fffffffll || n > 0x7fffffffll) return fail(PCU_HIP_ERR_INVALID, "problem too large: at most 65535 batches, 2^31-1 samples per measure");
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    const size_t nM = (size_t)nb * m * n, nu = (size_t)nb * m, nv = (size_t)nb * n;
    size_t need = 2 * (align_up(nu * sizeof(T), 256) + align_up(nv * sizeof(T), 256)) + 2 * align_up(((size_t)m / 4 + 64) * nv * sizeof(T), 256) + 8192;
    if (!on_dev) need += 2 * align_up(nM * sizeof(T), 256) + align_up(nu * sizeof(T), 256) + align_up(nv * sizeof(T), 256);
    if (ctx_begin(c, need)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        SinkArgs<T> k;
        if ((rc = stage_any(ar, a, nu, on_dev, s, &k.a)) || (rc = stage_any(ar, b, nv, on_dev, s, &k.b)) || (rc = stage_any(ar, M, nM, on_dev, s, &k.M))) break;
        T* dP = out_P;
        if (!on_dev && (rc = aalloc(ar, &dP, nM))) break;
        int* flagsd = nullptr;
        if ((rc = aalloc(ar, &k.u, nu)) || (rc = aalloc(ar, &k.v, nv)) || (rc = aalloc(ar, &k.du, nu)) || (rc = aalloc(ar, &k.dv, nv)) || (rc = aalloc(ar, &flagsd, 16))) break;
        HIP_TRY(hipMemsetAsync(k.u, 0, nu * sizeof(T), s)); HIP_TRY(hipMemsetAsync(k.v, 0, nv * sizeof(T), s));      // u = zeros_like(a), v = zeros_like(b) (:99-100)
        HIP_TRY(hipMemsetAsync(flagsd, 0, 16 * sizeof(int), s));
        k.nb = (int)nb; k.m = (int)m; k.n = (int)n; k.eps = (T)eps; k.done = flagsd;
        // n <= 4096: one launch per iteration reads M once for both updates (k_sink_iter); wider problems: row pass + column pass over
        // slabs of rows (so that the launch has a few hundred blocks even for one small batch). PCU_HIP_SINK_TWO_PASS=1: always the latter.
        static const bool two_pass = getenv("PCU_HIP_SINK_TWO_PASS") != nullptr;
        const int cpt = n <= 1024 ? 4 : (n <= 4096 ? 16 : 0);
        const bool fused = cpt > 0 && !two_pass;
        int n_slabs, rows_per_slab;
        if (fused) {
            // at least 8 rows per block (measured at 4096 x 4096 x 50 iterations: 4 rows 4.20 ms, 8 rows 3.97, 16 rows 4.89), ~512 blocks or more
            rows_per_slab = (int)std::max<long long>(8, ((long long)m * nb + 1023) / 1024);
            n_slabs = (int)((m + rows_per_slab - 1) / rows_per_slab);
        } else {
            const long long col_blocks = (long long)((n + 31) / 32) * nb;
            n_slabs = (int)std::max<long long>(1, std::min<long long>(32, (1024 + col_blocks - 1) / col_blocks));
            n_slabs = (int)std::min<long long>(n_slabs, (m + 255) / 256);
            rows_per_slab = (int)((m + n_slabs - 1) / n_slabs);
        }
        T *part_mx = nullptr, *part_sum = nullptr;
        if ((rc = aalloc(ar, &part_mx, (size_t)n_slabs * nv)) || (rc = aalloc(ar, &part_sum, (size_t)n_slabs * nv))) break;
        int host_flags[2] = {0, 0};
        for (int it = 0; it < max_iters; ++it) {
            if (fused) {
                if (cpt == 4) hipLaunchKernelGGL((k_sink_iter<T, 4>), dim3((unsigned)n_slabs, (unsigned)nb), dim3(256), 0, s, k, part_mx, part_sum, rows_per_slab);
                else hipLaunchKernelGGL((k_sink_iter<T, 16>), dim3((unsigned)n_slabs, (unsigned)nb), dim3(256), 0, s, k, part_mx, part_sum, rows_per_slab);
            } else {
                hipLaunchKernelGGL((k_sink_rows<T>), dim3((unsigned)m, (unsigned)nb), dim3(256), 0, s, k);
                hipLaunchKernelGGL((k_sink_cols<T>), dim3((unsigned)((n + 31) / 32), (unsigned)nb, (unsigned)n_slabs), dim3(1024), 0, s, k, part_mx, part_sum, rows_per_slab);
            }
            if (fused) hipLaunchKernelGGL((k_sink_cols_merge<T>), dim3((unsigned)((n + 31) / 32), (unsigned)nb), dim3(1024), 0, s, k, part_mx, part_sum, n_slabs);
            else hipLaunchKernelGGL((k_sink_cols_finish<T>), dim3((unsigned)((n + 255) / 256), (unsigned)nb), dim3(256), 0, s, k, part_mx, part_sum, n_slabs);
            hipLaunchKernelGGL((k_sink_check<T>), dim3(1), dim3(1024), 0, s, k, (T)stop_thresh, flagsd + 1);
            if ((it & 7) == 7) {             // every 8 iterations: has the stopping rule fired? (later launches are no-ops once it has)
                HIP_TRY(hipMemcpyAsync(host_flags, flagsd, sizeof host_flags, hipMemcpyDeviceToHost, s));
                HIP_WAIT(s);
                if (host_flags[0]) break;
            }
        }
        HIP_TRY(hipGetLastError());
        const long long pl_cols = (n + 255) / 256;
        if (pl_cols * m > 0x7fffffffll) { rc = fail(PCU_HIP_ERR_INVALID, "sinkhorn: cost matrix too large (more than 2^31-1 row segments of 256 entries per batch)"); break; }
        hipLaunchKernelGGL((k_sink_plan<T>), dim3((unsigned)(pl_cols * m), (unsigned)nb), dim3(256), 0, s, k, dP, (int)pl_cols);
        HIP_TRY(hipGetLastError());
        HIP_TRY(hipMemcpyAsync(host_flags, flagsd, sizeof host_flags, hipMemcpyDeviceToHost, s));
        if (!on_dev) HIP_TRY(hipMemcpyAsync(out_P, dP, nM * sizeof(T), hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        if (out_iters) *out_iters = host_flags[1];
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
template <typename T>
static int dot_impl(pcu_hip_ctx* c, const T* x, const T* y, int64_t count, double* out, unsigned flags, void* stream) {
    if (!c) return fail(PCU_HIP_ERR_INVALID, "null context");
    *out = 0.0;
    if (count <= 0) return 0;
    const bool on_dev = flags & PCU_HIP_PTRS_ON_DEVICE;
    hipStream_t s = pick_stream(c, flags, stream);
    constexpr int kParts = 1024;
    if (ctx_begin(c, (on_dev ? 0 : 2 * align_up((size_t)count * sizeof(T), 256)) + 65536)) return PCU_HIP_ERR_RUNTIME;
    Arena ar{c};
    int rc = 0;
    do {
        const T *dx, *dy; double* part = nullptr;
        if ((rc = stage_any(ar, x, (size_t)count, on_dev, s, &dx)) || (rc = stage_any(ar, y, (size_t)count, on_dev, s, &dy)) || (rc = aalloc(ar, &part, kParts))) break;
        hipLaunchKernelGGL((k_dot_partial<T>), dim3(kParts), dim3(256), 0, s, dx, dy, (size_t)count, part);
        HIP_TRY(hipGetLastError());
        double h[kParts];
        HIP_TRY(hipMemcpyAsync(h, part, sizeof h, hipMemcpyDeviceToHost, s));
        HIP_WAIT(s);
        double r = 0; for (int i = 0; i < kParts; ++i) r += h[i];
        *out = r;
    } while (0);
    ctx_end(c);
    return rc ? (rc < 0 ? rc : PCU_HIP_ERR_RUNTIME) : 0;
}
