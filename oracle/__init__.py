"""oracle -- CPU checkers for the KNN / Chamfer / Hausdorff hot path. TEST INFRASTRUCTURE ONLY.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this
package. The product (``point_cloud_utils_amd``) never imports it and has no CPU fallback.

Three checkers, strongest first:

* ``ref``    -- ``oracle/_ref/libpcu_ref.so``: the reference's *own* vendored nanoflann.hpp
               (``/root/reference/external/nanoflann/nanoflann.hpp``) compiled where it lies with the
               reference's flags, behind a driver that restates ``src/point_cloud_distance.cpp:21-99,
               211-225`` (``oracle/ref_shim.cpp``). Built only where ``/root/reference`` exists; the
               built ``.so`` travels to the GPU box with the repo snapshot.
* ``port``   -- ``oracle/libpcu_oracle.so``: a plain-C restatement of the same algorithm
               (``oracle/kdtree_oracle.c`` + ``kdtree_body.inc``), pinned bit-exactly against ``ref`` and
               against ``tests/golden/*.npz`` (generated from ``ref`` by ``tests/golden/make_golden.py``).
* ``brute``  -- exact-arithmetic brute force ordered by (d2, index); equals the kd-tree result whenever
               no exact distance tie touches the top-k (``SURVEY.md`` section 8c).

The Python tails (``hausdorff_distance``, ``chamfer_distance``) restate
``point_cloud_utils/__init__.py:52-120`` of the reference on top of whichever checker is selected.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_c_i64 = ctypes.c_int64
_c_int = ctypes.c_int


def build(force=False):
    """Compile the checkers (gcc / g++ only). `_ref` is rebuilt only where /root/reference exists."""
    port = os.path.join(_HERE, "libpcu_oracle.so")
    if force or not os.path.exists(port) or os.path.exists("/root/reference/external/nanoflann/nanoflann.hpp") \
            and not os.path.exists(os.path.join(_HERE, "_ref", "libpcu_ref.so")):
        subprocess.run(["make", "-C", _HERE, "all"], check=True, stdout=subprocess.DEVNULL)


def _load(path):
    return ctypes.CDLL(path) if os.path.exists(path) else None


_libs = {}


def lib(kind):
    if kind not in _libs:
        if kind == "ref":
            _libs[kind] = _load(os.path.join(_HERE, "_ref", "libpcu_ref.so"))
        else:
            p = os.path.join(_HERE, "libpcu_oracle.so")
            if not os.path.exists(p):
                build()
            _libs[kind] = _load(p)
    return _libs[kind]


def have_ref():
    return lib("ref") is not None


def _prep(a):
    a = np.ascontiguousarray(a)
    if a.dtype not in (np.float32, np.float64):
        raise ValueError("float32/float64 only")
    return a


def _suffix(a):
    return "f32" if a.dtype == np.float32 else "f64"


def _ptr(a):
    return a.ctypes.data_as(ctypes.c_void_p)


def knn(query, dataset, k, squared_distances=False, max_points_per_leaf=10, num_threads=-1, kind="port",
        timing=None):
    """(dists (n,k), corrs (n,k) int64) exactly as shortest_distances_nanoflann fills them (no squeeze)."""
    q, r = _prep(query), _prep(dataset)
    assert q.dtype == r.dtype and q.ndim == 2 and r.ndim == 2 and q.shape[1] == 3 and r.shape[1] == 3
    n = q.shape[0]
    d = np.empty((n, k), dtype=q.dtype)
    c = np.empty((n, k), dtype=np.int64)
    L = lib(kind if kind != "brute" else "port")
    if L is None:
        raise RuntimeError(f"oracle library for kind={kind!r} is not built")
    if kind == "ref":
        t = (ctypes.c_double * 2)()
        rc = getattr(L, "pcu_ref_knn_" + _suffix(q))(_ptr(q), _c_i64(n), _ptr(r), _c_i64(r.shape[0]), _c_int(k),
                                                   _c_int(int(squared_distances)), _c_int(max_points_per_leaf),
                                                   _c_int(num_threads), _ptr(d), _ptr(c), t)
        if timing is not None:
            timing["build_s"], timing["search_s"] = t[0], t[1]
    elif kind == "port":
        rc = getattr(L, "pcu_oracle_knn_" + _suffix(q))(_ptr(q), _c_i64(n), _ptr(r), _c_i64(r.shape[0]), _c_int(k),
                                                      _c_int(int(squared_distances)), _c_int(max_points_per_leaf),
                                                      _ptr(d), _ptr(c))
    elif kind == "brute":
        rc = getattr(L, "pcu_oracle_brute_knn_" + _suffix(q))(_ptr(q), _c_i64(n), _ptr(r), _c_i64(r.shape[0]),
                                                            _c_int(k), _c_int(int(squared_distances)),
                                                            _ptr(d), _ptr(c), None)
    else:
        raise ValueError(kind)
    if rc != 0:
        raise RuntimeError("oracle failed")
    return d, c


def brute_knn_with_ties(query, dataset, k, squared_distances=False):
    """brute-force result + per-query flag: an exact d2 tie inside the top-(k+1)."""
    q, r = _prep(query), _prep(dataset)
    n = q.shape[0]
    d = np.empty((n, k), dtype=q.dtype)
    c = np.empty((n, k), dtype=np.int64)
    tie = np.zeros(n, dtype=np.uint8)
    rc = getattr(lib("port"), "pcu_oracle_brute_knn_" + _suffix(q))(_ptr(q), _c_i64(n), _ptr(r), _c_i64(r.shape[0]),
                                                                  _c_int(k), _c_int(int(squared_distances)),
                                                                  _ptr(d), _ptr(c), _ptr(tie))
    assert rc == 0
    return d, c, tie.astype(bool)


def tree_dump(dataset, max_points_per_leaf=10):
    """The restatement's kd-tree: (vAcc, nodes_i[n,3]=(child1,child2,divfeat), nodes_f[n,2], nodes_lr[n,2])."""
    r = _prep(dataset)
    m = r.shape[0]
    cap = 2 * m + 16
    vacc = np.empty(m, np.int64)
    ni = np.empty((cap, 3), np.int32)
    nf = np.empty((cap, 2), r.dtype)
    nlr = np.empty((cap, 2), np.int64)
    fn = getattr(lib("port"), "pcu_oracle_tree_dump_" + _suffix(r))
    fn.restype = ctypes.c_int64
    nn = fn(_ptr(r), _c_i64(m), _c_int(max_points_per_leaf), _ptr(vacc), _ptr(ni), _ptr(nf), _ptr(nlr), _c_i64(cap))
    return vacc, ni[:nn], nf[:nn], nlr[:nn]


# ---- reference API surface on top of a checker (binding + Python tails) -------------------------------

def k_nearest_neighbors(query_points, dataset_points, k, squared_distances=False, max_points_per_leaf=10,
                        num_threads=-1, kind="port"):
    """src/point_cloud_distance.cpp:123-164 incl. the k==1 squeeze (tests/test_examples.py:363-368)."""
    if k <= 0:
        raise ValueError(f"Invalid value for k ({k}) must be greater than 0.")
    d, c = knn(query_points, dataset_points, k, squared_distances, max_points_per_leaf, num_threads, kind)
    if k == 1:
        return d[:, 0], c[:, 0]
    return d, c


def one_sided_hausdorff_distance(source, target, return_index=True, squared_distances=False,
                                 max_points_per_leaf=10, kind="port"):
    """src/point_cloud_distance.cpp:186-234."""
    d, c = knn(source, target, 1, squared_distances, max_points_per_leaf, 0, kind)
    i = int(np.argmax(d[:, 0]))          # np.argmax: first maximum, like Eigen's strict '>' visitor
    if return_index:
        return float(d[i, 0]), i, int(c[i, 0])
    return float(d[i, 0])


def hausdorff_distance(x, y, return_index=False, squared_distances=False, max_points_per_leaf=10, kind="port"):
    """point_cloud_utils/__init__.py:52-81."""
    hxy, ix1, iy1 = one_sided_hausdorff_distance(x, y, True, squared_distances, max_points_per_leaf, kind)
    hyx, iy2, ix2 = one_sided_hausdorff_distance(y, x, True, squared_distances, max_points_per_leaf, kind)
    h = max(hxy, hyx)
    if return_index and hxy > hyx:
        return h, ix1, iy1
    elif return_index and hxy <= hyx:
        return h, ix2, iy2
    return h


def chamfer_distance(x, y, return_index=False, p_norm=2, max_points_per_leaf=10, kind="port"):
    """point_cloud_utils/__init__.py:84-120."""
    x = np.asarray(x)
    y = np.asarray(y)
    _, cxy = k_nearest_neighbors(x, y, 1, False, max_points_per_leaf, kind=kind)
    _, cyx = k_nearest_neighbors(y, x, 1, False, max_points_per_leaf, kind=kind)
    dxy = np.linalg.norm(x[cyx] - y, axis=-1, ord=p_norm).mean()
    dyx = np.linalg.norm(y[cxy] - x, axis=-1, ord=p_norm).mean()
    cham = np.mean(dxy) + np.mean(dyx)
    if return_index:
        return cham, cxy, cyx
    return cham


# ---- normals: numpy restatement of src/point_cloud_normals.cpp:48-173 on top of a KNN checker ----------

def _orient_filter(normals, dirs, drop_angle_threshold):
    """:161-169: normal *= sign(normal . dir); drop if acos(normal . dir) > threshold. Returns (normals, keep)."""
    d = np.einsum("ij,ij->i", normals, dirs)
    normals = normals * np.sign(d)[:, None]
    ang = np.arccos(np.clip(np.einsum("ij,ij->i", normals, dirs), -1.0, 1.0))
    return normals, ~(ang > drop_angle_threshold)


def normals_knn(points, num_neighbors, view_directions=None, drop_angle_threshold=np.pi / 2, max_points_per_leaf=10, kind="port"):
    """estimate_local_normal_knn for every point (:115-173): neighbour offsets in the input dtype, widened to double, thin SVD,
    V[:, 2]. Returns (idx, normals (float64, sign as numpy's SVD gives it), gap) where gap = (s1 - s2) / s0 of the kept points'
    singular values (how well the smallest direction is separated: tests skip ill-conditioned fits)."""
    p = _prep(points)
    n = p.shape[0]
    _, c = knn(p, p, num_neighbors, True, max_points_per_leaf, kind=kind)
    ok = c[:, -1] >= 0
    a = (p[np.where(c >= 0, c, 0)] - p[:, None, :]).astype(np.float64)          # (n, k, 3), differences taken in the input dtype
    _, s, vt = np.linalg.svd(a, full_matrices=False)
    normals = vt[:, 2, :] if vt.shape[1] >= 3 else np.zeros((n, 3))
    gap = (s[:, 1] - s[:, 2]) / np.maximum(s[:, 0], 1e-300) if s.shape[1] >= 3 else np.zeros(n)
    if view_directions is not None and len(view_directions):
        normals, keep = _orient_filter(normals, np.asarray(view_directions, dtype=np.float64), drop_angle_threshold)
        ok &= keep
    idx = np.flatnonzero(ok)
    return idx, normals[idx], gap[idx]


def normals_ball(points, ball_radius, view_directions=None, drop_angle_threshold=np.pi / 2, min_pts_per_ball=3, weight_function="constant"):
    """estimate_local_normal_rbf for every point (:48-113), brute force (small clouds only). Members: d2 < T(ball_radius) with
    d2 = ((dx*dx)+(dy*dy))+(dz*dz) in the input dtype -- nanoflann's RadiusResultSet compares SQUARED distances (:278-283)."""
    p = _prep(points)
    n = p.shape[0]
    rad = p.dtype.type(ball_radius)
    normals = np.zeros((n, 3)); gap = np.zeros(n); ok = np.zeros(n, bool)
    for i in range(n):
        d = p[i] - p
        d2 = (d[:, 0] * d[:, 0] + d[:, 1] * d[:, 1]) + d[:, 2] * d[:, 2]
        m = np.flatnonzero(d2 < rad)
        if len(m) < min_pts_per_ball:
            continue
        m = m[np.argsort(d2[m], kind="stable")]
        w = np.ones(len(m))
        if weight_function == "rbf":
            r = np.sqrt(d2[m].astype(np.float64)) / ball_radius
            w = (1.0 - r) ** 4 * (4 * r + 1.0)
        a = (p[m] - p[i]).astype(np.float64) * w[:, None]
        _, s, vt = np.linalg.svd(a, full_matrices=False)
        normals[i] = vt[2]; gap[i] = (s[1] - s[2]) / max(s[0], 1e-300); ok[i] = True
    if view_directions is not None and len(view_directions):
        normals, keep = _orient_filter(normals, np.asarray(view_directions, dtype=np.float64), drop_angle_threshold)
        ok &= keep
    idx = np.flatnonzero(ok)
    return idx, normals[idx], gap[idx]


# ---- Morton codes, voxel-grid downsampling, duplicate removal ------------------------------------------------

def _morton_lib():
    """oracle/_ref/libpcu_ref_morton.so: the reference's own src/common/morton_code.cpp behind oracle/ref_morton_shim.cpp."""
    if "morton" not in _libs:
        _libs["morton"] = _load(os.path.join(_HERE, "_ref", "libpcu_ref_morton.so"))
    return _libs["morton"]


def have_ref_morton():
    return _morton_lib() is not None


_M_SIGN = np.uint64(0x7000000000000000)
_M_X = np.uint64(0x1249249249249249)


def _split21(r):
    r = r.astype(np.uint64)
    for sh, mask in ((32, 0x1f00000000ffff), (16, 0x1f0000ff0000ff), (8, 0x100f00f00f00f00f), (4, 0x10c30c30c30c30c3), (2, 0x1249249249249249)):
        r = (r | (r << np.uint64(sh))) & np.uint64(mask)
    return r


def _compact21(x):
    d = x & np.uint64(0x1249249249249249)
    for sh, mask in ((2, 0x10c30c30c30c30c3), (4, 0x100f00f00f00f00f), (8, 0x1f0000ff0000ff), (16, 0x1f00000000ffff)):
        d = (d | (d >> np.uint64(sh))) & np.uint64(mask)
    d = d | (d >> np.uint64(32))
    d = np.where(d & np.uint64(0x100000), d | np.uint64(0xffe00000), d)
    return d.astype(np.uint32).view(np.int32)


def morton_encode(pts, kind="port"):
    """src/morton.cpp:185-243 on MortonCode64(int32, int32, int32) (src/common/morton_code.cpp:46-66). kind "ref": the reference's
    own class compiled in place; "port": numpy restatement."""
    p = np.ascontiguousarray(np.asarray(pts).astype(np.int32))
    if kind == "ref":
        out = np.empty(p.shape[0], np.uint64)
        _morton_lib().pcu_ref_morton_encode(_ptr(p), _c_i64(p.shape[0]), _ptr(out))
        return out
    u = p.view(np.uint32)
    u = ((u & np.uint32(0x80000000)) >> np.uint32(11)) | (u & np.uint32(0x0fffff))
    return (_split21(u[:, 0]) | (_split21(u[:, 1]) << np.uint64(1)) | (_split21(u[:, 2]) << np.uint64(2))) ^ _M_SIGN


def morton_decode(codes, kind="port"):
    c = np.ascontiguousarray(np.asarray(codes).astype(np.uint64))
    if kind == "ref":
        out = np.empty((c.shape[0], 3), np.int32)
        _morton_lib().pcu_ref_morton_decode(_ptr(c), _c_i64(c.shape[0]), _ptr(out))
        return out
    d = c ^ _M_SIGN
    return np.stack([_compact21(d), _compact21(d >> np.uint64(1)), _compact21(d >> np.uint64(2))], axis=1)


def _morton_negate(data):
    ym, zm = _M_X << np.uint64(1), _M_X << np.uint64(2)
    d = ~data
    one = np.uint64(1)
    return (((d | ~_M_X) + one) & _M_X) | (((d | ~ym) + one) & ym) | (((d | ~zm) + one) & zm)


def morton_addsub(c1, c2, subtract=False, kind="port"):
    a = np.ascontiguousarray(np.asarray(c1).astype(np.uint64)); b = np.ascontiguousarray(np.asarray(c2).astype(np.uint64))
    if kind == "ref":
        out = np.empty(a.shape[0], np.uint64)
        _morton_lib().pcu_ref_morton_addsub(_ptr(a), _ptr(b), _c_i64(a.shape[0]), _c_int(int(subtract)), _ptr(out))
        return out
    if subtract:
        b = _morton_negate(b)
    x1, x2 = a ^ _M_SIGN, b ^ _M_SIGN
    ym, zm = _M_X << np.uint64(1), _M_X << np.uint64(2)
    with np.errstate(over="ignore"):
        xs = (x1 | ~_M_X) + (x2 & _M_X); ys = (x1 | ~ym) + (x2 & ym); zs = (x1 | ~zm) + (x2 & zm)
    return ((xs & _M_X) | (ys & ym) | (zs & zm)) ^ _M_SIGN


def morton_knn_window(codes, qcodes, k, kind="port"):
    """The window of src/morton.cpp:362-383 (sort_dist=False semantics)."""
    c = np.ascontiguousarray(np.asarray(codes).astype(np.uint64)); q = np.ascontiguousarray(np.asarray(qcodes).astype(np.uint64))
    n = c.shape[0]; k = min(int(k), n)
    if kind == "ref":
        out = np.empty((q.shape[0], k), np.int64)
        _morton_lib().pcu_ref_morton_knn_window(_ptr(c), _c_i64(n), _ptr(q), _c_i64(q.shape[0]), _c_int(k), _ptr(out))
        return out
    idx = np.searchsorted(c, q, side="left").astype(np.int64)
    up, down = k // 2, k - k // 2
    upper, lower = idx + up, idx - down
    over = upper >= n
    lower = np.where(over, lower - (upper - n), lower); upper = np.where(over, n, upper)
    neg = lower < 0
    upper = np.where(neg, upper - lower, upper); lower = np.where(neg, 0, lower)
    return lower[:, None] + np.arange(k, dtype=np.int64)[None, :]


def voxel_downsample(points, attrib, voxel_size, min_bound, min_points_per_voxel=1):
    """downsample_point_cloud_to_voxels (src/sample_point_cloud.cpp:163-235) restated with a dict: sums in input order in the
    input dtypes, mean = sum / count. Returns (v, attrib or None) sorted by voxel index (the reference's order is its hash table's)."""
    p = np.ascontiguousarray(points); T = p.dtype.type
    vs = np.asarray(voxel_size, dtype=p.dtype); mb = np.asarray(min_bound, dtype=p.dtype)
    key = np.floor((p - mb) / vs).astype(np.int32)
    acc = {}
    for i in range(p.shape[0]):
        k = (int(key[i, 0]), int(key[i, 1]), int(key[i, 2]))
        e = acc.get(k)
        if e is None:
            e = acc[k] = [np.zeros(3, p.dtype), None if attrib is None else np.zeros(attrib.shape[1], attrib.dtype), 0]
        e[0] = e[0] + p[i]
        if attrib is not None:
            e[1] = e[1] + attrib[i]
        e[2] += 1
    keys = sorted(k for k, e in acc.items() if e[2] >= min_points_per_voxel)
    v = np.array([acc[k][0] / T(acc[k][2]) for k in keys], dtype=p.dtype).reshape(-1, 3)
    a = None if attrib is None else np.array([acc[k][1] / attrib.dtype.type(acc[k][2]) for k in keys], dtype=attrib.dtype).reshape(-1, attrib.shape[1])
    return v, a


def deduplicate_point_cloud(points, epsilon):
    """remove_duplicate_vertices (src/remove_duplicates.cpp:11-36): rows equal after round(V / eps) are one; unique rows in
    lexicographic order (libigl unique_rows); representative = lowest input row (libigl's choice is not in the checkout)."""
    p = np.ascontiguousarray(points)
    r = p
    if epsilon > 0:                       # igl::round = std::round: half away from zero (np.round is half-to-even)
        q = p / p.dtype.type(epsilon)
        t = np.trunc(q)
        r = t + np.where(np.abs(q - t) >= 0.5, np.sign(q), 0).astype(p.dtype)      # q - trunc(q) is exact
    r = r + p.dtype.type(0)
    order = np.lexsort((np.arange(len(p)), r[:, 2], r[:, 1], r[:, 0]))
    rs = r[order]
    head = np.ones(len(p), bool); head[1:] = np.any(rs[1:] != rs[:-1], axis=1)
    run = np.cumsum(head) - 1
    svi = order[head].astype(np.int32)
    svj = np.empty(len(p), np.int32); svj[order] = run
    return p[svi], svi, svj


# ---- numpy restatement of point_cloud_utils/_sinkhorn.py (the reference module is pure numpy; it is imported
# from /root/reference where that exists to pin this restatement and to generate tests/golden/sinkhorn.npz) -------------------

def pairwise_distances(a, b, p=None):
    """_sinkhorn.py:4-33."""
    squeezed = False
    if len(a.shape) == 2 and len(b.shape) == 2:
        a = a[np.newaxis]; b = b[np.newaxis]; squeezed = True
    ret = np.linalg.norm(a[:, :, np.newaxis, :] - b[:, np.newaxis, :, :], axis=-1, ord=p)
    return np.squeeze(ret) if squeezed else ret


def sinkhorn(a, b, M, eps, max_iters=100, stop_thresh=1e-3):
    """_sinkhorn.py:36-130 (argument checks omitted). Returns (P, iterations run)."""
    M = np.squeeze(M); a = np.squeeze(a); b = np.squeeze(b)
    squeezed = False
    if M.ndim == 2:
        M = M[np.newaxis]; a = a[np.newaxis]; b = b[np.newaxis]; squeezed = True
    u = np.zeros_like(a); v = np.zeros_like(b)
    Mt = np.transpose(M, axes=(0, 2, 1))

    def lse(x):
        mx = x.max(2)
        return np.log(np.sum(np.exp(x - mx[:, :, np.newaxis]), axis=2)) + mx

    iters = 0
    for _ in range(max_iters):
        up, vp = u, v
        u = eps * (np.log(a) - lse((-M + np.expand_dims(v, 1)) / eps))
        v = eps * (np.log(b) - lse((-Mt + np.expand_dims(u, 1)) / eps))
        iters += 1
        if np.sum(np.abs(up - u), axis=1).max() < stop_thresh and np.sum(np.abs(vp - v), axis=1).max() < stop_thresh:
            break
    P = np.exp((-M + np.expand_dims(u, 2) + np.expand_dims(v, 1)) / eps)
    return (np.squeeze(P) if squeezed else P), iters


def earth_movers_distance(p, q, p_norm=2, eps=1e-4, max_iters=100, stop_thresh=1e-3):
    """_sinkhorn.py:133-156."""
    M = pairwise_distances(p, q, p_norm)
    a = np.ones(p.shape[0]) / p.shape[0]; b = np.ones(q.shape[0]) / q.shape[0]
    if a.dtype != M.dtype:
        raise ValueError("Tensors a, b, and M must have the same dtype")
    P, _ = sinkhorn(a, b, M, eps, max_iters, stop_thresh)
    return (P * M).sum(), P


def reference_sinkhorn_module():
    """The reference's own _sinkhorn.py, loaded from /root/reference (None where it does not exist, e.g. on the GPU box)."""
    path = "/root/reference/point_cloud_utils/_sinkhorn.py"
    if not os.path.exists(path):
        return None
    import importlib.util
    spec = importlib.util.spec_from_file_location("_pcu_ref_sinkhorn", path)
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod
