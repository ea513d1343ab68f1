"""point_cloud_utils_amd -- MI355X (gfx950) drop-in for point-cloud-utils' KNN / Chamfer / Hausdorff hot path.

The four callables below have the signatures, defaults, dtypes, shapes and error behaviour of the reference's
``pcu.k_nearest_neighbors`` / ``pcu.one_sided_hausdorff_distance`` (``src/point_cloud_distance.cpp:123-164``,
``:186-234``) and ``pcu.hausdorff_distance`` / ``pcu.chamfer_distance``
(``point_cloud_utils/__init__.py:52-81``, ``:84-120``), so

    import point_cloud_utils_amd as pcu

is a drop-in for that path. All computation happens in hand-written HIP kernels behind the C ABI of
``libpcu_hip.so`` (``include/pcu_hip.h``); there is no CPU fallback: without the library or without a GPU the
calls raise.

Inputs may be numpy arrays (host; copied to the GPU for the call, results returned as numpy / Python scalars,
exactly as the reference returns them) or, as an extension, CUDA/HIP ``torch`` tensors (device-resident: nothing
crosses PCIe, array results are returned as ``torch`` tensors on the same device).

``num_threads`` is an OpenMP knob of the reference: accepted, ignored. ``max_points_per_leaf`` is the reference's
kd-tree leaf size: it cannot change distances, but the order of *exactly tied* neighbours in the reference is the
order its kd-tree traversal meets them, so it is forwarded and that order is reproduced on the GPU.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import Stats, set_cell_occupancy, device_count  # noqa: F401

__all__ = ["k_nearest_neighbors", "one_sided_hausdorff_distance", "hausdorff_distance", "chamfer_distance",
           "estimate_point_cloud_normals_knn", "estimate_point_cloud_normals_ball",
           "morton_encode", "morton_decode", "morton_add", "morton_subtract", "morton_knn",
           "downsample_point_cloud_on_voxel_grid", "deduplicate_point_cloud",
           "pairwise_distances", "sinkhorn", "earth_movers_distance",
           "last_stats", "set_timing", "set_cell_occupancy", "device_count", "DatasetIndex", "cancel"]

_last_stats = [None]      # the Stats struct of the most recent call (turned into a dict on demand)
# PCU_HIP_NO_TIE_ORDER=1: skip the kd-tree tie-order resolver (exact ties then ordered by (d2, row)); for experiments.
_ENV_FLAGS = _lib.NO_TIE_ORDER if __import__("os").environ.get("PCU_HIP_NO_TIE_ORDER", "0") not in ("", "0") else 0
_TIMING = [int(__import__("os").environ.get("PCU_HIP_TIMING", "0") or 0)]


def cancel():
    """Ask the call that is in flight (any thread of this process) to stop: it returns early and raises KeyboardInterrupt in its caller, as
    Ctrl-C does -- the reference polls PyErr_CheckSignals() inside its search loops (src/point_cloud_distance.cpp:60-75, 96-98). Callable from
    another thread or a signal handler; a request made while no call is running is dropped when the next one starts."""
    _lib.lib().pcu_hip_cancel()


def set_timing(level):
    """Device-side timing written into last_stats(): 0 = none (default; counters only), 1 = the main search launches
    (ms_kernel_search / n_kernel_search), 2 = also the phases of the call (ms_index / ms_search / ms_total / ms_tie).
    Every HIP event costs a few microseconds of bubble between kernels, which is why it is opt-in. Returns the old level."""
    old = _TIMING[0]
    _TIMING[0] = int(level)
    return old


def _flags():
    return _ENV_FLAGS | (_lib.TIME_KERNELS if _TIMING[0] >= 1 else 0) | (_lib.TIME_PHASES if _TIMING[0] >= 2 else 0)


def last_stats():
    """Statistics of the most recent call (escalations, tie handling, device milliseconds)."""
    st = _last_stats[0]
    return st.as_dict() if st is not None else {}


def _is_torch(a):
    return type(a).__module__.startswith("torch") and hasattr(a, "data_ptr")


def _shape2(a):
    sh = tuple(a.shape)
    if len(sh) == 2:
        return sh
    if len(sh) == 1:            # numpyeigen maps a 1-D array to a column vector
        return (sh[0], 1)
    raise ValueError(f"Invalid number of dimensions ({len(sh)}): expected a matrix of shape (n, 3).")


_DT_FAST = {np.dtype("float32"): "float32", np.dtype("float64"): "float64"}


def _dtype_name(a):
    dt = getattr(a, "dtype", None)
    try:
        name = _DT_FAST.get(dt)            # the two supported dtypes, numpy or torch, without string work
    except TypeError:
        name = None
    if name is not None:
        return name
    if _is_torch(a):
        name = str(a.dtype).replace("torch.", "")
        if name in ("float32", "float64"):
            _DT_FAST[a.dtype] = name
        return name
    return np.asarray(a).dtype.name


def _check_pair(a, b, aname, bname, zero_fmt, dim_fmt):
    """dtype / shape validation in the order of src/point_cloud_distance.cpp:136-149 (and :195-208)."""
    da, db = _dtype_name(a), _dtype_name(b)
    if da not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({da}) for argument '{aname}'. Expected one of ['float32', 'float64'].")
    if db != da:
        raise ValueError(f"Invalid scalar type ({db}) for argument '{bname}'. Expected it to match argument "
                         f"'{aname}' which is of type {da}.")
    sa, sb = _shape2(a), _shape2(b)
    if sa[0] == 0 or sb[0] == 0:
        raise ValueError(zero_fmt.format(sa[0], sa[1], sb[0], sb[1]))
    if sa[1] != 3 or sb[1] != 3:
        raise ValueError(dim_fmt.format(sa[0], sa[1], sb[0], sb[1]))
    return da


_KNN_ZERO = ("Invalid input set with zero elements: query_points and dataset_points must have shape (n, 3) and (m, 3). "
             "Got query_points.shape = ({}, {}), dataset_points.shape = ({}, {}).")
_KNN_DIM = ("Only 3D inputs are supported: query_points and dataset_points must have shape (n, 3) and (m, 3). "
            "Got query_points.shape = ({}, {}), dataset_points.shape = ({}, {}).")
_HD_ZERO = ("Invalid input set with zero elements: source and targets must have shape (n, 3) and (m, 3). "
            "Got source.shape = ({}, {}), target.shape = ({}, {}).")
_HD_DIM = ("Only 3D inputs are supported: source and targets must have shape (n, 3) and (m, 3). "
           "Got source.shape = ({}, {}), target.shape = ({}, {}).")


class _Dev:
    """Resolved inputs of one call: contiguous buffers, pointers, device, stream, flags."""

    def __init__(self, a, b):
        self.torch = _is_torch(a) or _is_torch(b)
        if self.torch:
            import torch
            if not (_is_torch(a) and _is_torch(b)) or not (a.is_cuda and b.is_cuda) or a.device != b.device:
                raise ValueError("torch inputs must both be CUDA/HIP tensors on the same device")
            self.a, self.b = a.contiguous(), b.contiguous()
            self.device = a.device.index if a.device.index is not None else torch.cuda.current_device()
            self.tdev = a.device
            # the library launches on torch's current stream, so its kernels are ordered after the producers of the input
            # tensors (torch's default stream is the legacy NULL stream, handle 0: STREAM_GIVEN makes NULL mean exactly that)
            raw = getattr(torch._C, "_cuda_getCurrentRawStream", None)       # (the handle without building a Stream object: ~2 us per call)
            self.stream = raw(self.device) if raw is not None else torch.cuda.current_stream(a.device).cuda_stream
            self.flags = _lib.PTRS_ON_DEVICE | _flags() | _lib.STREAM_GIVEN
            self.pa, self.pb = self.a.data_ptr(), self.b.data_ptr()
            self.np_dtype = np.float32 if a.dtype == torch.float32 else np.float64
            self.t_dtype = a.dtype
        else:
            self.a, self.b = np.ascontiguousarray(a), np.ascontiguousarray(b)
            self.device = _lib.default_device()
            self.stream = None
            self.flags = _flags()
            self.pa, self.pb = self.a.ctypes.data, self.b.ctypes.data
            self.np_dtype = self.a.dtype.type
        self.suffix = "f32" if self.np_dtype == np.float32 else "f64"
        self.ctx = _lib.ctx(self.device)

    def empty(self, shape, kind):
        """kind: 'T' (input dtype) or 'i64'."""
        if self.torch:
            import torch
            return torch.empty(shape, dtype=self.t_dtype if kind == "T" else torch.int64, device=self.tdev)
        return np.empty(shape, dtype=self.np_dtype if kind == "T" else np.int64)

    @staticmethod
    def ptr(x):
        if x is None:
            return None
        return x.data_ptr() if _is_torch(x) else x.ctypes.data


def _record(st):
    _last_stats[0] = st


def _squeeze(dists, corrs, n, k):
    """npe::move(..., squeeze): singleton dimensions are dropped, as numpy.squeeze does -- (n,) for k == 1 (pinned by the
    reference's tests/test_examples.py:363-368), (k,) for n == 1, 0-d for n == k == 1 (unpinned in the reference)."""
    if n == 1 and k == 1:
        return dists.reshape(()), corrs.reshape(())
    return dists.reshape(-1), corrs.reshape(-1)


_FN = {}


def _fn(op, suffix):
    """Entry point `pcu_hip_<op>_<suffix>` (looked up once)."""
    f = _FN.get((op, suffix))
    if f is None:
        f = _FN[(op, suffix)] = getattr(_lib.lib(), f"pcu_hip_{op}_{suffix}")
    return f


def k_nearest_neighbors(query_points, dataset_points, k, squared_distances=False, max_points_per_leaf=10,
                        num_threads=-1):
    """
    Compute the k nearest neighbors (L2 distance) from each point in the query point cloud to the dataset point cloud.

    Args:
        query_points : n by 3 array of representing a set of n points (each row is a point of dimension 3).
        dataset_points : m by 3 array of representing a set of m points (each row is a point of dimension 3).
        k : the number of nearest neighbors to query per point. Any k > 0 (k <= 127: grid search; larger k: the reference's
            kd-tree traversal run on the GPU, slower per query). If k exceeds the dataset size the trailing slots hold -1 / -1.0.
        squared_distances : If set to True, then return squared L2 distances. Default is False.
        max_points_per_leaf : leaf size of the reference's kd-tree. Does not change distances; it only fixes the order of exactly tied neighbours, which is reproduced.
        num_threads : OpenMP knob of the reference; accepted and ignored.

    Limit of this implementation: point clouds of more than 2**27 - 16 (134,217,712) rows are rejected with a ValueError.

    Returns:
        dists : An (n, k)-shaped array such that `dists[i,k]` contains the k^th shortest L2 distance from the point `query_points[i, :]` to `dataset_points`
        corrs : An (n, k)-shaped array such that `corrs[i,k]` contains the index into `dataset_points` of the k^th nearest point to `query_points[i, :]`
        (singleton dimensions are squeezed, as the reference's binding does: k == 1 gives (n,)-shaped results)
    """
    k = int(k)
    if k <= 0:   # src/point_cloud_distance.cpp:133-135
        raise ValueError(f"Invalid value for k ({k}) must be greater than 0.")
    _check_pair(query_points, dataset_points, "query_points", "dataset_points", _KNN_ZERO, _KNN_DIM)
    d = _Dev(query_points, dataset_points)
    n, m = int(d.a.shape[0]), int(d.b.shape[0])
    dists = d.empty((n, k), "T")
    corrs = d.empty((n, k), "i64")
    st = Stats()
    flags = d.flags | (_lib.SQUARED if squared_distances else 0)
    rc = _fn("knn", d.suffix)(d.ctx, d.pa, n, d.pb, m, k, int(max_points_per_leaf), _Dev.ptr(dists), _Dev.ptr(corrs),
                              flags, d.stream, ctypes.addressof(st))
    _lib.check(rc)
    _record(st)
    if k == 1 or n == 1:
        return _squeeze(dists, corrs, n, k)
    if not d.torch:
        q = np.asarray(query_points)
        if q.flags.f_contiguous and not q.flags.c_contiguous:      # EigenDenseLike keeps the query's storage order
            dists = np.asfortranarray(dists)
    return dists, corrs


def one_sided_hausdorff_distance(source, target, return_index=True, squared_distances=False, max_points_per_leaf=10):
    """
    Compute the one sided Hausdorff distance from source to target

    Args:
        source : n by 3 array of representing a set of n points (each row is a point of dimension 3)
        target : m by 3 array of representing a set of m points (each row is a point of dimension 3)
        return_index : Optionally return the index pair `(i, j)` into source and target such that `source[i, :]` and `target[j, :]` are the two points with maximum shortest distance.
        squared_distances : If set to True, then return squared L2 distances.
        max_points_per_leaf : leaf size of the reference's kd-tree. Does not change distances; it only fixes the order of exactly tied neighbours, which is reproduced.

    Returns:
        d : The largest shortest distance, `d` between each point in `source` and the points in `target`.
        i, j : (if return_index) indices such that `source[i, :]` and `target[j, :]` are the two points with maximum shortest distance.
    """
    _check_pair(source, target, "source", "target", _HD_ZERO, _HD_DIM)
    d = _Dev(source, target)
    n, m = int(d.a.shape[0]), int(d.b.shape[0])
    od = np.zeros(2, dtype=d.np_dtype)
    oi = np.zeros(2, dtype=np.int64)
    oj = np.zeros(2, dtype=np.int64)
    st = Stats()
    flags = d.flags | (_lib.SQUARED if squared_distances else 0)
    rc = _fn("one_sided_hausdorff", d.suffix)(
        d.ctx, d.pa, n, d.pb, m, int(max_points_per_leaf), od.ctypes.data, oi.ctypes.data, oj.ctypes.data, flags, d.stream,
        ctypes.addressof(st))
    _lib.check(rc)
    _record(st)
    if return_index:
        return float(od[0]), int(oi[0]), int(oj[0])
    return float(od[0])


def hausdorff_distance(x, y, return_index=False, squared_distances=False, max_points_per_leaf=10):
    """
    Compute the Hausdorff distance between x and y

    Args:
        x : n by 3 array of representing a set of n points (each row is a point of dimension 3)
        y : m by 3 array of representing a set of m points (each row is a point of dimension 3)
        return_index : Optionally return the index pair `(i, j)` into x and y such that `x[i, :]` and `y[j, :]` are the two points with maximum shortest distance.
        squared_distances : If set to True, then return squared L2 distances. Default is False.
        max_points_per_leaf : leaf size of the reference's kd-tree. Does not change distances; it only fixes the order of exactly tied neighbours, which is reproduced.

    Returns:
        The largest shortest distance, `d` between each point in `source` and the points in `target`.
        If `return_index` is set, then this function returns a tuple (d, i, j).
    """
    _check_pair(x, y, "source", "target", _HD_ZERO, _HD_DIM)
    d = _Dev(x, y)
    n, m = int(d.a.shape[0]), int(d.b.shape[0])
    od = np.zeros(2, dtype=d.np_dtype)
    oi = np.zeros(2, dtype=np.int64)
    oj = np.zeros(2, dtype=np.int64)
    st = Stats()
    flags = d.flags | (_lib.SQUARED if squared_distances else 0)
    rc = _fn("hausdorff", d.suffix)(   # one call: both clouds are indexed once and searched in both directions
        d.ctx, d.pa, n, d.pb, m, int(max_points_per_leaf), od.ctypes.data, oi.ctypes.data, oj.ctypes.data, flags, d.stream,
        ctypes.addressof(st))
    _lib.check(rc)
    _record(st)
    # point_cloud_utils/__init__.py:69-81, on Python floats exactly as there
    hausdorff_x_to_y, idx_x1, idx_y1 = float(od[0]), int(oi[0]), int(oj[0])
    hausdorff_y_to_x, idx_y2, idx_x2 = float(od[1]), int(oi[1]), int(oj[1])
    hausdorff = max(hausdorff_x_to_y, hausdorff_y_to_x)
    if return_index and hausdorff_x_to_y > hausdorff_y_to_x:
        return hausdorff, idx_x1, idx_y1
    elif return_index and hausdorff_x_to_y <= hausdorff_y_to_x:
        return hausdorff, idx_x2, idx_y2
    return hausdorff


def chamfer_distance(x, y, return_index=False, p_norm=2, max_points_per_leaf=10):
    """
    Compute the chamfer distance between two point clouds x, and y

    Args:
        x : n by 3 array of points
        y : m by 3 array of points
        return_index: If set to True, will return a pair (corrs_x_to_y, corrs_y_to_x) where
                    corrs_x_to_y[i] stores the index into y of the closest point to x[i]
                    (i.e. y[corrs_x_to_y[i]] is the nearest neighbor to x[i] in y).
                    corrs_y_to_x is similar to corrs_x_to_y but with x and y reversed.
        max_points_per_leaf : leaf size of the reference's kd-tree. Does not change distances; it only fixes the order of exactly tied neighbours, which is reproduced.
        p_norm : Which norm to use. p_norm can be any real number, inf (for the max norm) -inf (for the min norm),
                0 (for sum(x != 0))
    Returns:
        The chamfer distance between x an dy.
        If return_index is set, then this function returns a tuple (chamfer_dist, corrs_x_to_y, corrs_y_to_x).
    """
    _check_pair(x, y, "query_points", "dataset_points", _KNN_ZERO, _KNN_DIM)
    d = _Dev(x, y)
    n, m = int(d.a.shape[0]), int(d.b.shape[0])
    cxy = d.empty((n,), "i64") if return_index else None
    cyx = d.empty((m,), "i64") if return_index else None
    means = (ctypes.c_double * 2)()
    st = Stats()
    rc = _fn("chamfer", d.suffix)(
        d.ctx, d.pa, n, d.pb, m, float(p_norm), int(max_points_per_leaf), ctypes.addressof(means), _Dev.ptr(cxy), _Dev.ptr(cyx),
        d.flags, d.stream, ctypes.addressof(st))
    if rc:
        _lib.check(rc)
    _record(st)
    # __init__.py:112-115: both means are scalars of the input dtype (np.mean of such a scalar is that scalar);
    # their sum, in that dtype, is the result
    dists_x_to_y = d.np_dtype(means[1])      # norm(x[corrs_y_to_x] - y).mean()
    dists_y_to_x = d.np_dtype(means[0])      # norm(y[corrs_x_to_y] - x).mean()
    cham_dist = dists_x_to_y + dists_y_to_x
    if return_index:
        return cham_dist, cxy, cyx
    return cham_dist


class DatasetIndex:
    """A dataset kept on the GPU together with its search index (not in the reference API, which rebuilds its kd-tree on
    every call -- three times, src/point_cloud_distance.cpp:41-42): build once, query many times.

        index = pcu.DatasetIndex(dataset_points, k_hint=8)
        dists, corrs = index.k_nearest_neighbors(query_points, 8)       # same results as pcu.k_nearest_neighbors

    `dataset_points`: (m, 3) float32 / float64, numpy or CUDA/HIP torch tensor (copied; the caller's array can go away).
    `k_hint` sizes the grid cells for the k that will mostly be asked for; any k > 0 is answered exactly.
    Queries must have the dataset's dtype. The index lives on one GPU; call close() (or use `with`) to free it."""

    def __init__(self, dataset_points, k_hint=1):
        da = _dtype_name(dataset_points)
        if da not in ("float32", "float64"):
            raise ValueError(f"Invalid scalar type ({da}) for argument 'dataset_points'. Expected one of ['float32', 'float64'].")
        sh = _shape2(dataset_points)
        if sh[0] == 0 or sh[1] != 3:
            raise ValueError(f"dataset_points must have shape (m, 3) with m > 0. Got dataset_points.shape = ({sh[0]}, {sh[1]}).")
        d = _Dev(dataset_points, dataset_points)
        self._suffix, self._np_dtype, self._torch, self._device = d.suffix, d.np_dtype, d.torch, d.device
        h = ctypes.c_void_p()
        rc = _fn("index_create", d.suffix)(d.ctx, d.pa, int(d.a.shape[0]), int(k_hint), d.flags, d.stream, ctypes.byref(h))
        if rc:
            _lib.check(rc)
        self._h = h
        self.num_points = int(d.a.shape[0])

    def k_nearest_neighbors(self, query_points, k, squared_distances=False, max_points_per_leaf=10):
        """See point_cloud_utils_amd.k_nearest_neighbors; the dataset is the indexed one."""
        if self._h is None:
            raise ValueError("the index has been closed")
        k = int(k)
        if k <= 0:
            raise ValueError(f"Invalid value for k ({k}) must be greater than 0.")
        dq = _dtype_name(query_points)
        want = "float32" if self._suffix == "f32" else "float64"
        if dq not in ("float32", "float64"):
            raise ValueError(f"Invalid scalar type ({dq}) for argument 'query_points'. Expected one of ['float32', 'float64'].")
        if dq != want:
            raise ValueError(f"Invalid scalar type ({dq}) for argument 'query_points'. Expected it to match the indexed dataset which is of type {want}.")
        sq = _shape2(query_points)
        if sq[0] == 0:
            raise ValueError(_KNN_ZERO.format(sq[0], sq[1], self.num_points, 3))
        if sq[1] != 3:
            raise ValueError(_KNN_DIM.format(sq[0], sq[1], self.num_points, 3))
        d = _Dev(query_points, query_points)
        if d.torch and d.device != self._device:
            raise ValueError("query tensor and index live on different devices")
        n = int(d.a.shape[0])
        dists = d.empty((n, k), "T")
        corrs = d.empty((n, k), "i64")
        st = Stats()
        flags = d.flags | (_lib.SQUARED if squared_distances else 0)
        rc = _fn("index_knn", d.suffix)(d.ctx, self._h, d.pa, n, k, int(max_points_per_leaf), _Dev.ptr(dists), _Dev.ptr(corrs),
                                        flags, d.stream, ctypes.addressof(st))
        if rc:
            _lib.check(rc)
        _record(st)
        if k == 1 or n == 1:
            return _squeeze(dists, corrs, n, k)
        if not d.torch:
            q = np.asarray(query_points)
            if q.flags.f_contiguous and not q.flags.c_contiguous:
                dists = np.asfortranarray(dists)
        return dists, corrs

    def close(self):
        if getattr(self, "_h", None) is not None:
            _lib.lib().pcu_hip_index_destroy(self._h)
            self._h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()
        return False

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


from ._normals import estimate_point_cloud_normals_knn, estimate_point_cloud_normals_ball  # noqa: E402,F401
from ._voxel import (morton_encode, morton_decode, morton_add, morton_subtract, morton_knn,  # noqa: E402,F401
                     downsample_point_cloud_on_voxel_grid, deduplicate_point_cloud)
from ._sinkhorn import pairwise_distances, sinkhorn, earth_movers_distance  # noqa: E402,F401
