"""Morton codes, voxel-grid downsampling and duplicate removal over the HIP kernels of csrc/morton.h and
csrc/voxel.h: the reference's Python surface (point_cloud_utils/__init__.py:123-200; bindings src/morton.cpp:27-414,
src/sample_point_cloud.cpp:336-368, src/remove_duplicates.cpp:108-129) with the same arguments, checks and returns."""
import ctypes

import numpy as np


def _ctx_flags(*arrays):
    """(ctx, flags, stream, torch?, device) for numpy (host) or CUDA/HIP torch inputs."""
    from . import _lib, _is_torch, _flags
    if any(_is_torch(a) for a in arrays if a is not None):
        import torch
        ts = [a for a in arrays if a is not None]
        if not all(_is_torch(a) and a.is_cuda and a.device == ts[0].device for a in ts):
            raise ValueError("torch inputs must all be CUDA/HIP tensors on the same device")
        dev = ts[0].device.index if ts[0].device.index is not None else torch.cuda.current_device()
        return _lib.ctx(dev), _lib.PTRS_ON_DEVICE | _lib.STREAM_GIVEN | (_flags() & _lib.NO_TIE_ORDER), torch.cuda.current_stream(ts[0].device).cuda_stream, True, ts[0].device
    return _lib.ctx(_lib.default_device()), 0, None, False, None


def _ptr(x):
    if x is None:
        return None
    return x.data_ptr() if hasattr(x, "data_ptr") else x.ctypes.data


def _empty(shape, np_dtype, torch_in, tdev):
    if torch_in:
        import torch
        return torch.empty(shape, dtype=getattr(torch, np.dtype(np_dtype).name), device=tdev)
    return np.empty(shape, dtype=np_dtype)


def _as(x, np_dtype):
    """x converted with C cast semantics (what the reference's assignments do), contiguous."""
    if hasattr(x, "data_ptr"):
        import torch
        return x.to(getattr(torch, np.dtype(np_dtype).name)).contiguous()
    return np.ascontiguousarray(np.asarray(x).astype(np_dtype, copy=False))


def _int_kind(x, allowed, name):
    dn = str(x.dtype).replace("torch.", "")
    if dn not in allowed:
        raise ValueError(f"Invalid scalar type ({dn}) for argument '{name}'. Expected one of {list(allowed)}.")
    return dn


# ------------------------------------------------------------------------------------------------------------ Morton codes
def morton_encode(pts, num_threads=-1):
    """
    Encode n 3D points using Morton coding, possibly sorting them

    Args:
        pts : an (n, 3)-shaped array of 3D points (int32 or int64; coordinates in [-2^20, 2^20))
        num_threads : OpenMP knob of the reference; accepted and ignored.

    Returns:
        morton_codes : an (n,)-shaped uint64 array of morton encoded points
    """
    from . import _lib
    _int_kind(pts, ("int32", "int64"), "pts")
    if len(pts.shape) != 2 or int(pts.shape[0]) <= 0:
        raise ValueError("pts must be an array of shape [n, 3] but got an empty array")
    if int(pts.shape[1]) != 3:
        raise ValueError("pts must be an array of shape [n, 3] but got an invalid number of columns")
    ctx, flags, stream, t, tdev = _ctx_flags(pts)
    p = _as(pts, np.int32)                       # int32_t px = pts(i, 0): narrowing as the reference narrows
    n = int(p.shape[0])
    if t:
        import torch
        out = torch.empty((n,), dtype=torch.int64, device=tdev)       # torch has no uint64 arithmetic: same bits, viewed as int64
    else:
        out = np.empty((n,), dtype=np.uint64)
    _lib.check(_lib.lib().pcu_hip_morton_encode(ctx, _ptr(p), n, _ptr(out), flags, stream))
    return out


def morton_decode(codes, num_threads=-1):
    """
    Decode n points along a Morton curve into 3D points

    Args:
        codes : an (n,)-shaped array of Morton codes (uint32 or uint64; torch: int64 holding the same bits)

    Returns:
        points : an (n, 3)-shaped int32 array of 3D points
    """
    from . import _lib
    c, n, (ctx, flags, stream, t, tdev) = _codes(codes, "codes")
    out = _empty((n, 3), np.int32, t, tdev)
    _lib.check(_lib.lib().pcu_hip_morton_decode(ctx, _ptr(c), n, _ptr(out), flags, stream))
    return out


def _codes(codes, name, others=()):
    if hasattr(codes, "data_ptr"):
        if str(codes.dtype) != "torch.int64":
            raise ValueError(f"Invalid scalar type ({codes.dtype}) for argument '{name}': torch codes are int64 tensors holding the uint64 bits")
        c = codes.contiguous().reshape(-1)
    else:
        _int_kind(codes, ("uint32", "uint64"), name)
        if codes.ndim == 2 and codes.shape[1] != 1:
            raise ValueError(f"{name} must be an array of shape [n] but got an invalid shape")
        c = _as(codes.reshape(-1), np.uint64)
    n = int(c.shape[0])
    if n <= 0:
        raise ValueError(f"{name} must be an array of shape [n] but got an empty array")
    return c, n, _ctx_flags(c, *others)


def _addsub(codes_1, codes_2, sub):
    from . import _lib
    c1, n, _ = _codes(codes_1, "codes_1")
    c2, n2, (ctx, flags, stream, t, tdev) = _codes(codes_2, "codes_2", (c1,))
    if n2 != n:
        raise ValueError("codes_1 and codes_2  must have the same number of entries.")
    out = _empty((n,), np.int64 if t else np.uint64, t, tdev)
    _lib.check(_lib.lib().pcu_hip_morton_addsub(ctx, _ptr(c1), _ptr(c2), n, 1 if sub else 0, _ptr(out), flags, stream))
    return out


def morton_add(codes_1, codes_2, num_threads=-1):
    """Add morton codes together (corresponding to adding the vectors they encode) -> (n,) uint64."""
    return _addsub(codes_1, codes_2, False)


def morton_subtract(codes_1, codes_2, num_threads=-1):
    """Subtract morton codes from each other (codes_1 - codes_2, i.e. subtracting the vectors they encode) -> (n,) uint64."""
    return _addsub(codes_1, codes_2, True)


def morton_knn(codes, qcodes, k, sort_dist=True):
    """
    Queries a sorted array of morton encoded points to find the (approximate) k nearest neighbors

    Args:
        codes : an (n)-shaped array of morton codes, sorted ascending
        qcodes : an (m)-shaped array of query codes
        k : an integer representing the number of nearest neighbors
        sort_dist : (optional, defaults to True) whether to return the nearest neighbors in distance sorted order. (In the
                    reference this ordering reads uninitialised memory; here the window is ordered by the distance between the
                    decoded query and the decoded entries.)

    Returns:
        nn_idx : an (m, min(k, n))-shaped int64 array of indices into codes
    """
    from . import _lib
    k = int(k)
    if k <= 0:
        raise ValueError("k must be greater than 0")
    c, n, _ = _codes(codes, "codes")
    q, m, (ctx, flags, stream, t, tdev) = _codes(qcodes, "qcodes", (c,))
    if not t and np.asarray(codes).dtype != np.asarray(qcodes).dtype:
        raise ValueError(f"Invalid scalar type ({np.asarray(qcodes).dtype}) for argument 'qcodes'. Expected it to match argument 'codes' which is of type {np.asarray(codes).dtype}.")
    kk = min(k, n)
    out = _empty((m, kk), np.int64, t, tdev)
    _lib.check(_lib.lib().pcu_hip_morton_knn(ctx, _ptr(c), n, _ptr(q), m, k, 1 if sort_dist else 0, _ptr(out), flags, stream))
    return out


# ------------------------------------------------------------------------------------------------------------ voxel grid
def _downsample_one(points, attrib, voxel_size, min_bound, max_bound, min_points_per_voxel):
    """downsample_point_cloud_voxel_grid_internal: (ret_v, ret_attrib) for one attribute matrix (or an empty one)."""
    from . import _lib, _dtype_name
    ctx, flags, stream, t, tdev = _ctx_flags(points, attrib if attrib is not None and attrib.shape[0] else None)
    pn = _dtype_name(points)
    if pn not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({pn}) for argument 'v'. Expected one of ['float32', 'float64'].")
    p = points.contiguous() if t else np.ascontiguousarray(points)
    n = int(p.shape[0])
    has_a = attrib is not None and int(np.prod(attrib.shape)) > 0
    an = _dtype_name(attrib) if has_a else pn
    if an not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({an}) for argument 'attrib'. Expected one of ['float32', 'float64'].")
    a2 = None; cols = 0; arows = 0
    if has_a:
        a2 = attrib.reshape(attrib.shape[0], -1)
        a2 = a2.contiguous() if t else np.ascontiguousarray(a2)
        arows, cols = int(a2.shape[0]), int(a2.shape[1])
    pd, ad = (np.float32 if pn == "float32" else np.float64), (np.float32 if an == "float32" else np.float64)
    out_v = _empty((max(n, 1), 3), pd, t, tdev)
    out_a = _empty((max(n, 1), max(cols, 1)), ad, t, tdev)
    vs = (ctypes.c_double * 3)(*[float(x) for x in voxel_size]); mn = (ctypes.c_double * 3)(*[float(x) for x in min_bound]); mx = (ctypes.c_double * 3)(*[float(x) for x in max_bound])
    cnt = ctypes.c_int64(0)
    fn = getattr(_lib.lib(), "pcu_hip_voxel_downsample_" + ("f32" if pn == "float32" else "f64") + "_" + ("f32" if an == "float32" else "f64"))
    _lib.check(fn(ctx, _ptr(p), n, _ptr(a2), arows, cols, vs, mn, mx, int(min_points_per_voxel), _ptr(out_v), _ptr(out_a), ctypes.byref(cnt), flags, stream))
    m = int(cnt.value)
    ret_v = out_v[:m]
    if has_a:
        ret_a = out_a[:m].reshape((m,) + tuple(attrib.shape[1:])) if len(attrib.shape) != 2 else out_a[:m]
    else:
        ret_a = _empty((0, 0), ad, t, tdev)
    return ret_v, ret_a


def downsample_point_cloud_on_voxel_grid(voxel_size, points, *args, min_bound=None, max_bound=None, min_points_per_voxel=1):
    """
    Downsample a point set to conform with a voxel grid by taking the average of points within each voxel.

    Args:
        voxel_size : a scalar representing the size of each voxel or a 3 tuple representing the size per axis of each voxel.
        points: a [#v, 3]-shaped array of 3d points.
        *args: Any additional arguments of shape [#v, *] are treated as attributes and will averaged into each voxel along with the points
        min_bound: a 3 tuple representing the minimum coordinate of the voxel grid or None to use the bounding box of the input point cloud.
        max_bound: a 3 tuple representing the maximum coordinate of the voxel grid or None to use the bounding box of the input point cloud.
        min_points_per_voxel: If a voxel contains fewer than this many points, then don't include the points in that voxel in the output.

    Returns:
        A tuple (v, attrib0, attrib1, ....) of downsampled points, and point attributes (in the order they are passed in);
        just the vertices if no attributes are passed in. Rows are ordered by voxel index (x, then y, then z) -- the reference
        returns them in the iteration order of its hash table; the voxel means themselves are bit-identical.
    """
    from . import _is_torch
    if np.isscalar(voxel_size):
        voxel_size = np.array([voxel_size] * 3)
    else:
        voxel_size = np.array(voxel_size)
        if len(voxel_size) != 3:
            raise ValueError("Invalid voxel size must be a 3-tuple or a single float")
    t = _is_torch(points)
    if not t and type(points) != np.ndarray:
        raise ValueError("points must be a numpy array but got type " + str(type(points)))
    attribs = []
    for i, arg in enumerate(args):
        if not (_is_torch(arg) if t else type(arg) == np.ndarray):
            raise ValueError("Additional arguments after points and before keyword arguments must be numpy arrays")
        if arg.shape[0] != points.shape[0]:
            raise ValueError("Attribute " + str(i) + " must have same first dimension as number of points (" + str(points.shape) + " but got attrib.shape = " + str(arg.shape))
        attribs.append(arg)
    pmin = (points.min(dim=0).values.cpu().numpy() if t else np.min(points, axis=0)) if (min_bound is None or max_bound is None) else None
    pmax = (points.max(dim=0).values.cpu().numpy() if t else np.max(points, axis=0)) if (min_bound is None or max_bound is None) else None
    if min_bound is None:
        min_bound = pmin - voxel_size * 0.5
    if max_bound is None:
        max_bound = pmax + voxel_size * 0.5
    min_bound = np.array(min_bound); max_bound = np.array(max_bound)
    if len(min_bound) != 3:
        raise ValueError("min_bound must be a 3 tuple")
    if len(max_bound) != 3:
        raise ValueError("max_bound must be a 3 tuple")
    if np.any(max_bound - min_bound <= 0.0):
        raise ValueError("Invalid min_bound and max_bound. max_bound must be greater than min_bound in all dimensions")
    ret_v, ret_a0 = _downsample_one(points, attribs[0] if attribs else None, voxel_size, min_bound, max_bound, min_points_per_voxel)
    ret = [ret_v, ret_a0] if int(np.prod(ret_a0.shape)) > 0 else [ret_v]
    for i in range(1, len(attribs)):
        _, ret_ai = _downsample_one(points, attribs[i], voxel_size, min_bound, max_bound, min_points_per_voxel)
        ret.append(ret_ai)
    return tuple(ret) if len(ret) > 1 else ret_v


def deduplicate_point_cloud(points, epsilon, return_index=True):
    """
    Removes duplicated points from a point cloud where two points are considered the same if their distance is below
    some threshold (they agree after rounding to multiples of epsilon; epsilon <= 0: exactly equal)

    Args:
        points : #x by 3 Matrix of 3D positions
        epsilon: threshold below which two points are considered equal
        return_index: If true, return indices to map between input and output

    Returns:
        x_new : #x_new x 3 Point cloud with duplicates removed (in lexicographic order of the rounded coordinates)
        if return indices is set, this function also returns:
            svi : #x_new indices (int32) so that x_new = x[svi]  (the lowest row of every group)
            svj : #x indices (int32) so that x ~ x_new[svj]
    """
    from . import _lib, _dtype_name
    pn = _dtype_name(points)
    if pn not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({pn}) for argument 'points'. Expected one of ['float32', 'float64'].")
    if len(points.shape) != 2 or int(points.shape[1]) != 3:      # validate_point_cloud, src/common/common.h:58-74 (zero rows are allowed)
        sh = tuple(points.shape) + (1,) * (2 - len(points.shape))
        raise ValueError(f"Only 3D inputs are supported: v must have shape (n, 3) (n > 0). Got points.shape =({sh[0]}, {sh[1]}).")
    ctx, flags, stream, t, tdev = _ctx_flags(points)
    p = points.contiguous() if t else np.ascontiguousarray(points)
    n = int(p.shape[0])
    pd = np.float32 if pn == "float32" else np.float64
    out = _empty((n, 3), pd, t, tdev); svi = _empty((n,), np.int32, t, tdev); svj = _empty((n,), np.int32, t, tdev)
    cnt = ctypes.c_int64(0)
    _lib.check(getattr(_lib.lib(), "pcu_hip_dedup_" + ("f32" if pn == "float32" else "f64"))(ctx, _ptr(p), n, float(epsilon), _ptr(out), _ptr(svi), _ptr(svj),
                                                                                          ctypes.byref(cnt), flags, stream))
    m = int(cnt.value)
    if return_index:
        return out[:m], svi[:m], svj
    return out[:m]
