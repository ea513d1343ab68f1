"""pairwise_distances / sinkhorn / earth_movers_distance: the reference's pure-numpy module
(point_cloud_utils/_sinkhorn.py:4-156) on the HIP kernels of csrc/sinkhorn.h. Same arguments, shape handling, error texts and
returns; inputs may be numpy arrays or CUDA/HIP torch tensors (device-resident)."""
import ctypes

import numpy as np


def _prep(arrs):
    """contiguous buffers, (ctx, flags, stream, torch?, device), dtype suffix"""
    from ._voxel import _ctx_flags
    from . import _dtype_name
    ctx, flags, stream, t, tdev = _ctx_flags(*arrs)
    dn = _dtype_name(arrs[0])
    if dn not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({dn}): expected float32 or float64")
    out = [a.contiguous() if t else np.ascontiguousarray(a) for a in arrs]
    return out, (ctx, flags, stream, t, tdev), ("f32" if dn == "float32" else "f64"), (np.float32 if dn == "float32" else np.float64)


def _ord_value(p):
    if p is None:
        return float("nan")
    if isinstance(p, str):
        raise ValueError("Invalid norm order for vectors.")        # numpy's message for 'fro' / 'nuc' with an axis
    return float(p)


def _promote_pair(a, b):
    """The dtype numpy gives `norm(a - b)` (reference: _sinkhorn.py:30-32), by numpy's own promotion table: float32 stays float32 against
    float32 / float16 / bool / 8-bit integers, everything else is float64 (wider integers with float32, integers alone: the norm of an
    integer array is float64). Two differences: a float16 result is computed and returned in float32 (no half kernels); and integers are
    converted BEFORE the subtraction, so small integer types do not wrap around as numpy's `a - b` does (uint8 3 - 5 is -2 here, 254 there)
    -- the distance of the values, not of their wrapped difference."""
    from . import _dtype_name, _is_torch

    def np_dtype(x):
        try:
            return np.dtype(_dtype_name(x))
        except TypeError:
            return np.dtype(np.float32)            # (torch.bfloat16 and the like: no numpy counterpart)
    rt = np.result_type(np_dtype(a), np_dtype(b))
    want = "float32" if rt in (np.dtype(np.float16), np.dtype(np.float32)) else "float64"
    out = []
    for x in (a, b):
        if _dtype_name(x) != want:
            if _is_torch(x):
                import torch
                x = x.to(torch.float32 if want == "float32" else torch.float64)
            else:
                x = np.asarray(x).astype(np.float32 if want == "float32" else np.float64)
        out.append(x)
    return out


def _expand(x, shape):
    from . import _is_torch
    if tuple(x.shape) == tuple(shape):
        return x
    return x.expand(*shape) if _is_torch(x) else np.broadcast_to(x, shape)


def pairwise_distances(a, b, p=None):
    """
    Compute the (batched) pairwise distance matrix between a and b which both have size [m, n, d] or [n, d]. The result is a tensor of size [m, n, n] (or [n, n]) whose entry [m, i, j] contains the distance_tensor between a[m, i, :] and b[m, j, :].

    Args:
      a : A tensor containing m batches of n points of dimension d. i.e. of size (m, n, d)
      b : A tensor containing m batches of n points of dimension d. i.e. of size (m, n, d)
      p : Norm to use for the distance_tensor (numpy.linalg.norm's vector `ord`; None = 2)

    Returns:
      M : A (m, n, n)-shaped array containing the pairwise distance_tensor between each pair of inputs in a batch.
    """
    from . import _lib, _dtype_name
    from ._voxel import _empty, _ptr
    squeezed = False
    if len(a.shape) == 2 and len(b.shape) == 2:
        a = a[None, :, :]; b = b[None, :, :]
        squeezed = True
    if len(a.shape) != 3:
        raise ValueError("Invalid shape for a. Must be [m, n, d] or [n, d] but got", a.shape)
    if len(b.shape) != 3:
        raise ValueError("Invalid shape for a. Must be [m, n, d] or [n, d] but got", b.shape)
    # the reference is one numpy expression (_sinkhorn.py:30-32): `a[:, :, None, :] - b[:, None, :, :]` broadcasts a batch of 1 against m and
    # d = 1 against d, and promotes integer / mixed inputs as numpy does (int -> float64, float32 with float64 -> float64)
    for ax in (0, 2):
        if a.shape[ax] != b.shape[ax] and 1 not in (a.shape[ax], b.shape[ax]):
            raise ValueError(f"operands could not be broadcast together with shapes {tuple(a.shape)} {tuple(b.shape)}")
    nb_, d_ = max(a.shape[0], b.shape[0]), max(a.shape[2], b.shape[2])
    if a.shape[0] == 0 or b.shape[0] == 0: nb_ = 0
    if a.shape[2] == 0 or b.shape[2] == 0: d_ = 0
    a, b = _promote_pair(a, b)
    a = _expand(a, (nb_, a.shape[1], d_)); b = _expand(b, (nb_, b.shape[1], d_))
    (a, b), (ctx, flags, stream, t, tdev), suf, npd = _prep([a, b])
    nb, m, d = (int(x) for x in a.shape); n = int(b.shape[1])
    out = _empty((nb, m, n), npd, t, tdev)
    if d == 0:          # norm over an empty last axis: numpy gives zeros of shape (nb, m, n); the kernel has nothing to launch over
        out = out.zero_() if t else np.zeros((nb, m, n), dtype=npd)
    else:
        _lib.check(getattr(_lib.lib(), "pcu_hip_pairwise_" + suf)(ctx, _ptr(a), _ptr(b), nb, m, n, d, _ord_value(p), _ptr(out), flags, stream))
    if squeezed:
        out = out.squeeze() if t else np.squeeze(out)
    return out


def sinkhorn(a, b, M, eps, max_iters=100, stop_thresh=1e-3):
    """
    Compute the (batched) Sinkhorn correspondences between two dirac delta distributions, U, and V.
    This implementation is numerically stable with float32.

    Args:
      a : A m-sized minibatch of weights for each dirac in the first distribution, U. i.e. shape = (m, n)
      b : A m-sized minibatch of weights for each dirac in the second distribution, V. i.e. shape = (m, n)
      M : A minibatch of n-by-n tensors storing the distance between each pair of diracs in U and V. i.e. shape = (m, n, n) and each i.e. M[k, i, j] = ||u[k,_i] - v[k, j]||
      eps : The reciprocal of the sinkhorn regularization parameter
      max_iters : The maximum number of Sinkhorn iterations
      stop_thresh : Stop if the change in iterates is below this value

    Returns:
      P : An (m, n, n)-shaped array of correspondences between distributions U and V
    """
    from . import _lib, _is_torch
    from ._voxel import _empty, _ptr
    sq = (lambda x: x.squeeze()) if _is_torch(M) else np.squeeze
    M = sq(M); a = sq(a); b = sq(b)
    squeezed = False
    if len(M.shape) == 2 and len(a.shape) == 1 and len(b.shape) == 1:
        M = M[None, :, :]; a = a[None, :]; b = b[None, :]
        squeezed = True
    elif len(M.shape) == 2 and len(a.shape) != 1:
        raise ValueError("Invalid shape for a %s, expected [m,] where m is the number of samples in a and "
                         "M has shape [m, n]" % str(tuple(a.shape)))
    elif len(M.shape) == 2 and len(b.shape) != 1:
        raise ValueError("Invalid shape for a %s, expected [m,] where n is the number of samples in a and "
                         "M has shape [m, n]" % str(tuple(b.shape)))
    if len(M.shape) != 3:
        raise ValueError("Got unexpected shape for M %s, should be [nb, m, n] where nb is batch size, and "
                         "m and n are the number of samples in the two input measures." % str(tuple(M.shape)))
    elif len(M.shape) == 3 and len(a.shape) != 2:
        raise ValueError("Invalid shape for a %s, expected [nb, m]  where nb is batch size, m is the number of samples "
                         "in a and M has shape [nb, m, n]" % str(tuple(a.shape)))
    elif len(M.shape) == 3 and len(b.shape) != 2:
        raise ValueError("Invalid shape for a %s, expected [nb, m]  where nb is batch size, m is the number of samples "
                         "in a and M has shape [nb, m, n]" % str(tuple(b.shape)))
    nb, m, n = (int(x) for x in M.shape)
    if a.dtype != b.dtype or a.dtype != M.dtype:
        raise ValueError("Tensors a, b, and M must have the same dtype got: dtype(a) = %s, dtype(b) = %s, dtype(M) = %s"
                         % (str(a.dtype), str(b.dtype), str(M.dtype)))
    if tuple(a.shape) != (nb, m):
        raise ValueError("Got unexpected shape for tensor a (%s). Expected [nb, m] where M has shape [nb, m, n]." % str(tuple(a.shape)))
    if tuple(b.shape) != (nb, n):
        raise ValueError("Got unexpected shape for tensor b (%s). Expected [nb, n] where M has shape [nb, m, n]." % str(tuple(b.shape)))
    (a, b, M), (ctx, flags, stream, t, tdev), suf, npd = _prep([a, b, M])
    P = _empty((nb, m, n), npd, t, tdev)
    iters = ctypes.c_int(0)
    _lib.check(getattr(_lib.lib(), "pcu_hip_sinkhorn_" + suf)(ctx, _ptr(a), _ptr(b), _ptr(M), nb, m, n, float(eps), int(max_iters), float(stop_thresh),
                                                              _ptr(P), ctypes.byref(iters), flags, stream))
    sinkhorn.last_iterations = int(iters.value)
    if squeezed:
        P = P.squeeze() if t else np.squeeze(P)
    return P


def earth_movers_distance(p, q, p_norm=2, eps=1e-4, max_iters=100, stop_thresh=1e-3):
    """
    Compute the (batched) Sinkhorn correspondences between two dirac delta distributions, U, and V.
    This implementation is numerically stable with float32.

    Args:
      p : An (n, d)-shaped array of d-dimensional points
      b : An (m, d)-shaped array of d-dimensional points
      p_norm : Which norm to use. Must be one of {non-zero int, inf, -inf, ‘fro’, ‘nuc’} (default is 2),
      eps : The reciprocal of the sinkhorn regularization parameter (default 1e-4)
      max_iters : The maximum number of Sinkhorn iterations
      stop_thresh : Stop if the change in iterates is below this value

    Returns:
      emd : The earth mover's distance between point clouds p and q
      P : An (n, m)-shaped array of correspondences between point clouds p and q
    """
    from . import _lib, _is_torch
    from ._voxel import _ptr
    M = pairwise_distances(p, q, p_norm)
    if _is_torch(p):
        import torch
        a = torch.ones(p.shape[0], dtype=torch.float64, device=p.device) / p.shape[0]      # np.ones(n) / n: float64 weights, as the
        b = torch.ones(q.shape[0], dtype=torch.float64, device=q.device) / q.shape[0]      # reference builds them (:151-152)
    else:
        a = np.ones(p.shape[0]) / p.shape[0]
        b = np.ones(q.shape[0]) / q.shape[0]
    P = sinkhorn(a, b, M, eps, max_iters, stop_thresh)
    (Pc, Mc), (ctx, flags, stream, t, tdev), suf, npd = _prep([P, M])
    out = ctypes.c_double(0.0)
    _lib.check(getattr(_lib.lib(), "pcu_hip_dot_" + suf)(ctx, _ptr(Pc), _ptr(Mc), int(np.prod(Pc.shape)), ctypes.byref(out), flags, stream))
    return npd(out.value), P
