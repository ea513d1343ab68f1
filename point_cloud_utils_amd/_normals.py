"""estimate_point_cloud_normals_knn / _ball: the reference's wrappers (point_cloud_utils/_pointcloud_normals.py:4-56, :59-125)
over the HIP plane-fit kernels (csrc/normals.h) instead of `_pcu_internal`. Same arguments, checks and returns."""
import ctypes

import numpy as np


def _run(kind, points, view_directions, drop_angle_threshold, max_points_per_leaf, **kw):
    from . import _lib, _Dev, _is_torch, _record, Stats
    if view_directions is None:
        view_directions = np.zeros([0, 3], dtype=points.dtype) if not _is_torch(points) else None
    torch_in = _is_torch(points)
    if not torch_in:
        if type(view_directions) != np.ndarray:
            raise ValueError("Invalid type for view_directions, must be None or a NumPy array, but got " + str(type(view_directions)) + ".")
        if type(points) != np.ndarray:
            raise ValueError("Invalid type for points, must be None or a NumPy array, but got " + str(type(points)) + ".")
    if len(points.shape) != 2 or points.shape[-1] != 3:
        raise ValueError("Invalid shape for points, must be (n, 3) but got " + str(tuple(points.shape)))
    if view_directions is not None and len(view_directions.shape) != 2:
        raise ValueError("Invalid shape for view_directions, must be (n, 3) but got " + str(tuple(view_directions.shape)))
    has_dirs = view_directions is not None and int(view_directions.shape[0]) > 0
    n = int(points.shape[0])
    if n == 0:       # validate_input, src/point_cloud_normals.cpp:27-33
        raise ValueError(f"Invalid point set with zero elements: points must have shape (n, 3), but got ot points.shape = ({n}, {points.shape[1]}).")
    if has_dirs and (int(view_directions.shape[0]) != n or int(view_directions.shape[1]) != 3):
        raise ValueError("Invalid view directions does not match the number of points. If view directions are passed in, they must have the same shape "
                         f"as points. Got points.shape = ({n}, {points.shape[1]}), and view_dirs.shape = ({view_directions.shape[0]}, {view_directions.shape[1]}).")
    from . import _dtype_name
    da = _dtype_name(points)
    if da not in ("float32", "float64"):
        raise ValueError(f"Invalid scalar type ({da}) for argument 'points'. Expected one of ['float32', 'float64'].")
    if has_dirs and _dtype_name(view_directions) != da:
        raise ValueError(f"Invalid scalar type ({_dtype_name(view_directions)}) for argument 'view_dirs'. Expected it to match argument 'points' which is of type {da}.")
    d = _Dev(points, view_directions if has_dirs else points)
    normals = d.empty((n, 3), "T")
    if d.torch:
        import torch
        keep = torch.empty((n,), dtype=torch.uint8, device=d.tdev)
    else:
        keep = np.empty((n,), dtype=np.uint8)
    st = Stats()
    pdirs = d.pb if has_dirs else None
    if kind == "knn":
        rc = getattr(_lib.lib(), "pcu_hip_normals_knn_" + d.suffix)(d.ctx, d.pa, n, pdirs, int(kw["num_neighbors"]), int(max_points_per_leaf),
                                                                    float(drop_angle_threshold), _Dev.ptr(normals), _Dev.ptr(keep), d.flags, d.stream,
                                                                    ctypes.addressof(st))
    else:
        rc = getattr(_lib.lib(), "pcu_hip_normals_ball_" + d.suffix)(d.ctx, d.pa, n, pdirs, float(kw["ball_radius"]), int(kw["min_pts_per_ball"]),
                                                                     int(kw["max_pts_per_ball"]), 1 if kw["weight_function"] == "rbf" else 0,
                                                                     float(drop_angle_threshold), _Dev.ptr(normals), _Dev.ptr(keep), d.flags, d.stream,
                                                                     ctypes.addressof(st))
    _lib.check(rc)
    _record(st)
    # the reference returns the kept points' indices (ascending in its serial driver, :250-272) and their normals
    if d.torch:
        idx = torch.nonzero(keep).reshape(-1)
        return idx, normals[idx]
    idx = np.flatnonzero(keep).astype(np.int64)
    return idx, normals[idx]


def estimate_point_cloud_normals_knn(points, num_neighbors, view_directions=None, drop_angle_threshold=np.deg2rad(90.0),
                                     max_points_per_leaf=10, num_threads=-1):
    """
    Estimate normals for a point cloud by locally fitting a plane to the k nearest neighbors of each point.

    This function can optionally consider directions to the sensor for each point to compute neighborhoods of points
    which are all facing the same direction, and align the final normal directions.

    Args:
        points : (n, 3)-shaped NumPy array of point positions (each row is a point)
        num_neighbors : Integer number of neighbors to use in each neigghborhood.
        view_directions : (n, 3)-shaped NumPy array or None, representing the unit direction to the sensor for each point. This parameter is used to align the normals and compute neighborhoods of similar facing points.
        drop_angle_threshold : If view_directions is passed in, drop points whose angle between the normal and view direction exceeds drop_angle_threshold (in radians). Useful for filtering out low quality points.
        max_points_per_leaf : leaf size of the reference's kd-tree; it only fixes the order of exactly tied neighbours, which is reproduced.
        num_threads : OpenMP knob of the reference; accepted and ignored.

    Returns:
        idx : an (m,)-shaped array of indices into points (the points that were kept, ascending)
        n : an (m, 3)-shaped array of unit normals for each of those points. Without view_directions the sign of a normal is
            not defined (in the reference it is whatever Eigen's JacobiSVD returns).
    """
    num_neighbors = int(num_neighbors)
    if num_neighbors <= 0:     # src/point_cloud_normals.cpp:385-389
        raise ValueError(f"Invalid number of neighbors ({num_neighbors}) must be greater than 0.")
    return _run("knn", points, view_directions, drop_angle_threshold, max_points_per_leaf, num_neighbors=num_neighbors)


def estimate_point_cloud_normals_ball(points, ball_radius, view_directions=None, drop_angle_threshold=np.deg2rad(90.0),
                                      min_pts_per_ball=3, max_pts_per_ball=-1, weight_function="constant",
                                      max_points_per_leaf=10, num_threads=-1):
    """
    Estimate normals for a point cloud by locally fitting a plane to all points within a radius of each point
    (possibly weighted by a radial basis function).

    Args:
        points: (n, 3)-shaped NumPy array of point positions (each row is a point)
        ball_radius: The radius of each neighborhood used to estimate normals. As in the reference, the neighbourhood of a point
                     is what nanoflann's radiusSearch returns for this value: the points whose SQUARED distance is below it.
        view_directions: (n, 3)-shaped NumPy array or None, the unit direction to the sensor for each point.
        drop_angle_threshold: drop points whose angle between the normal and view direction exceeds this (radians).
        min_pts_per_ball: Discard points whose neighborhood contains fewer than min_pts_per_ball points.
        max_pts_per_ball: If positive, fit only that many points of a larger neighbourhood (the reference draws them at random;
                          here they are taken evenly through the neighbourhood).
        weight_function: 'constant' (1.0) or 'rbf' ((1 - d/r)^4 * (4 d/r + 1), d = distance, r = ball_radius)
        max_points_per_leaf, num_threads: knobs of the reference's kd-tree / OpenMP; accepted and ignored.

    Returns:
        idx : an (m,)-shaped array of indices into points (the points that were kept, ascending)
        n : an (m, 3)-shaped array of unit normals for each of those points
    """
    # src/point_cloud_normals.cpp:318-329
    if ball_radius <= 0.0:
        raise ValueError("Invalid radius (" + "%f" % ball_radius + ") must be greater than 0.")
    if min_pts_per_ball < 3:
        raise ValueError(f"Invalid min_pts_per_ball ({min_pts_per_ball}) must be greater than 3.")
    if 0 < max_pts_per_ball < 3:
        raise ValueError(f"Invalid max_pts_per_ball ({max_pts_per_ball}) must either be negative (no max) or a number greater than 3.")
    if weight_function not in ("constant", "rbf"):
        raise ValueError("Invalid weight_function, must be one of 'constant' or 'rbf'.")
    return _run("ball", points, view_directions, drop_angle_threshold, max_points_per_leaf, ball_radius=ball_radius,
                min_pts_per_ball=min_pts_per_ball, max_pts_per_ball=max_pts_per_ball, weight_function=weight_function)
