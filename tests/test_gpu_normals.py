"""GPU parity tests (-m gpu) of estimate_point_cloud_normals_knn / _ball against the numpy restatement of
src/point_cloud_normals.cpp (oracle.normals_*: the reference's KNN sets, numpy's SVD). Tolerance: the normal is compared up
to sign (Eigen's sign convention is not part of the reference checkout) as 1 - |n . n0| <= 1e-8 for fits whose smallest
direction is separated ((s1 - s2) / s0 > 1e-2); with view directions the sign is fixed and n . n0 itself is compared, and the
set of kept points must agree except within 1e-6 rad of the threshold."""
import os

import numpy as np
import pytest

import oracle
from conftest import cloud

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0
    return m


def _surface(n, dtype, seed=3):
    """A wavy sheet with noise: well-defined normals, varied orientations."""
    rng = np.random.default_rng(seed)
    xy = rng.random((n, 2)) * 2 - 1
    z = 0.3 * np.sin(3 * xy[:, 0]) * np.cos(2 * xy[:, 1]) + rng.normal(0, 0.002, n)
    return np.ascontiguousarray(np.concatenate([xy, z[:, None]], 1).astype(dtype))


def _check(idx, nrm, idx0, nrm0, gap, signed):
    assert np.array_equal(idx, idx0)
    assert nrm.shape == nrm0.shape
    assert np.allclose(np.linalg.norm(nrm.astype(np.float64), axis=1), 1.0, atol=1e-5)
    good = gap > 1e-2
    assert good.mean() > 0.9
    dot = np.einsum("ij,ij->i", nrm.astype(np.float64), nrm0)
    tol = 1e-8 if nrm.dtype == np.float64 else 1e-6          # float32 outputs are the double normal rounded to float
    if signed:
        assert np.all(1.0 - dot[good] <= tol), float((1.0 - dot[good]).max())
    else:
        assert np.all(1.0 - np.abs(dot[good]) <= tol), float((1.0 - np.abs(dot[good])).max())


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("k", [8, 12, 30])
def test_normals_knn_vs_oracle(pcu, oracle_kind, dtype, k):
    p = _surface(60000, dtype)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(p, k)
    assert idx.dtype == np.int64 and nrm.dtype == dtype and nrm.shape == (len(idx), 3)
    idx0, nrm0, gap = oracle.normals_knn(p, k, kind=oracle_kind)
    _check(idx, nrm, idx0, nrm0, gap, signed=False)
    # view directions: sign fixed, points facing away beyond the threshold dropped
    dirs = np.tile(np.array([[0.0, 0.6, 0.8]], dtype=dtype), (len(p), 1))
    thr = np.deg2rad(40.0)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(p, k, view_directions=dirs, drop_angle_threshold=thr)
    idx0, nrm0, gap = oracle.normals_knn(p, k, view_directions=dirs, drop_angle_threshold=thr, kind=oracle_kind)
    assert 0.05 < len(idx) / len(p) < 0.999
    both = np.intersect1d(idx, idx0)
    assert len(np.setxor1d(idx, idx0)) <= 1e-3 * len(p)               # only points within rounding of the threshold may differ
    a, b = np.searchsorted(idx, both), np.searchsorted(idx0, both)
    _check(both, nrm[a], both, nrm0[b], gap[b], signed=True)


@pytest.mark.parametrize("shape", ["ribbon", "flat"])
def test_normals_thin_neighbourhoods_f64(pcu, oracle_kind, shape):
    """Near-degenerate fits in float64 (round-5 advice): a ribbon (extent 1 x 1e-7 x 1e-11; a 12-point neighbourhood spans ~6e-4 of it: the two smallest eigenvalues
    of the trace-1 covariance are ~1e-9 and ~1e-16, the eigenvector error of a Jacobi solver is off-diagonal mass / gap) and a sheet with 1e-9 of noise.
    The numpy SVD of the offset matrix resolves both directions; the GPU's eigen-solver of A^T A must agree to 1e-10 in 1 - |n . n0|
    wherever the smallest direction is separated at all (relative singular gap > 1e-7; src/point_cloud_normals.cpp:155-160)."""
    rng = np.random.default_rng(17)
    n, k = 20000, 12
    if shape == "ribbon":
        p = np.stack([rng.random(n), 1e-7 * rng.random(n), 1e-11 * rng.normal(size=n)], 1)
    else:
        p = np.stack([rng.random(n), rng.random(n), 1e-9 * rng.normal(size=n)], 1)
    rot = np.linalg.qr(rng.normal(size=(3, 3)))[0]              # (not axis-aligned: the rotations have work to do)
    p = np.ascontiguousarray(p @ rot.T)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(p, k)
    idx0, nrm0, gap = oracle.normals_knn(p, k, kind=oracle_kind)
    assert np.array_equal(idx, idx0)
    good = gap > 1e-7
    assert good.mean() > 0.9, good.mean()
    dot = np.abs(np.einsum("ij,ij->i", nrm, nrm0))
    assert np.all(1.0 - dot[good] <= 1e-10), float((1.0 - dot[good]).max())
    assert np.allclose(np.abs(nrm @ rot[:, 2]), 1.0, atol=1e-3 if shape == "ribbon" else 1e-6)       # and it is the geometric normal


def test_normals_knn_edge_cases(pcu):
    p = cloud(5, 50, np.float64)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(p, 60)           # more neighbours than points: every point is dropped
    assert len(idx) == 0 and nrm.shape == (0, 3)
    with pytest.raises(ValueError, match=r"Invalid number of neighbors \(0\) must be greater than 0"):
        pcu.estimate_point_cloud_normals_knn(p, 0)
    with pytest.raises(ValueError, match="Invalid point set with zero elements"):
        pcu.estimate_point_cloud_normals_knn(np.zeros((0, 3)), 5)
    with pytest.raises(ValueError, match="does not match the number of points"):
        pcu.estimate_point_cloud_normals_knn(p, 5, view_directions=np.ones((3, 3)))
    plane = np.concatenate([np.random.default_rng(1).random((4000, 2)), np.zeros((4000, 1))], 1)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(plane, 10)
    assert len(idx) == 4000 and np.allclose(np.abs(nrm[:, 2]), 1.0, atol=1e-12)
    import torch
    tidx, tn = pcu.estimate_point_cloud_normals_knn(torch.from_numpy(plane).cuda(), 10)
    assert tn.is_cuda and np.array_equal(tidx.cpu().numpy(), idx) and np.allclose(tn.cpu().numpy(), nrm)
    # the reference's own test body (tests/test_examples.py:427-442): shapes only
    v = np.load(os.path.join(GOLD, "bunny_v.npy")).astype(np.float64)
    _, n = pcu.estimate_point_cloud_normals_knn(v, 12)
    assert n.shape == v.shape


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("weight", ["constant", "rbf"])
def test_normals_ball_vs_oracle(pcu, dtype, weight):
    p = _surface(6000, dtype, seed=5)
    r = 0.012                                    # members: squared distance < 0.012, i.e. within ~0.11
    idx, nrm = pcu.estimate_point_cloud_normals_ball(p, r, weight_function=weight, min_pts_per_ball=5)
    idx0, nrm0, gap = oracle.normals_ball(p, r, min_pts_per_ball=5, weight_function=weight)
    _check(idx, nrm, idx0, nrm0, gap, signed=False)
    dirs = np.tile(np.array([[0.0, 0.0, 1.0]], dtype=dtype), (len(p), 1))
    idx, nrm = pcu.estimate_point_cloud_normals_ball(p, r, view_directions=dirs, drop_angle_threshold=np.deg2rad(30.0), weight_function=weight)
    idx0, nrm0, gap = oracle.normals_ball(p, r, view_directions=dirs, drop_angle_threshold=np.deg2rad(30.0), weight_function=weight)
    both = np.intersect1d(idx, idx0)
    assert len(np.setxor1d(idx, idx0)) <= 1e-3 * len(p) + 1
    a, b = np.searchsorted(idx, both), np.searchsorted(idx0, both)
    _check(both, nrm[a], both, nrm0[b], gap[b], signed=True)
    # sparse corner: a far point has too few neighbours and is dropped
    q = np.concatenate([p, np.array([[50.0, 50.0, 50.0]], dtype=dtype)])
    idx, _ = pcu.estimate_point_cloud_normals_ball(q, r)
    assert len(q) - 1 not in idx
    # max_pts_per_ball: a subset of each neighbourhood is fitted; on a smooth sheet the normal barely moves
    idx_s, nrm_s = pcu.estimate_point_cloud_normals_ball(p, r, max_pts_per_ball=20, weight_function=weight)
    idx_a, nrm_a = pcu.estimate_point_cloud_normals_ball(p, r, weight_function=weight)
    assert np.array_equal(idx_s, idx_a)
    assert np.median(np.abs(np.einsum("ij,ij->i", nrm_s.astype(np.float64), nrm_a.astype(np.float64)))) > 0.995
    with pytest.raises(ValueError, match="Invalid radius"):
        pcu.estimate_point_cloud_normals_ball(p, 0.0)
    with pytest.raises(ValueError, match="min_pts_per_ball"):
        pcu.estimate_point_cloud_normals_ball(p, 0.1, min_pts_per_ball=2)
    with pytest.raises(ValueError, match="weight_function"):
        pcu.estimate_point_cloud_normals_ball(p, 0.1, weight_function="gauss")


def test_normals_at_scale(pcu):
    """1M points, k = 16 (the 8f-1 use case): unit normals, reproducible, consistent with a direct fit on a random subset."""
    p = _surface(1_000_000, np.float32, seed=9)
    idx, nrm = pcu.estimate_point_cloud_normals_knn(p, 16)
    assert len(idx) == len(p)
    idx2, nrm2 = pcu.estimate_point_cloud_normals_knn(p, 16)
    assert np.array_equal(nrm, nrm2)
    sel = np.random.default_rng(0).choice(len(p), 300, replace=False)
    d, c = pcu.k_nearest_neighbors(p[sel], p, 16)
    a = (p[c] - p[sel][:, None, :]).astype(np.float64)
    _, s, vt = np.linalg.svd(a, full_matrices=False)
    good = (s[:, 1] - s[:, 2]) / s[:, 0] > 1e-2
    dot = np.abs(np.einsum("ij,ij->i", nrm[sel].astype(np.float64), vt[:, 2, :]))
    assert np.all(1 - dot[good] < 1e-6)
