"""GPU parity tests (-m gpu) of SURVEY.md 8f-4: Morton codes (bit-exact against the reference's own MortonCode64, compiled in
place into oracle/_ref/libpcu_ref_morton.so, or its pinned numpy restatement), voxel-grid downsampling (voxel means bit-identical
to the restated reference loop; rows compared after sorting both sides by voxel -- the reference's row order is its hash table's)
and duplicate removal (partition and order equal to the restatement; reference test bodies of tests/test_examples.py:84-99,
:444-525)."""
import numpy as np
import pytest

import oracle
from conftest import cloud

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0
    return m


@pytest.fixture(scope="module")
def mkind():
    return "ref" if oracle.have_ref_morton() else "port"


def test_morton_codes_bit_exact(pcu, mkind):
    rng = np.random.default_rng(0)
    for dt in (np.int32, np.int64):
        p = rng.integers(-(1 << 20), 1 << 20, (300000, 3)).astype(dt)
        p[:4] = [[0, 0, 0], [-1, -1, -1], [(1 << 20) - 1] * 3, [-(1 << 20)] * 3]
        codes = pcu.morton_encode(p)
        assert codes.dtype == np.uint64 and codes.shape == (300000,)
        assert np.array_equal(codes, oracle.morton_encode(p, mkind))
        back = pcu.morton_decode(codes)
        assert back.dtype == np.int32 and np.array_equal(back, p.astype(np.int32))
        assert np.array_equal(back, oracle.morton_decode(codes, mkind))
    q = rng.integers(-1000, 1000, (300000, 3)).astype(np.int32)
    c2 = pcu.morton_encode(q)
    assert np.array_equal(pcu.morton_add(codes, c2), oracle.morton_addsub(codes, c2, False, mkind))
    assert np.array_equal(pcu.morton_subtract(codes, c2), oracle.morton_addsub(codes, c2, True, mkind))
    small = rng.integers(-500, 500, (1000, 3)).astype(np.int32); other = rng.integers(-500, 500, (1000, 3)).astype(np.int32)
    assert np.array_equal(pcu.morton_decode(pcu.morton_add(pcu.morton_encode(small), pcu.morton_encode(other))), small + other)   # adds the vectors
    assert np.array_equal(pcu.morton_decode(pcu.morton_subtract(pcu.morton_encode(small), pcu.morton_encode(other))), small - other)
    assert np.array_equal(pcu.morton_decode(codes.astype(np.uint32)), oracle.morton_decode(codes.astype(np.uint32).astype(np.uint64), mkind))   # uint32 codes
    with pytest.raises(ValueError, match="empty array"):
        pcu.morton_encode(np.zeros((0, 3), np.int32))
    with pytest.raises(ValueError, match="invalid number of columns"):
        pcu.morton_encode(np.zeros((5, 2), np.int32))
    import torch
    tc = pcu.morton_encode(torch.from_numpy(p.astype(np.int32)).cuda())
    assert np.array_equal(tc.cpu().numpy().view(np.uint64), codes)


def test_morton_knn(pcu, mkind):
    """tests/test_examples.py:444-515 (big / small / tiny data) + the window against the reference's selection."""
    rng = np.random.default_rng(1)
    for num_pts, num_q, k in ((1000000, 10000, 7), (10, 10000, 7), (10, 10000, 15)):
        pts_int = (rng.random((num_pts, 3)) * 1000).astype(np.int32); qpts_int = (rng.random((num_q, 3)) * 1000).astype(np.int32)
        codes = pcu.morton_encode(pts_int)
        codes_sorted = codes[np.argsort(codes)]
        qcodes = pcu.morton_encode(qpts_int)
        nn_idx = pcu.morton_knn(codes_sorted, qcodes, k)
        assert nn_idx.shape == (num_q, min(k, num_pts)) and nn_idx.dtype == np.int64
        codes_sorted[nn_idx]
        win = pcu.morton_knn(codes_sorted, qcodes, k, sort_dist=False)
        assert np.array_equal(win, oracle.morton_knn_window(codes_sorted, qcodes, k, mkind))
        assert np.array_equal(np.sort(nn_idx, axis=1), win)                       # sort_dist reorders the same window ...
        d = np.linalg.norm((oracle.morton_decode(codes_sorted, mkind)[nn_idx] - oracle.morton_decode(qcodes, mkind)[:, None, :]).astype(np.float64), axis=-1)
        assert np.all(np.diff(d, axis=1) >= 0)                                    # ... by ascending distance to the query
    with pytest.raises(ValueError, match="k must be greater than 0"):
        pcu.morton_knn(codes_sorted, qcodes, 0)


def _sorted_rows(v, *rest):
    o = np.lexsort((v[:, 2], v[:, 1], v[:, 0]))
    return (v[o],) + tuple(r[o] for r in rest)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_voxel_downsample_vs_oracle(pcu, dtype):
    rng = np.random.default_rng(2)
    p = cloud(3, 40000, dtype)
    nrm = rng.normal(size=(40000, 3)).astype(np.float32 if dtype == np.float64 else np.float64)      # attribute dtype differs from the points'
    col = rng.random((40000, 4)).astype(dtype)
    vs = 1.0 / 16.0
    v, a0, a1 = pcu.downsample_point_cloud_on_voxel_grid(vs, p, nrm, col)
    assert v.dtype == dtype and a0.dtype == nrm.dtype and a1.dtype == dtype and a0.shape == (len(v), 3) and a1.shape == (len(v), 4)
    mb = np.min(p, axis=0) - np.array([vs] * 3) * 0.5
    v0, n0 = oracle.voxel_downsample(p, nrm, [vs] * 3, mb)
    _, c0 = oracle.voxel_downsample(p, col, [vs] * 3, mb)
    assert np.array_equal(v, v0) and np.array_equal(a0, n0) and np.array_equal(a1, c0)       # same order (voxel index), bit-identical means
    # anisotropic voxels, explicit bounds, min_points_per_voxel, no attributes
    v = pcu.downsample_point_cloud_on_voxel_grid((0.1, 0.05, 0.2), p, min_bound=(-0.3, -0.2, -0.1), max_bound=(2, 2, 2), min_points_per_voxel=12)
    v0, _ = oracle.voxel_downsample(p, None, (0.1, 0.05, 0.2), (-0.3, -0.2, -0.1), min_points_per_voxel=12)
    assert 0 < len(v) < 2000 and np.array_equal(v, v0)
    # every point its own voxel / all points in one voxel
    assert len(pcu.downsample_point_cloud_on_voxel_grid(1e-6, p[:5000])) == len(np.unique(np.floor((p[:5000] - (np.min(p[:5000], 0) - 5e-7)) / dtype(1e-6)), axis=0))
    one = pcu.downsample_point_cloud_on_voxel_grid(10.0, p)
    acc = np.zeros(3, dtype)
    for row in p:
        acc = acc + row
    assert one.shape == (1, 3) and np.array_equal(one[0], acc / dtype(len(p)))                # the reference's sequential sum, bit for bit
    with pytest.raises(ValueError, match="max_bound must be greater than min_bound"):
        pcu.downsample_point_cloud_on_voxel_grid(0.1, p, min_bound=(0, 0, 0), max_bound=(1, 0, 1))
    with pytest.raises(ValueError, match="Voxel size is negative"):
        pcu.downsample_point_cloud_on_voxel_grid(-0.1, p, min_bound=(0, 0, 0), max_bound=(1, 1, 1))
    with pytest.raises(ValueError, match="same first dimension"):
        pcu.downsample_point_cloud_on_voxel_grid(0.1, p, col[:10])


def test_voxel_downsample_large(pcu):
    """1M points, 1/128 voxels (the reference test's size): counts and means against numpy's grouping."""
    p = cloud(5, 1_000_000, np.float64)
    vs = 1.0 / 128.0
    v = pcu.downsample_point_cloud_on_voxel_grid(vs, p)
    key = np.floor((p - (np.min(p, 0) - vs * 0.5)) / vs).astype(np.int64)
    lin = (key[:, 0] << 42) | (key[:, 1] << 21) | key[:, 2]
    order = np.argsort(lin, kind="stable"); ls = lin[order]
    heads = np.flatnonzero(np.r_[True, ls[1:] != ls[:-1]])
    assert len(v) == len(heads)
    sums = np.add.reduceat(p[order], heads, axis=0); cnt = np.diff(np.r_[heads, len(p)])
    assert np.allclose(v, sums / cnt[:, None], rtol=1e-12, atol=1e-15)            # (reduceat's summation order is numpy's, not the sequential one)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_deduplicate_point_cloud(pcu, dtype):
    rng = np.random.default_rng(4)
    base = cloud(6, 5000, dtype)
    v = np.concatenate([base, base[rng.integers(0, 5000, 3000)], base[:10] + dtype(1e-13 if dtype == np.float64 else 1e-9)])
    v = v[rng.permutation(len(v))]
    for eps in (1e-11 if dtype == np.float64 else 1e-6, 0.0, 0.05):
        v2, i_v_to_v2, i_v2_to_v = pcu.deduplicate_point_cloud(v, eps, return_index=True)
        assert i_v_to_v2.dtype == np.int32 and i_v2_to_v.dtype == np.int32 and v2.dtype == dtype
        x0, svi0, svj0 = oracle.deduplicate_point_cloud(v, eps)
        assert np.array_equal(v2, x0) and np.array_equal(i_v_to_v2, svi0) and np.array_equal(i_v2_to_v, svj0)
        assert np.array_equal(v[i_v_to_v2], v2) and len(v2) < len(v)
        if eps <= 1e-6:
            assert np.allclose(v2[i_v2_to_v], v, atol=1e-6)                        # tests/test_examples.py:517-520 (exact there: its data has exact duplicates)
        assert np.array_equal(pcu.deduplicate_point_cloud(v, eps, return_index=False), v2)
    exact = np.concatenate([base, base])
    v2, a, b = pcu.deduplicate_point_cloud(exact, 1e-11)
    assert len(v2) == 5000 and np.array_equal(v2[b], exact) and np.array_equal(exact[a], v2)
    with pytest.raises(ValueError, match="Only 3D inputs are supported"):
        pcu.deduplicate_point_cloud(np.zeros((4, 2), dtype), 0.1)
