"""CPU tests of host-side logic that needs no GPU: numpy-style promotion / broadcasting of pairwise_distances' inputs
(point_cloud_utils_amd/_sinkhorn.py; reference: the single numpy expression of point_cloud_utils/_sinkhorn.py:30-32), the sharding of
pair batches over ranks, and the shape errors pairwise_distances raises before it touches the device."""
import itertools

import numpy as np
import pytest


DTYPES = [np.float16, np.float32, np.float64, np.int8, np.int32, np.int64, np.uint8, np.bool_]


@pytest.mark.parametrize("da,db", list(itertools.product(DTYPES, DTYPES)))
def test_promotion_follows_numpy_subtraction(da, db):
    """The dtype of the distance matrix is numpy's dtype of `norm(a[..., None, :] - b[..., None, :, :])` for every pair of input dtypes
    the reference accepts -- with one documented difference: float16 pairs are computed in float32 (no half kernels; the reference
    returns float16)."""
    from point_cloud_utils_amd._sinkhorn import _promote_pair
    a = np.ones((2, 3), dtype=da); b = np.ones((4, 3), dtype=db)
    pa, pb = _promote_pair(a, b)
    assert pa.dtype == pb.dtype and pa.dtype in (np.float32, np.float64)
    if da == np.bool_ and db == np.bool_:
        return                                    # numpy refuses bool - bool; we compute it in float64
    want = np.linalg.norm(a[:, None, :] - b[None, :, :], axis=-1).dtype
    if want == np.float16:
        want = np.dtype(np.float32)
    assert pa.dtype == want, (da, db, pa.dtype, want)
    assert np.array_equal(pa, a.astype(pa.dtype)) and np.array_equal(pb, b.astype(pb.dtype))


@pytest.mark.parametrize("sa,sb", [((1, 5, 3), (4, 6, 3)), ((4, 5, 1), (4, 6, 3)), ((4, 5, 3), (1, 6, 1)), ((2, 5, 3), (2, 6, 3))])
def test_expand_matches_numpy_broadcasting(sa, sb):
    from point_cloud_utils_amd._sinkhorn import _expand
    rng = np.random.default_rng(0)
    a, b = rng.random(sa), rng.random(sb)
    nb, d = max(sa[0], sb[0]), max(sa[2], sb[2])
    ea, eb = _expand(a, (nb, sa[1], d)), _expand(b, (nb, sb[1], d))
    ref = np.linalg.norm(a[:, :, None, :] - b[:, None, :, :], axis=-1)
    got = np.linalg.norm(np.asarray(ea)[:, :, None, :] - np.asarray(eb)[:, None, :, :], axis=-1)
    assert got.shape == ref.shape == (nb, sa[1], sb[1]) and np.array_equal(got, ref)


def test_pairwise_shape_errors_come_before_the_device():
    import point_cloud_utils_amd as pcu
    with pytest.raises(ValueError, match="Invalid shape"):
        pcu.pairwise_distances(np.zeros((3,)), np.zeros((3,)))
    with pytest.raises(ValueError, match="broadcast"):
        pcu.pairwise_distances(np.zeros((2, 4, 3)), np.zeros((3, 4, 3)))
    with pytest.raises(ValueError, match="broadcast"):
        pcu.pairwise_distances(np.zeros((4, 3)), np.zeros((4, 2)))


@pytest.mark.parametrize("total,world", [(1, 1), (7, 2), (32, 8), (5, 8), (0, 4)])
def test_shard_pairs_partitions_the_batch(total, world):
    """Pair p belongs to rank p mod world: the shards are disjoint, cover the batch, and differ in size by at most one."""
    from point_cloud_utils_amd import batched
    shards = [list(batched.shard_pairs(total, r, world)) for r in range(world)]
    flat = sorted(p for s in shards for p in s)
    assert flat == list(range(total))
    assert all(p % world == r for r, s in enumerate(shards) for p in s)
    assert max(map(len, shards)) - min(map(len, shards)) <= 1
