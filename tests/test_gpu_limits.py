"""The documented size limit (include/pcu_hip.h: clouds of up to 2**27 - 16 rows), exercised (-m gpu): the reference has no limit
(Eigen::Index, src/point_cloud_distance.cpp:157-158), this library addresses records with 32-bit byte offsets -- exactly at the edge for
float64 (32-byte records x 2**27 = 2**32). No CPU oracle reaches these sizes in test time, so the checks are size-independent properties,
on the device: every returned distance reproduces bit for bit from its returned index, a sample of queries equals an exact brute force over
the whole dataset (same arithmetic, separate multiplies and adds), a self-query returns every row itself at distance 0, and the
two-sided operators agree with the rows."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the gfx950 path has no CPU fallback"
    return m


def _d2(torch, q, r):
    """((dx*dx) + (dy*dy)) + (dz*dz) with separate elementwise kernels: the contract's arithmetic (no FMA across torch ops)."""
    d = q - r
    d = d * d
    return (d[..., 0] + d[..., 1]) + d[..., 2]


def _brute(torch, q, r, chunk=1 << 20):
    """exact nearest neighbour of the few rows of q over all of r, first minimum by row"""
    best_d = torch.full((q.shape[0],), float("inf"), dtype=q.dtype, device=q.device)
    best_i = torch.zeros((q.shape[0],), dtype=torch.int64, device=q.device)
    for s in range(0, r.shape[0], chunk):
        d = _d2(torch, q[:, None, :], r[None, s:s + chunk, :])
        m, i = d.min(dim=1)
        take = m < best_d
        best_d = torch.where(take, m, best_d); best_i = torch.where(take, i + s, best_i)
    return best_d, best_i


def _check_rows(torch, pcu, q, r, sample=200, seed=0):
    d, c = pcu.k_nearest_neighbors(q, r, 1, squared_distances=True)
    assert int(c.min()) >= 0 and int(c.max()) < r.shape[0]
    assert torch.equal(_d2(torch, q, r[c]), d)                     # distance bits reproduce from the returned index
    sel = torch.from_numpy(np.random.default_rng(seed).choice(q.shape[0], sample, replace=False)).to(q.device)
    bd, bi = _brute(torch, q[sel], r)
    assert torch.equal(bd, d[sel]), "a sampled query's distance differs from the exact brute force"
    same = bi == c[sel]
    if not bool(same.all()):                                      # an exact tie: the other index must be at the same distance
        assert torch.equal(_d2(torch, q[sel][~same], r[bi[~same]]), d[sel][~same])
    return d, c


def test_64m_vs_64m_f32_k1(pcu):
    import torch
    n = 64 * 1024 * 1024
    g = torch.Generator(device="cuda"); g.manual_seed(64)
    q = torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float32)
    r = torch.rand((n, 3), generator=g, device="cuda", dtype=torch.float32)
    d, c = _check_rows(torch, pcu, q, r)
    d1, c1 = pcu.k_nearest_neighbors(q, r, 1, squared_distances=True)
    assert torch.equal(c, c1) and torch.equal(d, d1)              # run to run
    ds, cs = pcu.k_nearest_neighbors(r, r, 1)
    assert bool((ds == 0).all())
    moved = cs != torch.arange(n, device="cuda")
    if bool(moved.any()):                                         # duplicated points only
        assert torch.equal(r[cs[moved]], r[moved])
    ch = pcu.chamfer_distance(q, r)
    dq, _ = pcu.k_nearest_neighbors(q, r, 1); dr, _ = pcu.k_nearest_neighbors(r, q, 1)
    ref = float(dq.double().mean() + dr.double().mean())
    assert abs(float(ch) - ref) <= 1e-4 * ref
    h = pcu.hausdorff_distance(q, r)
    assert h == float(max(dq.max(), dr.max()))


def test_f64_dataset_at_the_row_limit(pcu):
    import torch
    m = (1 << 27) - 16                                            # the largest dataset the library takes
    g = torch.Generator(device="cuda"); g.manual_seed(27)
    r = torch.rand((m, 3), generator=g, device="cuda", dtype=torch.float64)
    q = torch.rand((1 << 20, 3), generator=g, device="cuda", dtype=torch.float64)
    q[:4096] = r[-4096:] + 1e-9                                   # queries whose neighbours are the LAST records of the clouds' streams
    d, c = _check_rows(torch, pcu, q, r, sample=100)
    assert int(c[:4096].min()) >= m - 4096 - 64                   # (their neighbours really are there)
    # k > 1 at the limit: rows sorted, first neighbour = the k = 1 answer
    d4, c4 = pcu.k_nearest_neighbors(q[:200000], r, 4, squared_distances=True)
    assert torch.equal(c4[:, 0], c[:200000]) and torch.equal(d4[:, 0], d[:200000]) and bool((d4[:, 1:] >= d4[:, :-1]).all())
    assert torch.equal(_d2(torch, q[:200000, None, :], r[c4]), d4)
    # one row more is refused, as documented
    with pytest.raises(ValueError, match="2\\^27"):
        pcu.k_nearest_neighbors(q[:10], torch.zeros((m + 1, 3), device="cuda", dtype=torch.float64), 1)
