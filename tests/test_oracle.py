"""CPU tests: the oracle (plain-C restatement) against the golden vectors generated from the reference's own
nanoflann, against oracle/_ref when present, and against exact brute force."""
import glob
import os

import numpy as np
import pytest

import oracle
from conftest import cloud

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
CASES = sorted(p for p in glob.glob(os.path.join(GOLD, "*.npz")) if not p.endswith(("metrics.npz", "sinkhorn.npz", "config1.npz")))


@pytest.mark.parametrize("path", CASES, ids=[os.path.basename(p)[:-4] for p in CASES])
def test_port_matches_golden(path):
    g = np.load(path)
    d, c = oracle.knn(g["q"], g["r"], int(g["k"]), squared_distances=bool(g["squared"]), kind="port")
    assert np.array_equal(c, g["c"])
    assert np.array_equal(d.view(np.uint8), g["d"].view(np.uint8))   # bit-exact distances


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("leaf", [1, 10, 37])
def test_port_matches_ref_when_present(dtype, leaf):
    if not oracle.have_ref():
        pytest.skip("oracle/_ref not built on this box")
    q, r = cloud(1, 3000, dtype), cloud(2, 2500, dtype)
    r = np.concatenate([r, r[:500]])      # duplicates: tie order depends on the tree
    for k in (1, 6):
        d0, c0 = oracle.knn(q, r, k, max_points_per_leaf=leaf, kind="ref")
        d1, c1 = oracle.knn(q, r, k, max_points_per_leaf=leaf, kind="port")
        assert np.array_equal(c0, c1) and np.array_equal(d0, d1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_port_matches_brute_without_ties(dtype):
    q, r = cloud(3, 1500, dtype), cloud(4, 2000, dtype)
    for k in (1, 9):
        d0, c0, tie = oracle.brute_knn_with_ties(q, r, k)
        d1, c1 = oracle.knn(q, r, k, kind="port")
        ok = ~tie
        assert ok.sum() > 1400
        assert np.array_equal(c0[ok], c1[ok]) and np.array_equal(d0[ok], d1[ok])
        assert np.array_equal(d0, d1)         # distances are unique even under ties


def test_metrics_golden():
    g = np.load(os.path.join(GOLD, "metrics.npz"))
    for tag in ("f32", "f64"):
        a, b = g[f"a_{tag}"], g[f"b_{tag}"]
        h = oracle.hausdorff_distance(a, b, return_index=True, kind="port")
        assert list(g[f"hausdorff_{tag}"]) == [h[0], h[1], h[2]]
        assert tuple(g[f"one_sided_ab_{tag}"]) == oracle.one_sided_hausdorff_distance(a, b, kind="port")
        assert tuple(g[f"one_sided_ba_sq_{tag}"]) == oracle.one_sided_hausdorff_distance(b, a, squared_distances=True, kind="port")
        ch, cxy, cyx = oracle.chamfer_distance(a, b, return_index=True, kind="port")
        assert float(ch) == g[f"chamfer_{tag}"][0]
        assert np.array_equal(cxy, g[f"cxy_{tag}"]) and np.array_equal(cyx, g[f"cyx_{tag}"])


def test_nonfinite_metrics_golden():
    """tests/golden/nf_*_metrics.npz (generated from the reference's own nanoflann): the C port and the restated Python tails give the
    same answers on clouds with non-finite rows -- unmatched source rows are -1 / -1.0, Chamfer gathers through index -1."""
    import warnings
    for tag in ("f32", "f64"):
        g = np.load(os.path.join(GOLD, f"nf_{tag}_metrics.npz"))
        same = lambda a, b: np.array_equal(np.asarray(a, np.float64), np.asarray(b, np.float64), equal_nan=True)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for name in ("inf", "mixed", "last"):
                x, y = g[f"{name}_x"], g[f"{name}_y"]
                assert same(oracle.one_sided_hausdorff_distance(x, y, kind="port"), g[f"{name}_os_xy"])
                assert same(oracle.one_sided_hausdorff_distance(y, x, kind="port"), g[f"{name}_os_yx"])
                assert same(oracle.hausdorff_distance(x, y, True, kind="port"), g[f"{name}_h"])
                ch, cxy, cyx = oracle.chamfer_distance(x, y, return_index=True, kind="port")
                assert np.array_equal(cxy, g[f"{name}_cxy"]) and np.array_equal(cyx, g[f"{name}_cyx"])
                assert same([oracle.chamfer_distance(x, y, p_norm=p, kind="port") for p in (2, 1, np.inf, -np.inf, 0, 3)], g[f"{name}_ch"])
            assert same(oracle.one_sided_hausdorff_distance(g["nan_x"], g["nan_y"], kind="port"), g["nan_os_xy"])
            assert np.isnan(oracle.chamfer_distance(g["nan_x"], g["nan_y"], kind="port"))


def test_reference_test_knn_body_on_oracle():
    """tests/test_examples.py:349-396 of the reference, with the oracle standing in for pcu."""
    rng = np.random.default_rng(0)
    for _ in range(3):
        a, b = rng.random((1000, 3)), rng.random((500, 3))
        k = int(rng.integers(10)) + 1
        d, c = oracle.k_nearest_neighbors(a, b, k)
        assert d.shape == ((1000, k) if k > 1 else (1000,))
        if k == 1:
            d, c = d[:, None], c[:, None]
        assert np.all(np.abs(np.linalg.norm(a[:, None, :] - b[c], axis=-1) - d) < 1e-5)
    with pytest.raises(ValueError):
        oracle.k_nearest_neighbors(a, b, 0)


def test_oracle_normals_on_planes():
    """The numpy restatement of the normals path: exact planes give the plane normal, the view direction fixes the sign and
    filters (src/point_cloud_normals.cpp:115-173)."""
    rng = np.random.default_rng(2)
    xy = rng.random((500, 2))
    p = np.concatenate([xy, (0.5 * xy[:, :1] + 0.25 * xy[:, 1:2])], 1)           # plane z = 0.5 x + 0.25 y
    n0 = np.array([-0.5, -0.25, 1.0]); n0 /= np.linalg.norm(n0)
    idx, nrm, gap = oracle.normals_knn(p, 8)
    assert len(idx) == 500 and np.allclose(np.abs(nrm @ n0), 1.0, atol=1e-9)
    dirs = np.tile(-n0, (500, 1))
    idx, nrm, _ = oracle.normals_knn(p, 8, view_directions=dirs)
    assert len(idx) == 500 and np.allclose(nrm @ n0, -1.0, atol=1e-9)
    idx, _, _ = oracle.normals_knn(p, 8, view_directions=np.tile(np.array([1.0, 0, 0]), (500, 1)), drop_angle_threshold=np.deg2rad(20))
    assert len(idx) == 0                                                       # the normal is ~64 degrees off the x axis
    idx, nrm, _ = oracle.normals_ball(p[:200], 0.05)
    assert np.allclose(np.abs(nrm @ n0), 1.0, atol=1e-9)
    assert len(oracle.normals_knn(p[:5], 9)[0]) == 0                           # fewer points than neighbours: dropped


def test_oracle_morton_port_pinned_to_reference():
    """The numpy restatement of MortonCode64 against the reference's own class (src/common/morton_code.cpp compiled in place
    into oracle/_ref/libpcu_ref_morton.so) -- skipped where /root/reference never existed."""
    if not oracle.have_ref_morton():
        pytest.skip("oracle/_ref/libpcu_ref_morton.so not built (no /root/reference)")
    rng = np.random.default_rng(0)
    p = rng.integers(-(1 << 20), 1 << 20, (50000, 3)).astype(np.int32)
    p[:4] = [[0, 0, 0], [-1, -1, -1], [(1 << 20) - 1] * 3, [-(1 << 20)] * 3]
    c = oracle.morton_encode(p, "ref")
    assert np.array_equal(c, oracle.morton_encode(p, "port"))
    assert np.array_equal(oracle.morton_decode(c, "ref"), p) and np.array_equal(oracle.morton_decode(c, "port"), p)
    c2 = oracle.morton_encode(rng.integers(-2000, 2000, (50000, 3)).astype(np.int32), "ref")
    for sub in (False, True):
        assert np.array_equal(oracle.morton_addsub(c, c2, sub, "ref"), oracle.morton_addsub(c, c2, sub, "port"))
    cs = np.sort(c)
    for k in (1, 7, 16):
        assert np.array_equal(oracle.morton_knn_window(cs, c2[:2000], k, "ref"), oracle.morton_knn_window(cs, c2[:2000], k, "port"))
    assert np.array_equal(oracle.morton_knn_window(cs[:10], c2[:100], 15, "ref"), oracle.morton_knn_window(cs[:10], c2[:100], 15, "port"))


def test_oracle_voxel_and_dedup_restatements():
    rng = np.random.default_rng(1)
    p = rng.random((2000, 3)).astype(np.float32)
    v, a = oracle.voxel_downsample(p, p * 2, [0.25] * 3, [0, 0, 0])
    assert len(v) == 64 and np.allclose(a, v * 2, rtol=1e-6)
    key = np.floor(p / np.float32(0.25)).astype(int)
    m = np.all(key == 0, axis=1)
    assert np.allclose(v[0], p[m].mean(0), rtol=1e-5)
    x = np.concatenate([p[:100], p[:100]])
    u, svi, svj = oracle.deduplicate_point_cloud(x, 1e-7)
    assert len(u) == 100 and np.array_equal(x[svi], u) and np.array_equal(u[svj], x) and np.all(svi < 100)
    assert np.array_equal(oracle.deduplicate_point_cloud(np.array([[0.5, 1.5, -0.5], [2.5, -1.5, 0.49999997]], np.float32), 1.0)[0],
                          np.array([[0.5, 1.5, -0.5], [2.5, -1.5, 0.49999997]], np.float32)[[0, 1]])


def test_oracle_sinkhorn_pinned_to_reference_module_and_golden():
    """The numpy restatement of point_cloud_utils/_sinkhorn.py against tests/golden/sinkhorn.npz (generated from the reference's
    own module) and, where /root/reference exists, against that module itself on fresh inputs: bit-equal (same numpy calls)."""
    import os
    g = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sinkhorn.npz"))
    for tag in ("f32", "f64"):
        a, b = g[f"a_{tag}"], g[f"b_{tag}"]
        for p in (None, 1, np.inf, 3):
            assert np.array_equal(oracle.pairwise_distances(a, b, p), g[f"M_{tag}_p{p}"])
        dt = a.dtype.type
        P, _ = oracle.sinkhorn(np.full(96, 1.0 / 96, dt), np.full(80, 1.0 / 80, dt), g[f"M_{tag}_pNone"], eps=1e-2, max_iters=60)
        assert np.array_equal(P, g[f"P_{tag}"])
        Pb, _ = oracle.sinkhorn(g[f"wab_{tag}"], g[f"wbb_{tag}"], g[f"Mb_{tag}"], eps=5e-2, max_iters=100, stop_thresh=1e-4)
        assert np.array_equal(Pb, g[f"Pb_{tag}"])
    emd, P = oracle.earth_movers_distance(g["emd_p"], g["emd_q"], eps=1e-2)
    assert emd == g["emd"] and np.array_equal(P, g["emd_P"])
    ref = oracle.reference_sinkhorn_module()
    if ref is not None:
        rng = np.random.default_rng(5)
        x = rng.random((2, 30, 4)).astype(np.float32); y = rng.random((2, 25, 4)).astype(np.float32)
        M = ref.pairwise_distances(x, y, 2)
        assert np.array_equal(M, oracle.pairwise_distances(x, y, 2))
        wa = np.full((2, 30), 1 / 30, np.float32); wb = np.full((2, 25), 1 / 25, np.float32)
        assert np.array_equal(ref.sinkhorn(wa, wb, M, 1e-2), oracle.sinkhorn(wa, wb, M, 1e-2)[0])


def test_config1_cpu():
    """BASELINE config 1 ("chamfer_distance on two 10k-point fp64 random clouds via reference nanoflann CPU path; plumbing, no GPU"):
    the reference's nanoflann (oracle/_ref, where built) and the restatement agree bit for bit on value and correspondences, and a
    golden scalar generated from the reference (tests/golden/make_golden.py) pins both."""
    x, y = cloud(1000, 10_000, np.float64), cloud(1001, 10_000, np.float64)
    ch, cxy, cyx = oracle.chamfer_distance(x, y, return_index=True, kind="port")
    g = np.load(os.path.join(GOLD, "config1.npz"))
    assert float(ch) == float(g["chamfer"]) and np.array_equal(cxy, g["cxy"]) and np.array_equal(cyx, g["cyx"])
    if oracle.have_ref():
        ch1, cxy1, cyx1 = oracle.chamfer_distance(x, y, return_index=True, kind="ref")
        assert float(ch1) == float(ch) and np.array_equal(cxy1, cxy) and np.array_equal(cyx1, cyx)
