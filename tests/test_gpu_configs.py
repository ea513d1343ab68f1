"""GPU parity tests (-m gpu) at the FULL sizes of BASELINE.json's configs, against the reference's own nanoflann
(oracle kind "ref" where oracle/_ref was built, the pinned C restatement otherwise), through the public API = the C ABI:

  headline  chamfer_distance 1M-vs-1M f32: value within 1e-4, both correspondence arrays bit-exact
  C2        k_nearest_neighbors k=1, 1M-vs-1M f32: indices + distance bits
  C3        k_nearest_neighbors k=16, 4M-vs-4M f32: indices + distance bits (incl. the ~40 exact ties -> kd-tree order)
  C4        hausdorff_distance on full-size 262,144-point pairs through the batch entry point
  C5        chamfer_distance f64, bunny (2,885 vertices) vs 1M area-weighted mesh samples: both index arrays exact
plus k > 127 (the reference accepts any k > 0) and the batch entry points against the single-pair calls."""
import os

import numpy as np
import pytest

import oracle
from conftest import cloud, mesh_samples

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the gfx950 path has no CPU fallback"
    return m


def test_headline_chamfer_1m_vs_reference(pcu, oracle_kind):
    n = 1_000_000
    x, y = cloud(1000, n, np.float32), cloud(1001, n, np.float32)
    ch0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=oracle_kind)
    ch = pcu.chamfer_distance(x, y)                                   # the benchmarked call shape (fused epilogue, no rows)
    assert type(ch) == np.float32
    assert abs(float(ch) - float(ch0)) <= 1e-4 * float(ch0)
    ch2, cxy, cyx = pcu.chamfer_distance(x, y, return_index=True)      # row-based path: correspondences
    assert np.array_equal(cxy, cxy0) and np.array_equal(cyx, cyx0)
    assert abs(float(ch2) - float(ch0)) <= 1e-4 * float(ch0)
    assert float(ch) == float(pcu.chamfer_distance(x, y))             # reproducible run to run (exact accumulation of the stragglers)
    # device-resident inputs (what bench.py times) give the same value
    import torch
    tx, ty = torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()
    assert float(pcu.chamfer_distance(tx, ty)) == float(ch)
    # Hausdorff on the same pair (fused arg-max) against the reference
    assert pcu.hausdorff_distance(x, y, return_index=True) == oracle.hausdorff_distance(x, y, return_index=True, kind=oracle_kind)
    assert pcu.one_sided_hausdorff_distance(x, y) == oracle.one_sided_hausdorff_distance(x, y, kind=oracle_kind)


def test_config2_knn_k1_1m(pcu, oracle_kind):
    n = 1_000_000
    q, r = cloud(1000, n, np.float32), cloud(1001, n, np.float32)
    d, c = pcu.k_nearest_neighbors(q, r, 1)
    d0, c0 = oracle.k_nearest_neighbors(q, r, 1, kind=oracle_kind)
    assert np.array_equal(c, c0), pcu.last_stats()
    assert np.array_equal(d.view(np.uint32), d0.view(np.uint32))


def test_config3_knn_k16_4m(pcu, oracle_kind):
    n = 4_000_000
    q, r = cloud(1000, n, np.float32), cloud(1001, n, np.float32)
    d, c = pcu.k_nearest_neighbors(q, r, 16)
    st = pcu.last_stats()
    d0, c0 = oracle.k_nearest_neighbors(q, r, 16, kind=oracle_kind)
    assert np.array_equal(c, c0), (int((c != c0).any(1).sum()), st)
    assert np.array_equal(d.view(np.uint32), d0.view(np.uint32))
    assert st["n_tie_true"] > 0, st          # the exact ties of this config went through the kd-tree order


def test_config4_hausdorff_full_size_pairs(pcu, oracle_kind):
    from point_cloud_utils_amd import batched
    n, npairs = 262144, 6

    def get_pair(p):
        return cloud(1000 + 2 * p, n, np.float32), cloud(1001 + 2 * p, n, np.float32)

    hd = batched.batched_hausdorff(get_pair, npairs)
    for p in range(npairs):
        x, y = get_pair(p)
        assert tuple(hd[p]) == pcu.hausdorff_distance(x, y, return_index=True)          # batch entry point == single-pair call
        if p < 3:
            assert tuple(hd[p]) == oracle.hausdorff_distance(x, y, return_index=True, kind=oracle_kind)
    # device-resident pairs (the benchmarked mode)
    import torch
    pairs = [tuple(torch.from_numpy(a).cuda() for a in get_pair(p)) for p in range(npairs)]
    hd2 = batched.batched_hausdorff(lambda p: pairs[p], npairs, workers=3)
    assert np.array_equal(hd, hd2)
    ch = batched.batched_chamfer(lambda p: pairs[p], npairs)
    for p in range(2):
        x, y = get_pair(p)
        assert ch[p] == float(pcu.chamfer_distance(x, y))
        assert abs(ch[p] - float(oracle.chamfer_distance(x, y, kind=oracle_kind))) <= 1e-4 * ch[p]


def test_config5_bunny_vs_mesh_samples_f64(pcu, oracle_kind):
    bunny = np.load(os.path.join(GOLD, "bunny_v.npy")).astype(np.float64)
    f = np.load(os.path.join(GOLD, "bunny_f.npy"))
    s = mesh_samples(bunny, f, 1_000_000, seed=5)
    ch, cxy, cyx = pcu.chamfer_distance(bunny, s, return_index=True)
    ch0, cxy0, cyx0 = oracle.chamfer_distance(bunny, s, return_index=True, kind=oracle_kind)
    assert np.array_equal(cxy, cxy0) and np.array_equal(cyx, cyx0)
    assert abs(float(ch) - float(ch0)) <= 1e-6 * float(ch0)
    assert abs(float(pcu.chamfer_distance(bunny, s)) - float(ch0)) <= 1e-6 * float(ch0)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_k_beyond_the_grid_search(pcu, oracle_kind, dtype):
    """k > 127: the reference answers any k > 0 (src/point_cloud_distance.cpp:133-135), so does the GPU path (the reference's
    kd-tree traversal with a k-slot result set): indices and distance bits equal, -1 padding when k > m, ties included."""
    for n, m, k in ((3000, 5000, 128), (2000, 40000, 200), (300, 900, 1000), (50, 20000, 9000)):
        q, r = cloud(41, n, dtype), cloud(42, m, dtype)
        d, c = pcu.k_nearest_neighbors(q, r, k)
        d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=oracle_kind)
        assert np.array_equal(c, c0), (n, m, k)
        assert np.array_equal(d, d0), (n, m, k)
    base = cloud(43, 1500, dtype)
    r = np.concatenate([base, base])                                   # every distance tied: the kd-tree order decides
    d, c = pcu.k_nearest_neighbors(base[:400], r, 300, squared_distances=True)
    d0, c0 = oracle.k_nearest_neighbors(base[:400], r, 300, squared_distances=True, kind=oracle_kind)
    assert np.array_equal(c, c0) and np.array_equal(d, d0)
    with pcu.DatasetIndex(r) as index:
        d, c = index.k_nearest_neighbors(base[:100], 130)
        d0, c0 = oracle.k_nearest_neighbors(base[:100], r, 130, kind=oracle_kind)
        assert np.array_equal(c, c0) and np.array_equal(d, d0)


def test_fused_calls_fall_back_when_rows_are_needed(pcu, oracle_kind):
    """The fused epilogues (no result rows) must give way to the row-based path whenever they cannot stand: far-apart
    clouds (queries uncertified after radius 2), unbalanced dataset grids (refit), Hausdorff arg-max rows with tied
    neighbours. Same results as the reference either way."""
    rng = np.random.default_rng(77)
    n = 60000
    cases = {
        "far": (cloud(1, n, np.float32, scale=0.2), cloud(2, n, np.float32, scale=0.2, offset=3.0)),
        "cluster": (np.concatenate([rng.random((n, 3)), rng.normal(0.5, 0.001, (n, 3))]).astype(np.float32), cloud(3, n, np.float32)),
        "dups": (np.repeat(cloud(4, n // 4, np.float32), 4, axis=0), cloud(5, n, np.float32)),
    }
    for name, (x, y) in cases.items():
        for a, b in ((x, y), (y, x)):
            assert pcu.hausdorff_distance(a, b, return_index=True) == oracle.hausdorff_distance(a, b, return_index=True, kind=oracle_kind), name
            assert pcu.one_sided_hausdorff_distance(a, b) == oracle.one_sided_hausdorff_distance(a, b, kind=oracle_kind), name
            ch, ch0 = pcu.chamfer_distance(a, b), oracle.chamfer_distance(a, b, kind=oracle_kind)
            assert abs(float(ch) - float(ch0)) <= 1e-4 * float(ch0), name


def test_squeeze_of_singleton_shapes(pcu):
    q, r = cloud(1, 1, np.float32), cloud(2, 50, np.float32)
    d, c = pcu.k_nearest_neighbors(q, r, 1)
    assert d.shape == () and c.shape == () and c.dtype == np.int64
    d, c = pcu.k_nearest_neighbors(q, r, 3)
    assert d.shape == (3,) and c.shape == (3,)


SWITCHES = ["PCU_HIP_TWO_PASS=1", "PCU_HIP_NO_FUSE=1", "PCU_HIP_NO_FUSED_CONTINUE=1", "PCU_HIP_NO_SPIN=1", "PCU_HIP_NO_GRAPH=1",
            "PCU_HIP_NO_KD_SPEC=1", "PCU_HIP_NO_RESCALE=1", "PCU_HIP_KD_FULL=1", "PCU_HIP_NO_K1=1", "PCU_HIP_INDEX=atomic",
            "PCU_HIP_SINK_TWO_PASS=1", "PCU_HIP_DEBUG_SKEW=1", "PCU_HIP_NO_ESCALATE=1", "PCU_HIP_GRID_KERNEL=1", "PCU_HIP_REFIT_BASE=1",
            "PCU_HIP_PROF_BUILD=1", "PCU_HIP_PROF_KD=1",
            # round 4: first form of the one-pass build, Pt4 records for k = 1, cell-order rows + restore for k < 4, wave pass up front in
            # fused calls, level passes all the way down / one workgroup for the whole tie-order tree, refits one direction at a time
            "PCU_HIP_BUILD_V1=1", "PCU_HIP_NO_LEAN=1", "PCU_HIP_ROW_OUT_MIN_K=4", "PCU_HIP_FUSED_WAVE=1", "PCU_HIP_KD_FINISH_MAX=0",
            "PCU_HIP_KD_FINISH_MAX=1000000000", "PCU_HIP_NO_SKEW_OVERLAP=1", "PCU_HIP_SPEC_PRIORITY=1", "PCU_HIP_PROF_BUILD2=1",
            "PCU_HIP_HOST_PROF=1", "PCU_HIP_DEBUG_POISON=255", "PCU_HIP_NO_WAVE_MERGE=1",
            # round 5: k > 1 lane pass without the run list (k_search everywhere), the round-4 small-cloud thresholds (wave-per-query below
            # 16384 queries, atomic build below 32768 points), no SIGINT watch
            "PCU_HIP_KSEARCH_V1=1", "PCU_HIP_WAVE_ONLY_BELOW=16384", "PCU_HIP_BUCKET_MIN=32768", "PCU_HIP_NO_SIGINT=1",
            # round 6: every cloud of a two-sided call on its own grid again; the LDS-staged k = 1 pass over the shared grid (search_brick.h)
            # ... every fused call laid out from its own sample; 2048-point scatter blocks
            "PCU_HIP_NO_SHARED_GRID=1", "PCU_HIP_BRICK=1", "PCU_HIP_NO_GEO_CACHE=1", "PCU_HIP_BUILD_PTS=2",
            # ... Hausdorff's lane pass tracking every query's winner (round 5) instead of the value-only program + one resolution in the tail
            "PCU_HIP_NO_MAXVAL=1"]


@pytest.mark.gpu
@pytest.mark.parametrize("switch", SWITCHES)
def test_every_environment_switch_keeps_the_results(pcu, oracle_kind, tmp_path, switch):
    """Every environment switch the library reads (fallback pipelines and diagnostics; each is read once per process, hence a child
    process per switch) must leave the results untouched: a uniform pair with duplicated rows (exact ties -> tie-order resolver),
    a pair with a tight cluster and a far outlier (give-ups, refit, stragglers) and a small dense Sinkhorn problem, against the
    oracle / this process's default run."""
    import subprocess
    import sys
    rng = np.random.default_rng(91)
    x = rng.random((70_000, 3), dtype=np.float32); y = rng.random((60_000, 3), dtype=np.float32)
    y[:400] = x[:400]; y[400:800] = y[800:1200]
    cx = np.concatenate([rng.random((40_000, 3)), rng.normal(0.5, 0.002, (8_000, 3))]).astype(np.float32); cx[0] = [40.0, -30.0, 20.0]
    cy = np.concatenate([rng.random((30_000, 3)), rng.normal(0.5, 0.002, (9_000, 3))]).astype(np.float32)
    fx = rng.random((30_000, 3), dtype=np.float32)                                       # queries between two far-apart dataset clusters:
    fy = np.concatenate([rng.random((20_000, 3)) * 0.1, rng.random((20_000, 3)) * 0.1 + 0.9]).astype(np.float32)   # stragglers far from everything
    np.savez(tmp_path / "in.npz", x=x, y=y, cx=cx, cy=cy, fx=fx, fy=fy)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import point_cloud_utils_amd as pcu\n"
        "g = np.load(%r); x, y, cx, cy, fx, fy = g['x'], g['y'], g['cx'], g['cy'], g['fx'], g['fy']\n"
        "out = {}\n"
        "for rep in range(2):\n"
        "    for tag, a, b in (('u', x, y), ('c', cx, cy), ('f', fx, fy)):\n"
        "        out[tag + 'd1'], out[tag + 'i1'] = pcu.k_nearest_neighbors(a, b, 1)\n"
        "        out[tag + 'd5'], out[tag + 'i5'] = pcu.k_nearest_neighbors(a, b, 5)\n"
        "        out[tag + 'h'] = np.array(pcu.hausdorff_distance(a, b, return_index=True), dtype=np.float64)\n"
        "        out[tag + 'c'] = np.float64(pcu.chamfer_distance(a, b))\n"
        "        c, cxy, cyx = pcu.chamfer_distance(a, b, return_index=True); out[tag + 'cxy'] = cxy; out[tag + 'cyx'] = cyx\n"
        "M = pcu.pairwise_distances(x[:700], y[:600]); w = np.full(700, 1 / 700, np.float32); v = np.full(600, 1 / 600, np.float32)\n"
        "out['P'] = pcu.sinkhorn(w, v, M, eps=1e-2, max_iters=30, stop_thresh=0.0)\n"
        "np.savez(%r, **out)\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.npz"), str(tmp_path / "out.npz"))
    name, value = switch.split("=")
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, **{name: value}), timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "out.npz")
    for tag, a, b in (("u", x, y), ("c", cx, cy), ("f", fx, fy)):
        for k in (1, 5):
            d0, i0 = oracle.k_nearest_neighbors(a, b, k, kind=oracle_kind)
            assert np.array_equal(got[f"{tag}i{k}"], i0), (switch, tag, k)
            assert np.array_equal(got[f"{tag}d{k}"].view(np.uint32), np.asarray(d0).view(np.uint32)), (switch, tag, k)
        assert tuple(got[tag + "h"]) == tuple(float(v) for v in oracle.hausdorff_distance(a, b, return_index=True, kind=oracle_kind)), (switch, tag)
        c0, cxy0, cyx0 = oracle.chamfer_distance(a, b, return_index=True, kind=oracle_kind)
        assert np.array_equal(got[tag + "cxy"], cxy0) and np.array_equal(got[tag + "cyx"], cyx0), (switch, tag)
        assert abs(float(got[tag + "c"]) - float(c0)) <= 1e-4 * float(c0), (switch, tag)
    M = pcu.pairwise_distances(x[:700], y[:600]); w = np.full(700, 1 / 700, np.float32); v = np.full(600, 1 / 600, np.float32)
    P = pcu.sinkhorn(w, v, M, eps=1e-2, max_iters=30, stop_thresh=0.0)
    assert np.abs(got["P"] - P).max() <= 2e-4 * np.abs(P).max(), switch


@pytest.mark.gpu
def test_one_pass_build_overflow_falls_back(pcu, oracle_kind):
    """Uneven clouds overflow the fixed bucket slots of the one-pass index build (grid.h: k_bucket_onepass): every search pass
    gives up, the host rebuilds with the two-pass pipeline (pcu_hip.hip: search_finish) and the results are the reference's.
    Fresh contexts (a new device context per thread) so that the first call of each op takes the overflow path."""
    import threading
    rng = np.random.default_rng(31)
    x = rng.normal(0.5, 0.05, (300_000, 3)).astype(np.float32)
    y = rng.normal(0.5, 0.05, (250_000, 3)).astype(np.float32)
    d0, i0 = oracle.k_nearest_neighbors(x, y, 1, kind=oracle_kind)
    c0 = oracle.chamfer_distance(x, y, kind=oracle_kind)
    h0 = oracle.hausdorff_distance(x, y, return_index=True, kind=oracle_kind)
    res = {}

    def run(name, fn):
        def body():
            try:
                res[name] = fn()
            except Exception as e:      # surfaced below
                res[name] = e
        t = threading.Thread(target=body); t.start(); t.join()
    run("knn", lambda: pcu.k_nearest_neighbors(x, y, 1))
    run("chamfer", lambda: (pcu.chamfer_distance(x, y), pcu.chamfer_distance(x, y)))
    run("hausdorff", lambda: pcu.hausdorff_distance(x, y, return_index=True))
    for v in res.values():
        if isinstance(v, Exception):
            raise v
    d, i = res["knn"]
    assert np.array_equal(i, i0) and np.array_equal(np.asarray(d).view(np.uint32), np.asarray(d0).view(np.uint32))
    assert abs(float(res["chamfer"][0]) - float(c0)) <= 1e-4 * float(c0) and float(res["chamfer"][0]) == float(res["chamfer"][1])
    assert tuple(res["hausdorff"]) == tuple(h0)


@pytest.mark.gpu
def test_surface_clouds_take_the_finer_grid_and_stay_exact(pcu, oracle_kind):
    """Clouds sampled from a surface make the passes give up on the balance check and the call restarts on a finer grid, which the
    context then keeps (pcu_hip.hip: rescale_wanted); a volume-filling cloud afterwards switches back. Every call along the way
    returns the reference's neighbours."""
    rng = np.random.default_rng(41)
    def sphere(n):
        v = rng.normal(size=(n, 3)); v /= np.linalg.norm(v, axis=1, keepdims=True)
        return v.astype(np.float32)
    sx, sy = sphere(400_000), sphere(300_000)
    ux, uy = rng.random((300_000, 3), dtype=np.float32), rng.random((300_000, 3), dtype=np.float32)
    for x, y in ((sx, sy), (sx, sy), (ux, uy), (sx, uy), (ux, uy)):
        d, i = pcu.k_nearest_neighbors(x, y, 1)
        d0, i0 = oracle.k_nearest_neighbors(x, y, 1, kind=oracle_kind)
        assert np.array_equal(i, i0) and np.array_equal(np.asarray(d).view(np.uint32), np.asarray(d0).view(np.uint32))
        c, cxy, cyx = pcu.chamfer_distance(x, y, return_index=True)
        c0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=oracle_kind)
        assert np.array_equal(cxy, cxy0) and np.array_equal(cyx, cyx0) and abs(float(c) - float(c0)) <= 1e-4 * float(c0)
        assert abs(float(pcu.chamfer_distance(x, y)) - float(c0)) <= 1e-4 * float(c0)
        assert tuple(pcu.hausdorff_distance(x, y, return_index=True)) == tuple(oracle.hausdorff_distance(x, y, return_index=True, kind=oracle_kind))


@pytest.mark.gpu
def test_speculative_tree_top_keeps_tie_order(pcu, oracle_kind):
    """After a large call with genuine ties the next one starts the top of nanoflann's tree on a second stream while the searches run
    (pcu_hip.hip: kd_speculate) and the resolver adopts it. The calls before, with and after a speculation -- adopted, and wasted on a
    cloud without ties -- all return the reference's rows."""
    rng = np.random.default_rng(51)
    y = rng.random((300_000, 3), dtype=np.float32)
    y[1000:3000] = y[5000:7000]                          # duplicated rows: exact ties for every query near them
    x = rng.random((200_000, 3), dtype=np.float32)
    x[:500] = y[1000:1500]
    y2 = rng.random((300_000, 3), dtype=np.float32)      # generic: no ties
    for data, k in ((y, 4), (y, 4), (y2, 4), (y, 8), (y, 8)):
        d, i = pcu.k_nearest_neighbors(x, data, k)
        d0, i0 = oracle.k_nearest_neighbors(x, data, k, kind=oracle_kind)
        assert np.array_equal(i, i0) and np.array_equal(np.asarray(d).view(np.uint32), np.asarray(d0).view(np.uint32))


@pytest.mark.gpu
def test_poisoned_workspace_on_passes_that_give_up(pcu, oracle_kind, tmp_path):
    """A pass that gives up (balance check: refit or occupancy rescale) leaves its result rows unwritten, and the epilogues enqueued behind
    it must not follow what they find there (k_argmax_pair once read corr[0xffffffff] when every distance was NaN). The workspace is
    filled with 0xff before every call (PCU_HIP_DEBUG_POISON, a child process: the switch is read once) and a small uneven pair --
    a plane against a cloud with a tight cluster -- goes through every k = 1 op; results are the reference's."""
    import subprocess
    import sys
    rng = np.random.default_rng(12015)
    x = rng.random((922, 3)).astype(np.float32); x[:, 2] = 0.25
    y = np.concatenate([rng.random((815, 3)), rng.normal(0.5, 0.001, (271, 3))]).astype(np.float32)
    np.save(tmp_path / "x.npy", x); np.save(tmp_path / "y.npy", y)
    code = (
        "import sys, numpy as np; sys.path.insert(0, %r); import point_cloud_utils_amd as pcu\n"
        "x = np.load(%r); y = np.load(%r)\n"
        "out = {}\n"
        "for rep in range(2):\n"
        "    d, i = pcu.k_nearest_neighbors(x, y, 1)\n"
        "    h = pcu.hausdorff_distance(x, y, return_index=True); h1 = pcu.hausdorff_distance(y, x, return_index=True)\n"
        "    c, cxy, cyx = pcu.chamfer_distance(x, y, return_index=True)\n"
        "    c2 = pcu.chamfer_distance(x, y); h2 = pcu.hausdorff_distance(x, y)\n"
        "np.savez(%r, d=d, i=i, h=np.array(h, dtype=np.float64), h1=np.array(h1, dtype=np.float64), c=np.float64(c), cxy=cxy, cyx=cyx, c2=np.float64(c2), h2=np.float64(h2))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "x.npy"), str(tmp_path / "y.npy"), str(tmp_path / "out.npz"))
    subprocess.run([sys.executable, "-c", code], check=True, env=dict(os.environ, PCU_HIP_DEBUG_POISON="255"), timeout=600)
    got = np.load(tmp_path / "out.npz")
    d0, i0 = oracle.k_nearest_neighbors(x, y, 1, kind=oracle_kind)
    assert np.array_equal(got["i"], i0) and np.array_equal(got["d"].view(np.uint32), np.asarray(d0).view(np.uint32))
    assert tuple(got["h"]) == tuple(np.array(oracle.hausdorff_distance(x, y, return_index=True, kind=oracle_kind), dtype=np.float64))
    assert tuple(got["h1"]) == tuple(np.array(oracle.hausdorff_distance(y, x, return_index=True, kind=oracle_kind), dtype=np.float64))
    c0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=oracle_kind)
    assert np.array_equal(got["cxy"], cxy0) and np.array_equal(got["cyx"], cyx0)
    assert abs(float(got["c"]) - float(c0)) <= 1e-4 * float(c0) and abs(float(got["c2"]) - float(c0)) <= 1e-4 * float(c0)
    assert float(got["h2"]) == float(got["h"][0])
    # ... and a general-norm Chamfer on clouds with sparse tails: the first epilogue launch runs before the stragglers' rows exist and must
    # not follow what it finds in them (0x7f7f... = a huge row index)
    g1 = rng.normal(0.5, 0.05, (40_000, 3)).astype(np.float32); g2 = rng.normal(0.5, 0.05, (30_000, 3)).astype(np.float32)
    np.save(tmp_path / "g1.npy", g1); np.save(tmp_path / "g2.npy", g2)
    code2 = (
        "import sys, numpy as np; sys.path.insert(0, %r); import point_cloud_utils_amd as pcu\n"
        "a = np.load(%r); b = np.load(%r)\n"
        "for rep in range(2): v = [float(pcu.chamfer_distance(a, b, p_norm=p)) for p in (1, 3, np.inf)]\n"
        "np.save(%r, np.array(v))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "g1.npy"), str(tmp_path / "g2.npy"), str(tmp_path / "v.npy"))
    subprocess.run([sys.executable, "-c", code2], check=True, env=dict(os.environ, PCU_HIP_DEBUG_POISON="127"), timeout=600)
    v = np.load(tmp_path / "v.npy")
    for got_v, p in zip(v, (1, 3, np.inf)):
        v0 = float(oracle.chamfer_distance(g1, g2, p_norm=p, kind=oracle_kind))
        assert abs(got_v - v0) <= 1e-4 * v0


def test_config1_chamfer_10k_f64(pcu, oracle_kind):
    """BASELINE config 1, literally: chamfer_distance on two 10k-point fp64 U[0,1)^3 clouds (seeds 1000 / 1001) -- the
    reference's nanoflann CPU path gives the expected value and both index arrays; the GPU path must reproduce them (value within
    1e-6, indices exact). The CPU-only half (reference against the restatement) is tests/test_oracle.py::test_config1_cpu."""
    x, y = cloud(1000, 10_000, np.float64), cloud(1001, 10_000, np.float64)
    ch0, cxy0, cyx0 = oracle.chamfer_distance(x, y, return_index=True, kind=oracle_kind)
    ch, cxy, cyx = pcu.chamfer_distance(x, y, return_index=True)
    assert type(ch) == np.float64 and np.array_equal(cxy, cxy0) and np.array_equal(cyx, cyx0)
    assert abs(float(ch) - float(ch0)) <= 1e-6 * float(ch0)
    assert abs(float(pcu.chamfer_distance(x, y)) - float(ch0)) <= 1e-6 * float(ch0)
    for p in (1, np.inf):
        v0 = oracle.chamfer_distance(x, y, p_norm=p, kind=oracle_kind)
        assert abs(float(pcu.chamfer_distance(x, y, p_norm=p)) - float(v0)) <= 1e-6 * float(v0)
