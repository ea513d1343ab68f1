"""Randomised parity sweep inside the driver-run suite (-m gpu): the HIP path through the C ABI against the reference's own nanoflann
(oracle/_ref; the C restatement where that is not built) on cases drawn fresh every round.

Round 4's review: the 2 100-case sweeps lived in scratch/fuzz.py (builder-run), 14 cases in the suite, and each round's sweep found holes in
that round's kernels. This file moves the volume in front of the driver:
  * 160 cases of the general sweep (ten point distributions x sizes 1 .. 60k x f32 / f64 x k in {1, 1, 1, 2, 5, 16}); every operator per case;
  * the moderate-offset family: a query cloud 3, 10, 30, 100 box sizes away from its dataset (d2 gaps of tens of ulps between neighbours;
    before, only offset 1000 and one offset-3 case were covered), f32 and f64, k in {1, 16};
  * lane-pass near ties: dataset point pairs constructed to lie 1 .. 8 ulps apart in d2 from grid-interior queries, the regime in which a
    minimum-taking search and nanoflann's incremental bound could disagree (DESIGN 2 "Near ties").
Seeds derive from ROUND -- one more than the newest BENCH_rNN.json the driver has left at the repo root, so every round's run draws cases no
earlier round ran, without anybody editing this file -- or from PCU_SWEEP_SEED. Sizes are chosen so that the whole file runs in about two
minutes, the CPU reference included; every GPU call also carries a generous time bound (a pathological path is a finding too)."""
import os
import time

import numpy as np
import pytest

import oracle

pytestmark = pytest.mark.gpu



def _round_number():
    import glob
    import re
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    done = [int(m.group(1)) for f in glob.glob(os.path.join(root, "BENCH_r*.json")) for m in [re.search(r"BENCH_r(\d+)\.json$", f)] if m]
    return max(done, default=0) + 1


ROUND = _round_number()
SEED = int(os.environ.get("PCU_SWEEP_SEED", 1000 * ROUND + 7))
N_GENERAL = 160
DISTS = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]


@pytest.fixture(scope="module")
def pcu():
    import point_cloud_utils_amd as m
    from point_cloud_utils_amd import _lib
    assert _lib.device_count() > 0, "no GPU visible: the gfx950 path has no CPU fallback"
    return m


def make(rng, n, dist, dtype):
    if dist == "uniform": a = rng.random((n, 3))
    elif dist == "plane": a = rng.random((n, 3)); a[:, 2] = 0.25
    elif dist == "line": a = np.zeros((n, 3)); a[:, 0] = rng.random(n)
    elif dist == "clusters": c = rng.random((8, 3)); a = c[rng.integers(0, 8, n)] + rng.normal(0, 0.003, (n, 3))
    elif dist == "dups": b = rng.random((max(n // 3, 1), 3)); a = b[rng.integers(0, b.shape[0], n)]
    elif dist == "lattice": a = rng.integers(0, 12, (n, 3)).astype(np.float64)
    elif dist == "offset": a = rng.random((n, 3)) * 1e-3 + 1000.0
    elif dist == "aniso": a = rng.random((n, 3)) * [1000.0, 1.0, 0.001]
    elif dist == "sphere": v = rng.normal(size=(n, 3)); a = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
    else: a = np.concatenate([rng.random((n - n // 4, 3)), rng.normal(0.5, 0.001, (n // 4, 3))])
    return np.ascontiguousarray(a.astype(dtype))


def check_all_operators(pcu, kind, q, r, k, tag, bound_s=5.0):
    """k_nearest_neighbors (indices + distance bits); for k == 1 also Hausdorff (tuple), Chamfer with indices (both arrays) and the two fused
    calls (no indices asked for) -- everything the reference would return for this pair."""
    tol = 1e-4 if q.dtype == np.float32 else 1e-6
    t0 = time.perf_counter()
    d, c = pcu.k_nearest_neighbors(q, r, k)
    assert time.perf_counter() - t0 < bound_s, (tag, "slow k_nearest_neighbors", pcu.last_stats())
    d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=kind)
    # (n == k == 1: the product squeezes to 0-d -- numpy.squeeze semantics, unpinned in the reference -- the oracle's wrapper to (1,): compare squeezed)
    assert np.array_equal(np.squeeze(c), np.squeeze(c0)), (tag, pcu.last_stats())
    assert np.array_equal(np.atleast_1d(np.squeeze(d)).view(np.uint8), np.atleast_1d(np.squeeze(d0)).view(np.uint8)), tag
    if k != 1:
        return
    t0 = time.perf_counter()
    h = pcu.hausdorff_distance(q, r, return_index=True)
    ch, cxy, cyx = pcu.chamfer_distance(q, r, return_index=True)
    ch_f, h_f = pcu.chamfer_distance(q, r), pcu.hausdorff_distance(q, r)
    assert time.perf_counter() - t0 < 4 * bound_s, (tag, "slow metric", pcu.last_stats())
    h0 = oracle.hausdorff_distance(q, r, return_index=True, kind=kind)
    ch0, cxy0, cyx0 = oracle.chamfer_distance(q, r, return_index=True, kind=kind)
    assert h == h0, (tag, h, h0)
    assert np.array_equal(cxy, cxy0) and np.array_equal(cyx, cyx0), tag
    assert abs(float(ch) - float(ch0)) <= tol * abs(float(ch0)) + 1e-30, tag
    assert abs(float(ch_f) - float(ch0)) <= tol * abs(float(ch0)) + 1e-30 and h_f == h0[0], (tag, ch_f, ch0, h_f, h0)


@pytest.mark.parametrize("case", range(N_GENERAL))
def test_general_sweep(pcu, oracle_kind, case):
    rng = np.random.default_rng([SEED, case])
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    hi = 60000 if rng.random() < 0.5 else 3000
    n, m = int(rng.integers(1, hi)), int(rng.integers(1, hi))
    k = min(int(rng.choice([1, 1, 1, 2, 5, 16])), m)
    dq, dr = str(rng.choice(DISTS)), str(rng.choice(DISTS))
    q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
    check_all_operators(pcu, oracle_kind, q, r, k, f"seed {SEED} case {case}: {dtype.__name__} n={n} m={m} k={k} q={dq} r={dr}")


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("k", [1, 16])
@pytest.mark.parametrize("boxes", [3, 10, 30, 100])
def test_moderate_offset(pcu, oracle_kind, dtype, k, boxes):
    """The query cloud sits `boxes` dataset-box sizes away from the dataset along a random direction: no query finds its neighbours in the 27
    cells, every one goes through the wave-per-query rounds, and the gaps between candidate d2 shrink to tens of ulps (float32) as the offset
    grows -- the regime between the grid-interior case and the offset-1000 fixtures of round 4."""
    rng = np.random.default_rng([SEED, 7001, boxes, k, dtype().itemsize])
    n, m = int(rng.integers(8000, 40000)), int(rng.integers(8000, 40000))
    u = rng.normal(size=3); u /= np.linalg.norm(u)
    r = rng.random((m, 3)).astype(dtype)
    q = np.ascontiguousarray((rng.random((n, 3)) * rng.choice([0.05, 1.0]) + boxes * u).astype(dtype))
    check_all_operators(pcu, oracle_kind, q, r, k, f"offset {boxes} boxes, {dtype.__name__}, k={k}, n={n}, m={m}", bound_s=10.0)
    check_all_operators(pcu, oracle_kind, r[: n // 2], q, k, f"offset {boxes} boxes (roles swapped), {dtype.__name__}, k={k}", bound_s=10.0)


def near_tie_pairs(rng, dtype, nq, m_background):
    """Queries in the interior of a uniform dataset, each given two extra dataset points r1, r2 that are its two nearest neighbours with
    computed squared distances 1 .. 8 ulps apart (in the reference's own operation order and type). Queries and the (y, z) offsets of the
    pair lie on a binary lattice coarse enough for every difference, square and sum to be exact: r1 = q + (0, a, b), r2 = q + (delta, -a, -b)
    would tie exactly for delta = 0, and delta = sqrt(t ulp(d2)) lifts r2 by t ulps (the construction is checked in the input type below).
    Returns (queries, dataset, number of constructed pairs)."""
    T = dtype
    L = 2.0 ** (12 if T == np.float32 else 24)
    q = (np.floor((0.2 + 0.6 * rng.random((nq, 3))) * L) / L).astype(T)
    bg = rng.random((m_background, 3)).astype(T)
    steps = max(2, int(0.05 * m_background ** (-1.0 / 3.0) * L))        # the pair sits well inside the background's spacing
    a = (rng.integers(1, steps + 1, nq) * rng.choice([-1, 1], nq) / L).astype(T)
    b = (rng.integers(0, steps + 1, nq) * rng.choice([-1, 1], nq) / L).astype(T)
    def d2(u, v):
        dx, dy, dz = u[:, 0] - v[:, 0], u[:, 1] - v[:, 1], u[:, 2] - v[:, 2]
        return ((dx * dx) + (dy * dy)) + (dz * dz)
    r1 = np.stack([q[:, 0], q[:, 1] + a, q[:, 2] + b], 1).astype(T)
    d1 = d2(q, r1)
    t = rng.integers(1, 9, nq)
    delta = np.sqrt(t * np.spacing(d1).astype(np.float64)) * rng.choice([-1, 1], nq)
    swap = rng.random(nq) < 0.5                                          # r2's (y, z) offsets: (-a, -b) or (-b, -a)
    oy, oz = np.where(swap, -b, -a).astype(T), np.where(swap, -a, -b).astype(T)
    r2 = np.stack([(q[:, 0] + delta.astype(T)).astype(T), q[:, 1] + oy, q[:, 2] + oz], 1).astype(T)
    gap = (d2(q, r2) - d1) / np.spacing(d1)
    sel = np.nonzero((gap >= 1) & (gap <= 8))[0]
    data = np.concatenate([bg, r1[sel], r2[sel]]).astype(T)
    data = np.ascontiguousarray(data[rng.permutation(len(data))])        # (row order decides nothing here; mixed in for good measure)
    return np.ascontiguousarray(q[sel]), data, len(sel)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("rep", range(4))
def test_lane_pass_near_ties(pcu, oracle_kind, dtype, rep):
    """Queries that the lane-per-query pass certifies, whose two best candidates are 1 .. 8 ulps apart in d2: the GPU must return what the
    reference returns (k = 1: the lane pass; k = 2: the pair in the reference's order; the fused metrics through the same queries)."""
    rng = np.random.default_rng([SEED, 9001, rep, dtype().itemsize])
    q, data, npairs = near_tie_pairs(rng, dtype, 20000, 60000)
    assert npairs > 10000, npairs                     # the construction works for most queries
    # the constructed pairs really are the two nearest, 1 .. 8 ulps apart, for nearly all of the queries (a background point may intrude)
    d0, c0 = oracle.k_nearest_neighbors(q, data, 2, kind=oracle_kind, squared_distances=True)
    gap = (d0[:, 1] - d0[:, 0]) / np.spacing(d0[:, 0])
    assert ((gap >= 1) & (gap <= 8)).mean() > 0.9
    for k in (1, 2):
        check_all_operators(pcu, oracle_kind, q, data, k, f"near ties rep {rep} {dtype.__name__} k={k}")
    # many more queries than a straggler pass would take: they are answered by the lane-per-query kernels
    st = pcu.last_stats()
    assert st.get("n_unresolved", 0) + st.get("n_escalated", 0) < len(q) // 10, st


@pytest.mark.parametrize("nt", [-1, 0, 4])
def test_num_threads_is_accepted(pcu, oracle_kind, nt):
    """a11 (src/common/common.h:182-212 OmpSetParallelism, src/point_cloud_distance.cpp:129): `num_threads` is the reference's OpenMP knob --
    -1 all cores, 0 serial, n threads; results do not depend on it there, and here it is accepted (positionally and by keyword) and ignored."""
    rng = np.random.default_rng([SEED, 11, nt + 1])
    q, r = rng.random((120_000, 3)).astype(np.float32), rng.random((110_000, 3)).astype(np.float32)      # (>= 100 000 rows: where the reference's team engages)
    d0, c0 = oracle.k_nearest_neighbors(q, r, 3, kind=oracle_kind)
    d, c = pcu.k_nearest_neighbors(q, r, 3, num_threads=nt)
    assert np.array_equal(c, c0) and np.array_equal(d, d0)
    d, c = pcu.k_nearest_neighbors(q, r, 3, False, 10, nt)
    assert np.array_equal(c, c0) and np.array_equal(d, d0)
    d1, c1 = pcu.k_nearest_neighbors(q[:500], r[:700], 1, squared_distances=True, max_points_per_leaf=7, num_threads=nt)
    e1, f1 = oracle.k_nearest_neighbors(q[:500], r[:700], 1, squared_distances=True, max_points_per_leaf=7, kind=oracle_kind)
    assert np.array_equal(c1, f1) and np.array_equal(d1, e1)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("dq,dr", [("uniform", "uniform"), ("uniform", "sphere"), ("clusters", "uniform"), ("mixed", "mixed"), ("plane", "uniform"),
                                   ("dups", "uniform"), ("aniso", "aniso"), ("sphere", "sphere")])
def test_fused_sum_is_the_sum_of_the_rows(pcu, dtype, dq, dr):
    """The fused Chamfer sum has no rows to compare, and its value tolerance (1e-4 / 1e-6) would hide a query served a wrong neighbour. Its
    fp64 means as the C ABI returns them (before the wrapper rounds to the input dtype) must therefore equal the fp64 sum of the k = 1 rows of
    the same clouds to summation rounding: one wrong minimum among 10^5 queries moves a mean by >= 1e-8 relative. Exercises the lane pass of the
    fused kernel -- centre row, cuts, run list, adoption (search.h) -- on ten kinds of input, both directions."""
    import ctypes
    from point_cloud_utils_amd import _Dev, _fn, Stats
    rng = np.random.default_rng([SEED, 4242, len(dq), len(dr), dtype().itemsize])
    n, m = int(rng.integers(30000, 120000)), int(rng.integers(30000, 120000))
    x, y = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
    dxy, _ = pcu.k_nearest_neighbors(x, y, 1)
    dyx, _ = pcu.k_nearest_neighbors(y, x, 1)
    dv = _Dev(x, y); means = (ctypes.c_double * 2)(); st = Stats()
    for _ in range(2):                                # (twice: both parities of the build's fill words, a warm context)
        rc = _fn("chamfer", dv.suffix)(dv.ctx, dv.pa, n, dv.pb, m, 2.0, 10, ctypes.addressof(means), None, None, dv.flags, dv.stream, ctypes.addressof(st))
        assert rc == 0
        for got, rows, cnt in ((means[0], dxy, n), (means[1], dyx, m)):
            want = float(np.asarray(rows).astype(np.float64).sum()) / cnt
            assert abs(got - want) <= 1e-10 * want + 1e-300, (dq, dr, got, want)


def test_staged_pass_on_the_shared_grid_sums_the_same_rows(pcu, tmp_path):
    """Round 6: two-sided calls lay ONE grid over both clouds (grid2.h); with PCU_HIP_BRICK=1 the fused Chamfer sum of float32 clouds takes the
    LDS-staged lane pass (csrc/search_brick.h) instead of k_search1_flat. Same check as above -- the C ABI's fp64 means against the fp64 sum of
    the k = 1 rows, 1e-10 -- in a child process with the switch set, on a uniform pair at the headline's density (every block staged), sizes that
    leave a partial last block, clouds of different size, and inputs whose blocks fall back to the global scan (plane, clusters: short uneven rows);
    the rows come from THIS process (k_nearest_neighbors: per-cloud grids, k_search1_flat)."""
    import subprocess
    import sys
    rng = np.random.default_rng([SEED, 777])
    cases = [("uniform", "uniform", 1_000_000, 1_000_000), ("uniform", "uniform", 300_001, 170_003), ("uniform", "uniform", 5_000, 9_000),
             ("plane", "uniform", 90_000, 120_000), ("clusters", "mixed", 150_000, 100_000), ("sphere", "sphere", 200_000, 200_000)]
    arrs = {}
    for i, (dq, dr, n, m) in enumerate(cases):
        arrs[f"x{i}"], arrs[f"y{i}"] = make(rng, n, dq, np.float32), make(rng, m, dr, np.float32)
    np.savez(tmp_path / "in.npz", **arrs)
    code = (
        "import sys, ctypes, numpy as np; sys.path.insert(0, %r); import point_cloud_utils_amd as pcu\n"
        "from point_cloud_utils_amd import _Dev, _fn, Stats\n"
        "g = np.load(%r); out = {}\n"
        "for i in range(%d):\n"
        "    x, y = g['x%%d' %% i], g['y%%d' %% i]\n"
        "    dv = _Dev(x, y); means = (ctypes.c_double * 2)(); st = Stats()\n"
        "    for rep in range(3):\n"
        "        rc = _fn('chamfer', dv.suffix)(dv.ctx, dv.pa, len(x), dv.pb, len(y), 2.0, 10, ctypes.addressof(means), None, None, dv.flags, dv.stream, ctypes.addressof(st))\n"
        "        assert rc == 0\n"
        "        out['m%%d_%%d' %% (i, rep)] = np.array([means[0], means[1]])\n"
        "np.savez(%r, **out)\n") % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), str(tmp_path / "in.npz"), len(cases), str(tmp_path / "out.npz"))
    r = subprocess.run([sys.executable, "-c", code], env=dict(os.environ, PCU_HIP_BRICK="1"), timeout=900, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    got = np.load(tmp_path / "out.npz")
    for i, (dq, dr, n, m) in enumerate(cases):
        x, y = arrs[f"x{i}"], arrs[f"y{i}"]
        dxy, _ = pcu.k_nearest_neighbors(x, y, 1)
        dyx, _ = pcu.k_nearest_neighbors(y, x, 1)
        want = np.array([float(np.asarray(dxy).astype(np.float64).sum()) / n, float(np.asarray(dyx).astype(np.float64).sum()) / m])
        for rep in range(3):
            assert np.all(np.abs(got[f"m{i}_{rep}"] - want) <= 1e-10 * want + 1e-300), (cases[i], rep, got[f"m{i}_{rep}"], want)


@pytest.mark.parametrize("dtype", [np.float32, np.float64], ids=["f32", "f64"])
@pytest.mark.parametrize("n,m", [(63, 63), (64, 64), (65, 63), (63, 5000), (5000, 63), (64, 100000), (100, 257), (1000, 1023), (1025, 700),
                                 (2047, 2047), (2048, 2048), (2049, 2047), (2047, 40000), (40000, 2047), (5000, 2049), (16383, 16384), (16384, 32767),
                                 (32768, 2048), (3000, 100000), (100000, 3000)])
def test_sizes_around_the_small_cloud_thresholds(pcu, oracle_kind, dtype, n, m):
    """Round 5 moved two thresholds from 16384 / 32768 to 64 (pcu_hip.hip: wave_only_below -- query clouds below it go wave-per-query from
    the start -- and bucket_plan's minimum for the one-pass index build; 2048 for most of the round): every combination of the paths on either
    side of the old, the intermediate and the new values (and of the build's 1024-point sample), all operators, one-shot and through a
    persistent index."""
    rng = np.random.default_rng([SEED, 2048, n, m, dtype().itemsize])
    q, r = rng.random((n, 3)).astype(dtype), rng.random((m, 3)).astype(dtype)
    c = min(n, m, 64)
    r[:c] = q[:c]                                     # a few exact zero distances / duplicates across the clouds
    for k in (1, 5):
        check_all_operators(pcu, oracle_kind, q, r, min(k, m), f"thresholds n={n} m={m} k={k} {dtype.__name__}")
    with pcu.DatasetIndex(r, k_hint=2) as index:
        for k in (1, 3):
            d, c = index.k_nearest_neighbors(q, k)
            d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=oracle_kind)
            assert np.array_equal(c, c0) and np.array_equal(np.asarray(d).view(np.uint8), np.asarray(d0).view(np.uint8)), (n, m, k)
