#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the reference's own nanoflann (oracle/_ref/libpcu_ref.so, compiled in place
from /root/reference/external/nanoflann/nanoflann.hpp). Run where /root/reference exists:

    python tests/golden/make_golden.py

Each fixture stores small seeded inputs and the reference outputs (distances as raw bits via exact float arrays).
The reference ships no golden vectors for this path, so these are generated from the reference code."""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
import oracle  # noqa: E402

REF_DATA = "/root/reference/data"


def nonfinite_cases():
    """Fixtures with non-finite and signed-zero coordinates (names nf_*): what the reference's nanoflann returns where its kd-tree
    survives the input -- any query (rows with a non-finite coordinate find nothing: -1 / -1.0), datasets with -0.0, and datasets
    whose infinities have one sign per axis. (A NaN in the dataset, or +inf and -inf along one axis, make the reference's tree
    bounds NaN and its rows traversal-dependent: those inputs are rejected by the GPU path, tests/test_gpu_parity.py.)"""
    rng = np.random.default_rng(20250924)
    cases = {}

    def add(name, q, r, k, squared=False):
        d, c = oracle.knn(q, r, k, squared_distances=squared, kind="ref")
        cases["nf_" + name] = dict(q=q, r=r, k=np.int64(k), squared=np.int64(squared), d=d, c=c)

    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        # signed zeros: a lattice around the origin, half of the zeros negative; duplicates -> exact ties
        g = np.stack(np.meshgrid(*[np.arange(-3, 4)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(dt)
        gz = g.copy(); gz[(gz == 0) & (rng.random(gz.shape) < 0.5)] = -0.0
        qz = (rng.integers(-6, 7, (300, 3)) / 2).astype(dt); qz[(qz == 0) & (rng.random(qz.shape) < 0.5)] = -0.0
        add(f"negzero_k3_{tag}", qz, np.concatenate([gz, g[::3]]), 3)
        q = rng.random((700, 3), dtype=dt); r = rng.random((1100, 3), dtype=dt)
        qi = q.copy(); qi[5, 0] = np.inf; qi[17, 1] = -np.inf; qi[40] = [np.inf, np.inf, -np.inf]; qi[699, 2] = np.inf
        add(f"inf_query_k1_{tag}", qi, r, 1)
        add(f"inf_query_k3_{tag}", qi, r, 3, squared=True)
        qn = q.copy(); qn[0, 1] = np.nan; qn[33] = np.nan; qn[500, 2] = np.nan; qn[501, 0] = np.inf
        add(f"nan_query_k1_{tag}", qn, r, 1)
        add(f"nan_query_k4_{tag}", qn, r, 4)
        rp = r.copy(); rp[0, 2] = np.inf; rp[7, 0] = np.inf; rp[400] = np.inf; rp[1099, 1] = np.inf
        add(f"pinf_dataset_k1_{tag}", q, rp, 1)
        add(f"pinf_dataset_k5_{tag}", qn, rp, 5)
        rm = r.copy(); rm[3, 0] = np.inf; rm[9, 1] = -np.inf; rm[10, 1] = -np.inf; rm[800, 2] = np.inf
        add(f"mixed_axes_inf_dataset_k2_{tag}", q, rm, 2)
        add(f"all_inf_column_k1_{tag}", q[:200], np.concatenate([r[:300, :2], np.full((300, 1), -np.inf, dt)], 1), 1)
    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **c)
    print("wrote", len(cases), "non-finite fixtures")
    nonfinite_metric_cases(rng)
    # BASELINE config 1: chamfer_distance of two 10k-point fp64 clouds through the reference's CPU path (inputs are seeded, not stored)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import cloud
    x, y = cloud(1000, 10_000, np.float64), cloud(1001, 10_000, np.float64)
    ch, cxy, cyx = oracle.chamfer_distance(x, y, return_index=True, kind="ref")
    np.savez_compressed(os.path.join(HERE, "config1.npz"), chamfer=np.float64(ch), cxy=cxy, cyx=cyx)


P_NORMS = (2, 1, np.inf, -np.inf, 0, 3)


def near_tie_cases():
    """near_tie_*.npz: float32 clouds at offset 1000 from each other (the randomised sweep's seed 405, cases 289 and 37, regenerated draw by
    draw). d2 ~ 3e6 with an ulp of 0.25: the reference's incremental branch bound (nanoflann.hpp:1601-1613) discards the branch of the true
    minimum for some queries, i.e. its answer is one ulp WORSE than the minimum of its own distance arithmetic (asserted here). k = 1, squared."""
    def make(rng, n, dist):
        if dist == "clusters":
            c = rng.random((8, 3)); a = c[rng.integers(0, 8, n)] + rng.normal(0, 0.003, (n, 3))
        elif dist == "offset": a = rng.random((n, 3)) * 1e-3 + 1000.0
        elif dist == "sphere": v = rng.normal(size=(n, 3)); a = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
        return np.ascontiguousarray(a.astype(np.float32))
    dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
    worse = 0
    for tag, case in (("a", 289), ("b", 37)):
        rng = np.random.default_rng(405 * 1000 + case)
        assert rng.random() < 0.6; rng.random()                      # (float32; size class)
        n = int(rng.integers(1, 3000)); m = int(rng.integers(1, 3000))
        rng.choice([1, 1, 1, 2, 5, 16])
        dq, dr = rng.choice(dists), rng.choice(dists)
        q, r = make(rng, n, dq), make(rng, m, dr)
        for name, a, b in ((f"near_tie_{tag}_xy", q, r), (f"near_tie_{tag}_yx", r, q)):
            d, c = oracle.knn(a, b, 1, squared_distances=True, kind="ref")
            D = (a[:, None, :] - b[None, :, :]).astype(np.float32)
            worse += int((np.asarray(d).reshape(-1) > ((D[..., 0] * D[..., 0] + D[..., 1] * D[..., 1]) + D[..., 2] * D[..., 2]).min(axis=1)).sum())
            np.savez_compressed(os.path.join(HERE, name + ".npz"), q=a, r=b, k=np.int64(1), squared=np.int64(1), d=d, c=c)
    assert worse > 0, "the fixtures no longer hold a query whose reference answer is not the minimum"
    print("wrote 4 near-tie fixtures;", worse, "reference answers are not the minimum of their own arithmetic")


def nonfinite_metric_cases(rng):
    """nf_{f32,f64}_metrics.npz: the metrics on clouds with non-finite rows, where the reference has a stable answer
    (src/point_cloud_distance.cpp:90-93,223 and __init__.py:112-115: unmatched source rows carry -1 / -1.0, Hausdorff's max ignores
    them, Chamfer gathers through index -1). Cases: `inf` (single-signed infinities in x), `mixed` (+inf and -inf on different axes),
    `last` (additionally y's LAST row infinite: the gather through -1 meets inf - inf = NaN), `nan` (NaN rows in x: one-sided x -> y is
    stable, and so is Chamfer's VALUE -- NaN -- for every ord but 0), `allbad` (no finite source row: (-1.0, 0, -1))."""
    import warnings
    warnings.simplefilter("ignore")
    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        out = {}
        x = rng.random((900, 3), dtype=dt); y = rng.random((700, 3), dtype=dt)
        xi = x.copy(); xi[5, 0] = np.inf; xi[100, 1] = np.inf; xi[899, 2] = np.inf; xi[40] = np.inf
        xm = x.copy(); xm[5, 0] = np.inf; xm[77, 0] = np.inf; xm[9, 1] = -np.inf; xm[10, 1] = -np.inf; xm[899, 2] = np.inf; xm[40] = [np.inf, -np.inf, np.inf]      # one sign per axis
        yl = y.copy(); yl[-1, 0] = np.inf
        xn = x.copy(); xn[7, 1] = np.nan; xn[50] = np.nan; xn[51, 0] = np.inf
        for name, a, b in (("inf", xi, y), ("mixed", xm, y), ("last", xi, yl)):
            out[f"{name}_x"] = a; out[f"{name}_y"] = b
            out[f"{name}_os_xy"] = np.array(oracle.one_sided_hausdorff_distance(a, b, kind="ref"), np.float64)
            out[f"{name}_os_yx"] = np.array(oracle.one_sided_hausdorff_distance(b, a, kind="ref"), np.float64)
            out[f"{name}_os_xy_sq"] = np.array(oracle.one_sided_hausdorff_distance(a, b, squared_distances=True, kind="ref"), np.float64)
            out[f"{name}_h"] = np.array(oracle.hausdorff_distance(a, b, True, kind="ref"), np.float64)
            out[f"{name}_h_rev"] = np.array(oracle.hausdorff_distance(b, a, True, kind="ref"), np.float64)
            ch, cxy, cyx = oracle.chamfer_distance(a, b, return_index=True, kind="ref")
            out[f"{name}_cxy"] = cxy; out[f"{name}_cyx"] = cyx
            out[f"{name}_ch"] = np.array([oracle.chamfer_distance(a, b, p_norm=p, kind="ref") for p in P_NORMS], np.float64)
            out[f"{name}_ch_rev"] = np.array([oracle.chamfer_distance(b, a, p_norm=p, kind="ref") for p in P_NORMS], np.float64)
        out["nan_x"] = xn; out["nan_y"] = y
        out["nan_os_xy"] = np.array(oracle.one_sided_hausdorff_distance(xn, y, kind="ref"), np.float64)
        for leaf in (1, 33):      # (stable: the same for every tree)
            assert oracle.one_sided_hausdorff_distance(xn, y, max_points_per_leaf=leaf, kind="ref") == tuple(out["nan_os_xy"])
        out["nan_ch"] = np.array([oracle.chamfer_distance(xn, y, p_norm=p, kind="ref") for p in P_NORMS if p != 0], np.float64)
        assert np.isnan(out["nan_ch"]).all() and np.isnan(oracle.chamfer_distance(y, xn, kind="ref"))
        bad = np.full((12, 3), np.nan, dt); bad[3] = np.inf
        out["allbad_os"] = np.array(oracle.one_sided_hausdorff_distance(bad, y, kind="ref"), np.float64)
        np.savez_compressed(os.path.join(HERE, f"nf_{tag}_metrics.npz"), **out)
    print("wrote the non-finite metric fixtures")


def main():
    oracle.build()
    assert oracle.have_ref(), "needs /root/reference (oracle/_ref)"
    if "--only-nonfinite" in sys.argv:
        return nonfinite_cases()
    if "--only-near-ties" in sys.argv:
        return near_tie_cases()
    nonfinite_cases()
    near_tie_cases()
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    from conftest import read_ply_vertices
    rng = np.random.default_rng(20250321)
    cases = {}

    def add(name, q, r, k, squared=False):
        d, c = oracle.knn(q, r, k, squared_distances=squared, kind="ref")
        cases[name] = dict(q=q, r=r, k=np.int64(k), squared=np.int64(squared), d=d, c=c)

    for dt, tag in ((np.float32, "f32"), (np.float64, "f64")):
        add(f"uniform_k1_{tag}", rng.random((1500, 3), dtype=dt), rng.random((1200, 3), dtype=dt), 1)
        add(f"uniform_k5_{tag}", rng.random((800, 3), dtype=dt), rng.random((900, 3), dtype=dt), 5, squared=True)
        add(f"uniform_k16_{tag}", rng.random((600, 3), dtype=dt), rng.random((1000, 3), dtype=dt), 16)
        add(f"k_gt_m_{tag}", rng.random((50, 3), dtype=dt), rng.random((7, 3), dtype=dt), 10)
        add(f"single_ref_{tag}", rng.random((40, 3), dtype=dt), rng.random((1, 3), dtype=dt), 1)
        base = rng.random((700, 3), dtype=dt)
        add(f"duplicates_k1_{tag}", rng.random((500, 3), dtype=dt), np.concatenate([base, base]), 1)
        add(f"duplicates_k4_{tag}", rng.random((500, 3), dtype=dt), np.concatenate([base, base]), 4)
        add(f"self_k3_{tag}", base[:400], base[:400], 3)
        g = np.stack(np.meshgrid(*[np.arange(9)] * 3, indexing="ij"), -1).reshape(-1, 3).astype(dt)
        add(f"lattice_k7_{tag}", (rng.integers(0, 18, (400, 3)) / 2).astype(dt), g, 7)
        add(f"far_queries_{tag}", (rng.random((300, 3), dtype=dt) * dt(0.1) + dt(5.0)), rng.random((900, 3), dtype=dt), 2)
        add(f"planar_{tag}", rng.random((500, 3), dtype=dt) * np.array([1, 1, 0], dt), rng.random((800, 3), dtype=dt) * np.array([1, 1, 0], dt), 3)
    bunny = read_ply_vertices(os.path.join(REF_DATA, "bunny.ply"))
    dup = read_ply_vertices(os.path.join(REF_DATA, "bunny_duplicates.ply"))
    add("bunny_vs_dup_f32", bunny.astype(np.float32)[::3], dup.astype(np.float32), 2)
    add("dup_self_f64", dup.astype(np.float64)[::2], dup.astype(np.float64), 3)
    # extreme dynamic range: exponents spread over the whole fp32 range -> kd-tree hundreds of levels deep,
    # d2 overflowing to +inf for most pairs
    bits = rng.integers(0, 2**32, (3000, 3), dtype=np.uint64).astype(np.uint32)
    ext = bits.view(np.float32).copy(); ext[~np.isfinite(ext)] = 1.0
    add("extreme_range_k2_f32", ext[::3], ext, 2)
    add("extreme_range_small_k3_f32", (ext * np.float32(1e-20))[::5], ext * np.float32(1e-20), 3)

    for name, c in cases.items():
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **c)
    # Hausdorff / Chamfer scalars on a pair of the fixtures
    out = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        a = rng.random((1000, 3), dtype=dt); b = rng.random((500, 3), dtype=dt)
        h = oracle.hausdorff_distance(a, b, return_index=True, kind="ref")
        o1 = oracle.one_sided_hausdorff_distance(a, b, kind="ref")
        o2 = oracle.one_sided_hausdorff_distance(b, a, squared_distances=True, kind="ref")
        ch, cxy, cyx = oracle.chamfer_distance(a, b, return_index=True, kind="ref")
        ch1 = oracle.chamfer_distance(a, b, p_norm=1, kind="ref")
        chinf = oracle.chamfer_distance(a, b, p_norm=np.inf, kind="ref")
        ch3 = oracle.chamfer_distance(a, b, p_norm=3, kind="ref")
        out[f"a_{tag}"] = a; out[f"b_{tag}"] = b
        out[f"hausdorff_{tag}"] = np.array([h[0], h[1], h[2]], dtype=np.float64)
        out[f"one_sided_ab_{tag}"] = np.array(o1, dtype=np.float64)
        out[f"one_sided_ba_sq_{tag}"] = np.array(o2, dtype=np.float64)
        out[f"chamfer_{tag}"] = np.array([ch, ch1, chinf, ch3], dtype=np.float64)
        out[f"cxy_{tag}"] = cxy; out[f"cyx_{tag}"] = cyx
    np.savez_compressed(os.path.join(HERE, "metrics.npz"), **out)
    print("wrote", len(cases) + 1, "fixtures")


if __name__ == "__main__":
    main()
