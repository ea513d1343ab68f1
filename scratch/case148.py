"""fuzz seed 404 case 148 alone, operator by operator against the reference: case148.py [seed] [case]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import point_cloud_utils_amd as pcu
import oracle
src = open(os.path.join(ROOT, "scratch", "fuzz.py")).read()
ns = {}
exec(src[src.index("def make("):src.index("dists = [")], {"np": np}, ns)
make = ns["make"]
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 404; case = int(sys.argv[2]) if len(sys.argv) > 2 else 148
rng = np.random.default_rng(seed0 * 1000 + case)
dtype = np.float32 if rng.random() < 0.6 else np.float64
big = rng.random() < 0.5
n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
dq, dr = rng.choice(dists), rng.choice(dists)
q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
print(f"case {case}: {dtype.__name__} n={n} m={m} k={k} q={dq} r={dr}", flush=True)
kind = "ref"
if os.environ.get("ONLY_KNN"):
    d, c = pcu.k_nearest_neighbors(q, r, k); d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind="ref")
    bad = np.nonzero((c != c0) | (d.view(np.uint32) != np.asarray(d0).view(np.uint32)))[0]
    print("knn bad rows", bad.size, pcu.last_stats()["n_tie_true"], pcu.last_stats()["n_passes"], flush=True); sys.exit(0)
if os.environ.get("ONLY_CH"):
    ch, cxy, cyx = pcu.chamfer_distance(q, r, return_index=True); ch0, cxy0, cyx0 = oracle.chamfer_distance(q, r, return_index=True, kind="ref")
    print("chamfer idx mismatches", (cxy != cxy0).sum(), (cyx != cyx0).sum(), np.nonzero(cyx != cyx0)[0][:5], flush=True)
    for rep in range(3):
        ch, cxy, cyx = pcu.chamfer_distance(q, r, return_index=True)
        print("  again: mismatches", (cxy != cxy0).sum(), (cyx != cyx0).sum(), np.nonzero(cyx != cyx0)[0][:5], pcu.last_stats()["n_tie_true"], flush=True)
    d, c = pcu.k_nearest_neighbors(r, q, 1); d0, c0 = oracle.k_nearest_neighbors(r, q, 1, kind="ref")
    print("  knn r->q mismatches", (c != c0).sum(), np.nonzero(c != c0)[0][:5], pcu.last_stats()["n_tie_true"], flush=True)
    sys.exit(0)
for rep in range(2):
    d, c = pcu.k_nearest_neighbors(q, r, k); d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=kind)
    bad = np.nonzero((c != c0) | (d.view(np.uint32) != np.asarray(d0).view(np.uint32)))[0]
    print("knn", rep, "bad rows", bad.size, bad[:8], pcu.last_stats(), flush=True)
    for i in bad[:5]: print("   row", i, "q", q[i], "got", c[i], d[i], "want", c0[i], d0[i], "d(got)", np.linalg.norm(q[i] - r[c[i]]), "d(want)", np.linalg.norm(q[i] - r[c0[i]]))
    if k == 1:
        h = pcu.hausdorff_distance(q, r, return_index=True); h0 = oracle.hausdorff_distance(q, r, return_index=True, kind=kind)
        print("hausdorff idx", h, h0, h == h0, pcu.last_stats(), flush=True)
        ch, cxy, cyx = pcu.chamfer_distance(q, r, return_index=True); ch0, cxy0, cyx0 = oracle.chamfer_distance(q, r, return_index=True, kind=kind)
        print("chamfer idx", float(ch), float(ch0), (cxy != cxy0).sum(), (cyx != cyx0).sum(), pcu.last_stats(), flush=True)
        b2 = np.nonzero(cyx != cyx0)[0]
        for i in b2[:5]: print("   cyx row", i, "got", cyx[i], "want", cyx0[i], np.linalg.norm(r[i] - q[cyx[i]]), np.linalg.norm(r[i] - q[cyx0[i]]))
        b1 = np.nonzero(cxy != cxy0)[0]
        for i in b1[:5]: print("   cxy row", i, "got", cxy[i], "want", cxy0[i], np.linalg.norm(q[i] - r[cxy[i]]), np.linalg.norm(q[i] - r[cxy0[i]]))
        cf = float(pcu.chamfer_distance(q, r)); print("chamfer fused", cf, abs(cf - float(ch0)) / abs(float(ch0)), pcu.last_stats(), flush=True)
        hf = pcu.hausdorff_distance(q, r); print("hausdorff fused", hf, h0[0], hf == h0[0], pcu.last_stats(), flush=True)
