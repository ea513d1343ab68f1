"""fuzz seed 512 case 75 (float32, ONE lattice query against 1520 points on the unit sphere), operator by operator against the reference"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import oracle
src = open(os.path.join(ROOT, "scratch", "fuzz.py")).read()
ns = {}
exec(src[src.index("def make("):src.index("dists = [")], {"np": np}, ns)
make = ns["make"]
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 512; case = int(sys.argv[2]) if len(sys.argv) > 2 else 75
rng = np.random.default_rng(seed0 * 1000 + case)
dtype = np.float32 if rng.random() < 0.6 else np.float64
big = rng.random() < 0.5
n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
dq, dr = rng.choice(dists), rng.choice(dists)
q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
print(f"case {case}: {dtype.__name__} n={n} m={m} k={k} q={dq} r={dr}", q[:3], flush=True)
oracle.build()
d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind="ref")
# brute force in the input type, the reference's operation order
dx = q[:, None, 0] - r[None, :, 0]; dy = q[:, None, 1] - r[None, :, 1]; dz = q[:, None, 2] - r[None, :, 2]
d2 = ((dx * dx) + (dy * dy)) + (dz * dz)
o = np.argsort(d2[0], kind="stable")[:5]
print("reference:", d0, c0, " brute force best five:", o, d2[0][o], np.sqrt(d2[0][o]))
print("ulps between the two best d2:", (d2[0][o[1]] - d2[0][o[0]]) / np.spacing(d2[0][o[0]]))
h0 = oracle.hausdorff_distance(q, r, return_index=True, kind="ref"); ch0 = oracle.chamfer_distance(q, r, return_index=True, kind="ref")
print("reference hausdorff", h0, "chamfer", ch0[0], ch0[1][:3], ch0[2][:3])
if "--cpu" in sys.argv: sys.exit(0)
import point_cloud_utils_amd as pcu
d, c = pcu.k_nearest_neighbors(q, r, k); print("gpu knn:", d, c, pcu.last_stats())
print("gpu hausdorff idx", pcu.hausdorff_distance(q, r, return_index=True), pcu.last_stats())
print("gpu hausdorff", pcu.hausdorff_distance(q, r))
ch = pcu.chamfer_distance(q, r, return_index=True); print("gpu chamfer idx", ch[0], ch[1][:3], ch[2][:3], "mismatching y->x rows", int((np.asarray(ch[2]) != ch0[2]).sum()), "x->y", int((np.asarray(ch[1]) != ch0[1]).sum()))
print("gpu chamfer", pcu.chamfer_distance(q, r), "ref", ch0[0])
print("one-sided r->q", pcu.one_sided_hausdorff_distance(r, q), oracle.one_sided_hausdorff_distance(r, q, kind="ref"))
