"""Replay of one fuzz case under a fixed cell occupancy (debugging aid). python scratch/repro.py seed case occ"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import point_cloud_utils_amd as pcu
src = open(os.path.join(ROOT, "scratch", "fuzz.py")).read()
ns = {}
exec(src[src.index("def make("):src.index("dists = [")], {"np": np}, ns)
make = ns["make"]
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
seed0, case, occ = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
rng = np.random.default_rng(seed0 * 1000 + case)
dtype = np.float32 if rng.random() < 0.6 else np.float64
big = rng.random() < 0.5
n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
dq, dr = rng.choice(dists), rng.choice(dists)
q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
print(f"case {case}: {dtype.__name__} n={n} m={m} k={k} q={dq} r={dr} occ={occ}", flush=True)
if occ > 0: pcu.set_cell_occupancy(occ)
ops = sys.argv[4].split(',') if len(sys.argv) > 4 else None
for name, fn in (("knn", lambda: pcu.k_nearest_neighbors(q, r, k)), ("hausdorff idx", lambda: pcu.hausdorff_distance(q, r, return_index=True)),
                 ("hausdorff", lambda: pcu.hausdorff_distance(q, r)), ("chamfer idx", lambda: pcu.chamfer_distance(q, r, return_index=True)), ("chamfer", lambda: pcu.chamfer_distance(q, r))):
    if ops and name not in ops: continue
    print(" ", name, flush=True); fn(); print("   ok", pcu.last_stats()["n_passes"], flush=True)
