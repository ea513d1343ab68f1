import sys, time, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import point_cloud_utils_amd as pcu, oracle
from test_gpu_parity import _fuzz_cloud
case = int(sys.argv[1]) if len(sys.argv) > 1 else 5
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere"]
rng = np.random.default_rng(4200 + case)
dtype = np.float32 if case % 3 else np.float64
hi = 300000 if case % 2 else 3000
n, m = int(rng.integers(1, hi)), int(rng.integers(1, hi))
k = min(int(rng.choice([1, 1, 2, 5, 16])), m)
q, r = _fuzz_cloud(rng, n, dists[case % 9], dtype), _fuzz_cloud(rng, m, dists[(case * 5 + 3) % 9], dtype)
print("case", case, dists[case % 9], dists[(case * 5 + 3) % 9], n, m, k, dtype.__name__, flush=True)
pcu.k_nearest_neighbors(q[:100], r[:100], 1)
for name, f in (("knn", lambda: pcu.k_nearest_neighbors(q, r, k)), ("hausdorff", lambda: pcu.hausdorff_distance(q, r, return_index=True)),
                ("chamfer_idx", lambda: pcu.chamfer_distance(q, r, return_index=True)), ("chamfer", lambda: pcu.chamfer_distance(q, r))):
    t = time.time(); f(); dt = time.time() - t
    print(f"GPU {name}: {dt*1e3:.1f} ms", {k_: v for k_, v in pcu.last_stats().items() if k_.startswith("n_")}, flush=True)
kind = "ref" if oracle.have_ref() else "port"
for name, f in (("knn", lambda: oracle.k_nearest_neighbors(q, r, k, kind=kind)), ("hausdorff", lambda: oracle.hausdorff_distance(q, r, return_index=True, kind=kind)),
                ("chamfer", lambda: oracle.chamfer_distance(q, r, return_index=True, kind=kind))):
    t = time.time(); f(); print(f"CPU {name}: {(time.time()-t)*1e3:.1f} ms", flush=True)
