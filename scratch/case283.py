import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import point_cloud_utils_amd as pcu, oracle
rng = np.random.default_rng(283)
n, m, k = 231388, 167472, 16
q = (rng.random((n, 3)) * 1e-3 + 1000.0)
v = rng.normal(size=(m, 3)); r = v / np.linalg.norm(v, axis=1, keepdims=True)
pcu.k_nearest_neighbors(q[:100], r[:100], 1)
for kk in (16, 1):
    t = time.time(); d, c = pcu.k_nearest_neighbors(q, r, kk); dt = time.time() - t
    print(f"GPU knn k={kk}: {dt*1e3:.1f} ms", {a: b for a, b in pcu.last_stats().items() if a.startswith("n_")}, flush=True)
    t = time.time(); d0, c0 = oracle.k_nearest_neighbors(q, r, kk, kind="ref" if oracle.have_ref() else "port"); print(f"CPU: {(time.time()-t)*1e3:.1f} ms", "equal:", np.array_equal(c, c0) and np.array_equal(d, d0), flush=True)
t = time.time(); ch = pcu.chamfer_distance(q.astype(np.float32), r.astype(np.float32)); print(f"GPU chamfer f32: {(time.time()-t)*1e3:.1f} ms", flush=True)
