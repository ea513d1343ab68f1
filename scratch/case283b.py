import sys, time, numpy as np
sys.path.insert(0, "/root/repo")
import point_cloud_utils_amd as pcu
rng = np.random.default_rng(283)
n, m, k = 231388, 167472, 16
q = (rng.random((n, 3)) * 1e-3 + 1000.0)
v = rng.normal(size=(m, 3)); r = v / np.linalg.norm(v, axis=1, keepdims=True)
pcu.k_nearest_neighbors(q[:100], r[:100], 1)
t = time.time(); d, c = pcu.k_nearest_neighbors(q, r, 16); print("k=16", time.time() - t, pcu.last_stats())
t = time.time(); d, c = pcu.k_nearest_neighbors(q, r, 1); print("k=1", time.time() - t, pcu.last_stats())
pcu.k_nearest_neighbors(q[:100], r[:100], 1)
