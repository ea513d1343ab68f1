import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import point_cloud_utils_amd as pcu
dtype = np.float32
rng = np.random.default_rng(11)
n = 200000
r = np.concatenate([rng.random((n * 9 // 10, 3)), rng.normal(0.5, 0.002, (n // 10, 3))]).astype(dtype)
r[0] = [900.0, -700.0, 800.0]
q = np.concatenate([rng.random((n // 2, 3)), rng.normal(0.5, 0.002, (n // 2, 3))]).astype(dtype)
q[1] = [-500.0, 500.0, 0.0]
k = int(sys.argv[1]) if len(sys.argv) > 1 else 1
print("start", flush=True)
d, c = pcu.k_nearest_neighbors(q, r, k)
print("done", pcu.last_stats(), flush=True)
