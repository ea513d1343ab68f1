import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = int(sys.argv[1]); k = int(sys.argv[2])
q = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
r = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for _ in range(3): pcu.k_nearest_neighbors(q, r, k)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10): pcu.k_nearest_neighbors(q, r, k)
torch.cuda.synchronize(); print(f"n={n} k={k}: {(time.perf_counter()-t0)/10*1e3:.3f} ms", pcu.last_stats(), flush=True)
