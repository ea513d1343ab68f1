import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n, k = 1_000_000, 16
rng = np.random.default_rng(9)
xy = rng.random((n, 2)) * 2 - 1
p = np.ascontiguousarray(np.concatenate([xy, (0.3 * np.sin(3 * xy[:, :1]) * np.cos(2 * xy[:, 1:2]))], 1).astype(np.float32))
tp = torch.from_numpy(p).cuda()
pcu.set_timing(2)
for i in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pcu.estimate_point_cloud_normals_knn(tp, k)
    torch.cuda.synchronize(); print("normals %.3f ms" % ((time.perf_counter() - t0) * 1e3), pcu.last_stats())
for i in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    pcu.k_nearest_neighbors(tp, tp, k)
    torch.cuda.synchronize(); print("self-knn %.3f ms" % ((time.perf_counter() - t0) * 1e3), pcu.last_stats())
