"""hausdorff_distance 1M-vs-1M and k = 1 knn timings (row / arg-max variants of k_search1_flat)."""
import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1_000_000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for name, fn in (("hausdorff 1M", lambda: pcu.hausdorff_distance(x, y)), ("chamfer+idx 1M", lambda: pcu.chamfer_distance(x, y, return_index=True))):
    for _ in range(10): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(200): fn()
    torch.cuda.synchronize(); print(name, "%.4f ms" % ((time.perf_counter() - t0) / 200 * 1e3), flush=True)
