import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np
import point_cloud_utils_amd as pcu
rng = np.random.default_rng(3)
q2 = rng.random((200_000, 3), dtype=np.float32); r2 = rng.random((200_000, 3), dtype=np.float32)
for i in range(5):
    t0 = time.perf_counter(); pcu.k_nearest_neighbors(q2, r2, 200); print("k=200 200k/200k: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
