import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1_000_000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for _ in range(30): pcu.hausdorff_distance(x, y)
