import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1_000_000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
print("hausdorff", file=sys.stderr)
for _ in range(520): pcu.hausdorff_distance(x, y)
print("chamfer", file=sys.stderr)
for _ in range(520): pcu.chamfer_distance(x, y)
