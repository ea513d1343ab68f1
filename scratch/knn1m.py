import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1000000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for i in range(3):
    d, c = pcu.k_nearest_neighbors(x, y, 1)
print(pcu.last_stats())
