import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import point_cloud_utils_amd as pcu
which = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200000
x = np.random.default_rng(1000).random((n, 3), dtype=np.float32); y = np.random.default_rng(1001).random((n, 3), dtype=np.float32)
t0 = time.time()
if which == "chamfer": print(pcu.chamfer_distance(x, y))
elif which == "chamfer_idx": print(pcu.chamfer_distance(x, y, return_index=True)[0])
elif which == "hausdorff": print(pcu.hausdorff_distance(x, y, return_index=True))
elif which == "onesided": print(pcu.one_sided_hausdorff_distance(x, y))
elif which == "torch":
    import torch
    print(pcu.chamfer_distance(torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda()))
print(which, "done in %.2f s" % (time.time() - t0), pcu.last_stats(), flush=True)
