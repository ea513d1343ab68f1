"""Stage timers of the one-pass build (PCU_HIP_PROF_BUILD2=1) at the headline's and config 4's cloud sizes."""
import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import numpy as np, torch
import point_cloud_utils_amd as pcu
for n in (1_000_000, 262_144):
    x = torch.from_numpy(np.random.default_rng(1).random((n, 3), dtype=np.float32)).cuda()
    y = torch.from_numpy(np.random.default_rng(2).random((n, 3), dtype=np.float32)).cuda()
    print("n =", n, file=sys.stderr, flush=True)
    for _ in range(4):
        pcu.chamfer_distance(x, y)
