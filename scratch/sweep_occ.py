import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import point_cloud_utils_amd as pcu
from conftest import cloud
n, k = int(sys.argv[1]), int(sys.argv[2])
occs = [float(x) for x in sys.argv[3:]]
q, r = cloud(1000, n, np.float32), cloud(1001, n, np.float32)
tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
for occ in occs:
    pcu.set_cell_occupancy(occ)
    pcu.k_nearest_neighbors(tq, tr, k); torch.cuda.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pcu.k_nearest_neighbors(tq, tr, k); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = pcu.last_stats()
    print(f"occ {occ:5.2f}: {np.median(ts)*1e3:8.3f} ms  esc {st['n_escalated']:7d} ties {st['n_tie_flagged']:5d}/{st['n_tie_true']:4d}  idx {st['ms_index']:.3f} search {st['ms_search']:.3f} main {st['ms_kernel_search']:.3f}", flush=True)
