"""k = 16 at 4M-vs-4M: time vs cell occupancy (the default comes from default_occupancy(k)). python scratch/occ_k16.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
rng = np.random.default_rng(3)
n = 4_000_000
x = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda(); y = torch.from_numpy(rng.random((n, 3), dtype=np.float32)).cuda()
for occ in (0, 4.5, 5.5, 6.5, 9.0, 0):
    pcu.set_cell_occupancy(occ)
    for _ in range(2): pcu.k_nearest_neighbors(x, y, 16)
    torch.cuda.synchronize(); ts = []
    for _ in range(5):
        t0 = time.perf_counter(); pcu.k_nearest_neighbors(x, y, 16); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    st = pcu.last_stats()
    print(f"occ {occ}: {np.median(ts)*1e3:.3f} ms  esc {st['n_escalated']} ties {st['n_tie_true']}", flush=True)
