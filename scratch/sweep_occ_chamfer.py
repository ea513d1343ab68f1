import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = int(sys.argv[1])
occs = [float(x) for x in sys.argv[2:]]
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for occ in occs:
    pcu.set_cell_occupancy(occ)
    for _ in range(3): pcu.chamfer_distance(x, y)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(20): pcu.chamfer_distance(x, y)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 20
    st = pcu.last_stats()
    print(f"occ {occ:5.2f}: {dt*1e3:8.4f} ms/step  esc {st['n_escalated']:7d} ties {st['n_tie_flagged']:5d}", flush=True)
