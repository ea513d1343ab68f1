"""Headline step time against the grid occupancy (points per cell): python scratch/occ_headline.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1000000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
def run(steps=60):
    for _ in range(5): pcu.chamfer_distance(x, y)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(steps): v = pcu.chamfer_distance(x, y)
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / steps * 1e3, float(v)
for occ in (0, 1.25, 1.5, 1.75, 2.0, 2.5, 3.0, 4.0):
    pcu.set_cell_occupancy(occ)
    ms, v = run(); st = pcu.last_stats()
    print(f"occupancy {occ:4.2f}: {ms:.4f} ms per step  value {v:.9g}  escalated {st['n_escalated']}", flush=True)
pcu.set_cell_occupancy(0)
