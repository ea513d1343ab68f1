"""Chamfer / k = 1 timings at small cloud sizes (device-resident, f32 and f64): python scratch/small_sizes.py"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import point_cloud_utils_amd as pcu
for dt in (torch.float32, torch.float64):
    for n in (64, 256, 512, 1000, 2000, 4000, 10000):
        x = torch.rand((n, 3), device="cuda", dtype=dt); y = torch.rand((n, 3), device="cuda", dtype=dt)
        for _ in range(5): pcu.chamfer_distance(x, y)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): pcu.chamfer_distance(x, y)
        torch.cuda.synchronize(); dt_c = (time.perf_counter() - t0) / 50
        for _ in range(5): pcu.k_nearest_neighbors(x, y, 1)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(50): pcu.k_nearest_neighbors(x, y, 1)
        torch.cuda.synchronize(); dt_k = (time.perf_counter() - t0) / 50
        print(f"{str(dt)[6:]:8s} n={n:6d} chamfer {dt_c * 1e3:.4f} ms  knn k=1 {dt_k * 1e3:.4f} ms", flush=True)
