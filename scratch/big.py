"""Scale check: 16M-vs-16M f32 Chamfer / KNN; 3000 random rows verified by brute force on the GPU (torch, fp64)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16000000
g = torch.Generator(device="cuda"); g.manual_seed(1)
x = torch.rand((n, 3), device="cuda", generator=g, dtype=torch.float32)
y = torch.rand((n, 3), device="cuda", generator=g, dtype=torch.float32)
for k in (1, 4):
    t0 = time.perf_counter(); d, c = pcu.k_nearest_neighbors(x, y, k); torch.cuda.synchronize(); t1 = time.perf_counter()
    d, c = pcu.k_nearest_neighbors(x, y, k); torch.cuda.synchronize(); t2 = time.perf_counter()
    rows = torch.randint(0, n, (3000,), device="cuda")
    bad = 0
    for r0 in range(0, 3000, 250):
        rr = rows[r0:r0 + 250]
        dd = torch.cdist(x[rr].double(), y.double())            # (250, n) fp64
        best = torch.topk(dd, k, dim=1, largest=False)
        got_d = d[rr].reshape(250, k).double(); got_i = c[rr].reshape(250, k)
        bad += int((got_i != best.indices).any(1).sum().item()) if k == 1 else int(((got_d - best.values).abs() > 1e-6).any(1).sum().item())
    print(f"n={n} k={k}: first {1e3*(t1-t0):.2f} ms, steady {1e3*(t2-t1):.2f} ms, rows off vs brute force: {bad}/3000", pcu.last_stats()["n_escalated"], flush=True)
t0 = time.perf_counter(); ch = pcu.chamfer_distance(x, y); t1 = time.perf_counter(); ch2 = pcu.chamfer_distance(x, y); t2 = time.perf_counter()
print(f"chamfer {n}: {ch} steady {1e3*(t2-t1):.2f} ms  ({2*n/(t2-t1)/1e9:.2f} G q-pts/s)", flush=True)
