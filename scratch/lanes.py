"""Batched Hausdorff (config 4 shape) against the number of pairs kept in flight: lanes.py [pairs] [n]"""
import sys, time, torch
from point_cloud_utils_amd import batched
P = int(sys.argv[1]) if len(sys.argv) > 1 else 32; n = int(sys.argv[2]) if len(sys.argv) > 2 else 262144
g = torch.Generator(device="cuda"); g.manual_seed(4)
xs = [torch.rand((n, 3), generator=g, device="cuda") for _ in range(P)]; ys = [torch.rand((n, 3), generator=g, device="cuda") for _ in range(P)]
for rep in range(2):
    for w in (1, 2, 3, 4, 5, 6, 8):
        batched.batched_hausdorff(lambda p: (xs[p], ys[p]), P, workers=w)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(5): batched.batched_hausdorff(lambda p: (xs[p], ys[p]), P, workers=w)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
        print(f"workers {w}: {dt * 1e3:.3f} ms per {P} pairs = {dt / P * 1e6:.1f} us per pair", flush=True)
