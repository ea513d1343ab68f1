"""Where a synchronous Chamfer step spends its host time: the plain Python loop of bench.py, the same loop on the raw C entry point
(ctypes, arguments prepared once), and the library's own marks (PCU_HIP_HOST_PROF=1 -> stderr every 1000 calls)."""
import ctypes, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
from point_cloud_utils_amd import _lib, _Dev, _fn, Stats
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
x = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
y = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
for _ in range(20): pcu.chamfer_distance(x, y)
N = 2000
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(N): v = float(pcu.chamfer_distance(x, y))
t1 = time.perf_counter()
print(f"python loop      {1e6 * (t1 - t0) / N:8.2f} us/step  value {v}")
d = _Dev(x, y); f = _fn("chamfer", d.suffix); means = (ctypes.c_double * 2)(); st = Stats()
args = (d.ctx, d.pa, n, d.pb, n, 2.0, 10, ctypes.addressof(means), None, None, d.flags, d.stream, ctypes.addressof(st))
for _ in range(20): f(*args)
t0 = time.perf_counter()
for _ in range(N): f(*args)
t1 = time.perf_counter()
print(f"raw C entry loop {1e6 * (t1 - t0) / N:8.2f} us/step  value {np.float32(means[0]) + np.float32(means[1])}")
