import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import point_cloud_utils_amd as pcu
n = 1000000
q = torch.from_numpy(np.random.default_rng(1000).random((n, 3), dtype=np.float32)).cuda()
r = torch.from_numpy(np.random.default_rng(1001).random((n, 3), dtype=np.float32)).cuda()
def timeit(fn, reps=30):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
for k in (1, 8):
    t1 = timeit(lambda: pcu.k_nearest_neighbors(q, r, k))
    ix = pcu.DatasetIndex(r, k_hint=k)
    t2 = timeit(lambda: ix.k_nearest_neighbors(q, k))
    a = pcu.k_nearest_neighbors(q, r, k); b = ix.k_nearest_neighbors(q, k)
    print(f"k={k}: one-shot {t1*1e3:.3f} ms, persistent index {t2*1e3:.3f} ms, equal={bool((a[1] == b[1]).all()) and bool((a[0] == b[0]).all())}", flush=True)
    ix.close()
