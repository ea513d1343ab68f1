"""Round-6 stress: threads with their own contexts, alternating sizes / dtypes / operators (handed-down layouts, shared grids), cancellation in a batch."""
import os, sys, threading, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "."), "tests"))
import numpy as np
import point_cloud_utils_amd as pcu
from point_cloud_utils_amd import batched
import oracle
oracle.build(); kind = "ref" if oracle.have_ref() else "port"
rng = np.random.default_rng(66)
cases = []
for n, m, dt, sc in [(200_000, 200_000, np.float32, 1.0), (200_000, 200_000, np.float32, 2.5), (150_000, 90_000, np.float32, 1.0), (200_000, 200_000, np.float64, 1.0),
                     (5_000, 5_000, np.float32, 1.0), (200_000, 200_000, np.float32, 0.2), (1_500, 200_000, np.float64, 1.0)]:
    x = (rng.random((n, 3)) * sc).astype(dt); y = (rng.random((m, 3)) * sc + (0.3 if sc > 2 else 0.0)).astype(dt)
    cases.append((x, y, float(oracle.chamfer_distance(x, y, kind=kind)), oracle.hausdorff_distance(x, y, return_index=True, kind=kind)))
errs = []
def worker(seed):
    r = np.random.default_rng(seed)
    try:
        for it in range(60):
            x, y, ch0, h0 = cases[int(r.integers(0, len(cases)))]
            if r.random() < 0.5:
                ch = float(pcu.chamfer_distance(x, y)); tol = 1e-4 if x.dtype == np.float32 else 1e-6
                assert abs(ch - ch0) <= tol * ch0, (ch, ch0)
            else:
                assert pcu.hausdorff_distance(x, y, return_index=True) == h0
    except BaseException as e:
        errs.append(e)
ths = [threading.Thread(target=worker, args=(s,)) for s in range(4)]
t0 = time.perf_counter()
for t in ths: t.start()
for t in ths: t.join()
print("threads done", time.perf_counter() - t0, "errors", errs)
assert not errs
# cancellation in the middle of a batch, then the batch again
P = 24
xs = [rng.random((262144, 3), dtype=np.float32) for _ in range(P)]; ys = [rng.random((262144, 3), dtype=np.float32) for _ in range(P)]
rows = batched.batched_hausdorff(lambda p: (xs[p], ys[p]), P)
th = threading.Thread(target=lambda: (time.sleep(0.01), pcu.cancel())); th.start()
try:
    batched.batched_hausdorff(lambda p: (xs[p], ys[p]), P); print("batch finished before the request")
except KeyboardInterrupt:
    print("batch cancelled")
th.join()
rows2 = batched.batched_hausdorff(lambda p: (xs[p], ys[p]), P)
assert np.array_equal(np.asarray(rows), np.asarray(rows2))
for p in (0, 7, 23):
    assert tuple(rows2[p]) == tuple(float(v) for v in oracle.hausdorff_distance(xs[p], ys[p], return_index=True, kind=kind))
print("stress ok")
