"""The slowest single call of the round-5 sweep, alone (kernel trace: scratch/trace_py.sh line scratch/line_case.py)."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "."))
import point_cloud_utils_amd as pcu
rng = np.random.default_rng(3)
q = rng.random((237_000, 3)); t = np.random.default_rng(4).random(267_000)
r = np.ascontiguousarray(np.stack([t, t * 0.5, t * 0.25], axis=1))
for _ in range(3):
    t0 = time.perf_counter(); pcu.k_nearest_neighbors(q, r, 16); print(time.perf_counter() - t0, pcu.last_stats())
