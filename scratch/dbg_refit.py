import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import point_cloud_utils_amd as pcu
import oracle
dtype = np.float32
rng = np.random.default_rng(9)
r = (rng.standard_normal((40000, 3)) * np.array([1.0, 0.05, 0.3])).astype(dtype)
q = (rng.standard_normal((30000, 3)) * np.array([1.5, 0.5, 0.5])).astype(dtype)
d, c = pcu.k_nearest_neighbors(q, r, 3); st = pcu.last_stats()
d0, c0 = oracle.k_nearest_neighbors(q, r, 3, kind="ref" if oracle.have_ref() else "port")
bad = np.nonzero((d != d0).any(1))[0]
print(st); print("bad rows", len(bad), bad[:10])
for i in bad[:5]:
    print(i, q[i], d[i], d0[i], c[i], c0[i])
print("r bbox", r.min(0), r.max(0))
