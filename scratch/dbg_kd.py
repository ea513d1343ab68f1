import sys, ctypes, numpy as np
sys.path.insert(0, '.')
import oracle
from point_cloud_utils_amd import _lib
rng = np.random.default_rng(3)
for n in (7, 50, 3000, 20000):
    pts = rng.random((n, 3)).astype(np.float32)
    vacc0, ni, nf, nlr = oracle.tree_dump(pts, 10)
    vacc = np.empty(n, np.int64); nn = ctypes.c_int64(0)
    print("build", n, flush=True)
    rc = _lib.lib().pcu_hip_debug_kd_tree_f32(_lib.ctx(), pts.ctypes.data, n, 10, vacc.ctypes.data, ctypes.addressof(nn))
    print(n, rc, nn.value, ni.shape[0], np.array_equal(vacc, vacc0), flush=True)
    if not np.array_equal(vacc, vacc0):
        print(vacc[:20], vacc0[:20])
