"""GPU-built kd-tree against the oracle's on the x cloud of a fuzz case: kdcmp.py [seed] [case] [q|r]"""
import os, sys, ctypes
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import point_cloud_utils_amd as pcu
from point_cloud_utils_amd import _lib
import oracle
src = open(os.path.join(ROOT, "scratch", "fuzz.py")).read()
ns = {}
exec(src[src.index("def make("):src.index("dists = [")], {"np": np}, ns)
make = ns["make"]
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 404; case = int(sys.argv[2]) if len(sys.argv) > 2 else 148
rng = np.random.default_rng(seed0 * 1000 + case)
dtype = np.float32 if rng.random() < 0.6 else np.float64
big = rng.random() < 0.5
n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
dq, dr = rng.choice(dists), rng.choice(dists)
q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
pts = q if (len(sys.argv) <= 3 or sys.argv[3] == "q") else r
suf = "f32" if dtype == np.float32 else "f64"
vacc0, ni, nf, nlr = oracle.tree_dump(pts, 10)
for rep in range(3):
    vacc = np.empty(pts.shape[0], np.int64); nn = ctypes.c_int64(0)
    rc = getattr(_lib.lib(), "pcu_hip_debug_kd_tree_" + suf)(_lib.ctx(), pts.ctypes.data, pts.shape[0], 10, vacc.ctypes.data, ctypes.addressof(nn))
    bad = np.nonzero(vacc != vacc0)[0]
    print("rep", rep, "rc", rc, "nodes", nn.value, ni.shape[0], "vacc mismatches", bad.size, bad[:6], flush=True)
    if bad.size:
        b0 = bad[0]; print("   first at position", b0, "got", vacc[b0:b0 + 6], "want", vacc0[b0:b0 + 6]); print("   pts", pts[vacc[b0]], pts[vacc0[b0]])
