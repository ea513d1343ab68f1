import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import point_cloud_utils_amd as pcu
rng = np.random.default_rng(0)
n = 1000000
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
cases = {
  "uniform": (rng.random((n,3)), rng.random((n,3))),
  "gauss_s0.05": (rng.normal(0.5,0.05,(n,3)), rng.normal(0.5,0.05,(n,3))),
  "mix_10pct_cluster": (np.concatenate([rng.random((n*9//10,3)), rng.normal(0.5,0.005,(n//10,3))]), np.concatenate([rng.random((n*9//10,3)), rng.normal(0.5,0.005,(n//10,3))])),
  "sphere_surface": None, "outlier_bbox": None,
}
v = rng.normal(size=(n,3)); v /= np.linalg.norm(v,axis=1,keepdims=True); w = rng.normal(size=(n,3)); w /= np.linalg.norm(w,axis=1,keepdims=True)
cases["sphere_surface"] = (v, w)
a = rng.random((n,3)); a[0] = [1000,1000,1000]; b = rng.random((n,3)); b[0] = [-1000,-1000,-1000]
cases["outlier_bbox"] = (a, b)
G = os.path.join(ROOT, 'tests', 'golden'); sys.path.insert(0, os.path.join(ROOT, 'tests'))
from conftest import mesh_samples
bv = np.load(os.path.join(G, 'bunny_v.npy')).astype(np.float64); bf = np.load(os.path.join(G, 'bunny_f.npy'))
cases["mesh_samples"] = (mesh_samples(bv, bf, n, seed=1), mesh_samples(bv, bf, n, seed=2))
cases["uniform_again"] = cases["uniform"]
sel = sys.argv[1:]
for name, (x, y) in cases.items():
    if sel and name not in sel: continue
    tx, ty = torch.from_numpy(x.astype(np.float32)).cuda(), torch.from_numpy(y.astype(np.float32)).cuda()
    t = timeit(lambda: pcu.chamfer_distance(tx, ty), n=3)
    st = pcu.last_stats()
    print(f"{name:20s} {t*1e3:9.3f} ms  esc {st['n_escalated']:8d} builds {st['n_grid_builds']} passes {st['n_passes']} main {st['ms_kernel_search']:.3f} idx {st['ms_index']:.3f}", flush=True)
