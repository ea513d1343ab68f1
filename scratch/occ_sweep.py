"""Chamfer time vs grid occupancy on uneven clouds (potential of an occupancy rescale). python scratch/occ_sweep.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import point_cloud_utils_amd as pcu
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from conftest import mesh_samples
rng = np.random.default_rng(0)
n = 1000000
def timeit(fn, reps=3):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
v = rng.normal(size=(n,3)); v /= np.linalg.norm(v,axis=1,keepdims=True); w = rng.normal(size=(n,3)); w /= np.linalg.norm(w,axis=1,keepdims=True)
cases = {
  "uniform": (rng.random((n,3)), rng.random((n,3))),
  "gauss": (rng.normal(0.5,0.05,(n,3)), rng.normal(0.5,0.05,(n,3))),
  "sphere": (v, w),
  "mesh": None,
  "plane_noise": (np.c_[rng.random((n,2)), 0.001*rng.normal(size=n)], np.c_[rng.random((n,2)), 0.001*rng.normal(size=n)]),
}
G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), 'tests', 'golden')
bv = np.load(os.path.join(G, 'bunny_v.npy')).astype(np.float64); bf = np.load(os.path.join(G, 'bunny_f.npy'))
cases['mesh'] = (mesh_samples(bv, bf, n, seed=1), mesh_samples(bv, bf, n, seed=2))
for name, (x, y) in cases.items():
    tx, ty = torch.from_numpy(np.ascontiguousarray(x, dtype=np.float32)).cuda(), torch.from_numpy(np.ascontiguousarray(y, dtype=np.float32)).cuda()
    ref = None; out = []
    for occ in (2.0, 1.0, 0.5, 0.25, 0.125, 0.06, 0.03):
        pcu.set_cell_occupancy(occ)
        t = timeit(lambda: pcu.chamfer_distance(tx, ty))
        val = float(pcu.chamfer_distance(tx, ty)); st = pcu.last_stats()
        if ref is None: ref = val
        out.append(f"occ {occ:5.3f}: {t*1e3:6.3f} ms esc {st['n_escalated']:6d} b{st['n_grid_builds']}{'' if val == ref else ' VALUE DIFFERS'}")
    print(f"{name:12s} " + " | ".join(out), flush=True)
pcu.set_cell_occupancy(0)
