"""How often does a k = 1 search of fuzz case (seed, case) miss a genuinely tied query? tieflag.py [reps]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, collections
import point_cloud_utils_amd as pcu
src = open(os.path.join(ROOT, "scratch", "fuzz.py")).read()
ns = {}
exec(src[src.index("def make("):src.index("dists = [")], {"np": np}, ns)
make = ns["make"]
dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
seed0, case = 404, 148
rng = np.random.default_rng(seed0 * 1000 + case)
dtype = np.float32 if rng.random() < 0.6 else np.float64
big = rng.random() < 0.5
n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
dq, dr = rng.choice(dists), rng.choice(dists)
q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
import torch
tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 12
cnt = collections.Counter(); got = collections.Counter()
for _ in range(reps):
    d, c = pcu.k_nearest_neighbors(tr, tq, 1)
    st = pcu.last_stats(); cnt[(st["n_tie_true"], st["n_escalated"])] += 1; got[int(c[119605])] += 1
print(dict(cnt), dict(got), flush=True)
