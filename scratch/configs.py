"""Runs BASELINE.json configs 2-5 on one GPU: timing (device-resident) + parity vs the oracle."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import point_cloud_utils_amd as pcu
import oracle
from conftest import cloud, read_ply_vertices
kind = "ref" if oracle.have_ref() else "port"
which = sys.argv[1:] or ["2", "3", "4", "5"]
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize(); ts = []
    for _ in range(n):
        t0 = time.perf_counter(); fn(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    return float(np.median(ts))
if "2" in which:
    q, r = cloud(1000, 1000000, np.float32), cloud(1001, 1000000, np.float32)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    t = timeit(lambda: pcu.k_nearest_neighbors(tq, tr, 1)); st = pcu.last_stats()
    d, c = pcu.k_nearest_neighbors(tq, tr, 1)
    d0, c0 = oracle.k_nearest_neighbors(q, r, 1, kind=kind)
    print("C2 knn k=1 1M/1M f32: %.3f ms  %.3g q/s  idx_eq=%s d_eq=%s" % (t * 1e3, 1e6 / t, np.array_equal(c.cpu().numpy(), c0), np.array_equal(d.cpu().numpy(), d0)), st, flush=True)
if "3" in which:
    q, r = cloud(1000, 4000000, np.float32), cloud(1001, 4000000, np.float32)
    tq, tr = torch.from_numpy(q).cuda(), torch.from_numpy(r).cuda()
    t = timeit(lambda: pcu.k_nearest_neighbors(tq, tr, 16), n=3); st = pcu.last_stats()
    d, c = pcu.k_nearest_neighbors(tq, tr, 16)
    t0 = time.perf_counter(); d0, c0 = oracle.k_nearest_neighbors(q, r, 16, kind=kind); tc = time.perf_counter() - t0
    ce = np.array_equal(c.cpu().numpy(), c0); de = np.array_equal(d.cpu().numpy(), d0)
    print("C3 knn k=16 4M/4M f32: %.3f ms  %.3g q/s  idx_eq=%s d_eq=%s  (cpu %.1f s)" % (t * 1e3, 4e6 / t, ce, de, tc), st, flush=True)
    if not ce: print("   rows differing:", int((c.cpu().numpy() != c0).any(1).sum()))
if "4" in which:
    npairs = 8
    pairs = [(torch.from_numpy(cloud(1000 + 2 * p, 262144, np.float32)).cuda(), torch.from_numpy(cloud(1001 + 2 * p, 262144, np.float32)).cuda()) for p in range(npairs)]
    from point_cloud_utils_amd import batched
    def run():
        return [pcu.hausdorff_distance(x, y, return_index=True) for x, y in pairs]
    t = timeit(run); res = run()
    for w in (2, 4, 8):
        tb = timeit(lambda: batched.batched_hausdorff(lambda p: pairs[p], npairs, workers=w))
        rb = batched.batched_hausdorff(lambda p: pairs[p], npairs, workers=w)
        print("   batched workers=%d: %.3f ms total (%.3g q-pts/s) same=%s" % (w, tb * 1e3, npairs * 2 * 262144 / tb, all(tuple(rb[p]) == res[p] for p in range(npairs))), flush=True)
    ok = True
    for p in range(2):
        h0 = oracle.hausdorff_distance(pairs[p][0].cpu().numpy(), pairs[p][1].cpu().numpy(), return_index=True, kind=kind)
        ok &= (res[p] == h0)
    print("C4 hausdorff %d x (256k/256k) f32: %.3f ms total, %.3f ms/pair, %.3g q-pts/s  parity(2 pairs)=%s" % (npairs, t * 1e3, t * 1e3 / npairs, npairs * 2 * 262144 / t, ok), pcu.last_stats(), flush=True)
if "5" in which:
    p = os.path.join(ROOT, "tests", "golden", "bunny_v.npy")
    bunny = np.load(p).astype(np.float64)
    f = np.load(os.path.join(ROOT, "tests", "golden", "bunny_f.npy"))
    rng = np.random.default_rng(5)
    tri = bunny[f]; areas = 0.5 * np.linalg.norm(np.cross(tri[:, 1] - tri[:, 0], tri[:, 2] - tri[:, 0]), axis=1)
    fi = rng.choice(len(f), 1000000, p=areas / areas.sum()); u = rng.random(1000000); v = rng.random(1000000); su = np.sqrt(u)
    S = (1 - su)[:, None] * tri[fi, 0] + (su * (1 - v))[:, None] * tri[fi, 1] + (su * v)[:, None] * tri[fi, 2]
    S = np.ascontiguousarray(S)
    tb, ts_ = torch.from_numpy(bunny).cuda(), torch.from_numpy(S).cuda()
    t = timeit(lambda: pcu.chamfer_distance(tb, ts_, return_index=True)); st = pcu.last_stats()
    ch, cxy, cyx = pcu.chamfer_distance(tb, ts_, return_index=True)
    ch0, cxy0, cyx0 = oracle.chamfer_distance(bunny, S, return_index=True, kind=kind)
    print("C5 chamfer bunny(2885) vs 1M samples f64: %.3f ms  %.3g q-pts/s  cxy_eq=%s cyx_eq=%s rel=%.2e" % (t * 1e3, (len(bunny) + 1e6) / t, np.array_equal(cxy.cpu().numpy(), cxy0), np.array_equal(cyx.cpu().numpy(), cyx0), abs(float(ch) - float(ch0)) / float(ch0)), st, flush=True)
