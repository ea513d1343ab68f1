"""Randomised parity sweep against the oracle (reference's nanoflann when built, else the C restatement)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import point_cloud_utils_amd as pcu
import oracle
kind = "ref" if oracle.have_ref() else "port"
seed0 = int(sys.argv[1]) if len(sys.argv) > 1 else 0
ncases = int(sys.argv[2]) if len(sys.argv) > 2 else 40
first = int(sys.argv[3]) if len(sys.argv) > 3 else 0          # (to replay a tail of a sweep)

def make(rng, n, dist, dtype):
    if dist == "uniform": a = rng.random((n, 3))
    elif dist == "plane": a = rng.random((n, 3)); a[:, 2] = 0.25
    elif dist == "line": a = np.zeros((n, 3)); a[:, 0] = rng.random(n)
    elif dist == "clusters":
        c = rng.random((8, 3)); a = c[rng.integers(0, 8, n)] + rng.normal(0, 0.003, (n, 3))
    elif dist == "dups": b = rng.random((max(n // 3, 1), 3)); a = b[rng.integers(0, b.shape[0], n)]
    elif dist == "lattice": a = rng.integers(0, 12, (n, 3)).astype(np.float64)
    elif dist == "offset": a = rng.random((n, 3)) * 1e-3 + 1000.0
    elif dist == "aniso": a = rng.random((n, 3)) * [1000.0, 1.0, 0.001]
    elif dist == "sphere": v = rng.normal(size=(n, 3)); a = v / np.maximum(np.linalg.norm(v, axis=1, keepdims=True), 1e-30)
    elif dist == "mixed": a = np.concatenate([rng.random((n - n // 4, 3)), rng.normal(0.5, 0.001, (n // 4, 3))])
    return np.ascontiguousarray(a.astype(dtype))

dists = ["uniform", "plane", "line", "clusters", "dups", "lattice", "offset", "aniso", "sphere", "mixed"]
bad = 0
t0 = time.time()
for case in range(first, ncases):
    rng = np.random.default_rng(seed0 * 1000 + case)
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    big = rng.random() < 0.5
    n = int(rng.integers(1, 300000 if big else 3000)); m = int(rng.integers(1, 300000 if big else 3000))
    k = int(rng.choice([1, 1, 1, 2, 5, 16])); k = min(k, m)
    dq, dr = rng.choice(dists), rng.choice(dists)
    q, r = make(rng, n, dq, dtype), make(rng, m, dr, dtype)
    tag = f"case {case}: {dtype.__name__} n={n} m={m} k={k} q={dq} r={dr}"
    if os.environ.get("FUZZ_VERBOSE"): print(tag, flush=True)
    try:
        V = bool(os.environ.get('FUZZ_VERBOSE'))
        if V: print('  knn', flush=True)
        tg = time.time(); d, c = pcu.k_nearest_neighbors(q, r, k); tg = time.time() - tg
        if tg > 1.0: print(f'SLOW knn {tg:.2f} s', tag, pcu.last_stats(), flush=True)
        if V: print('  knn done', pcu.last_stats(), flush=True)
        d0, c0 = oracle.k_nearest_neighbors(q, r, k, kind=kind)
        # (n == k == 1: the product squeezes to 0-d -- numpy.squeeze semantics, unpinned in the reference -- the oracle's wrapper to (1,): compare squeezed)
        ok = np.array_equal(np.squeeze(c), np.squeeze(c0)) and np.array_equal(np.atleast_1d(np.squeeze(d)).view(np.uint8), np.atleast_1d(np.squeeze(d0)).view(np.uint8))
        if not ok: tag += f" KNN rows differing {int((np.atleast_1d(c) != np.atleast_1d(c0)).sum())} first c={np.atleast_2d(c)[:2]} c0={np.atleast_2d(c0)[:2]} d={np.atleast_2d(d)[:2]} d0={np.atleast_2d(d0)[:2]} stats={pcu.last_stats()}"
        if k == 1 and ok:
            if V: print('  hausdorff', flush=True)
            tg = time.time(); h = pcu.hausdorff_distance(q, r, return_index=True); tg = time.time() - tg
            if tg > 1.0: print(f'SLOW hausdorff {tg:.2f} s', tag, flush=True)
            h0 = oracle.hausdorff_distance(q, r, return_index=True, kind=kind)
            if V: print('  chamfer idx', flush=True)
            ch, cxy, cyx = pcu.chamfer_distance(q, r, return_index=True); ch0, cxy0, cyx0 = oracle.chamfer_distance(q, r, return_index=True, kind=kind)
            parts = {"hausdorff": h == h0, "cxy": np.array_equal(cxy, cxy0), "cyx": np.array_equal(cyx, cyx0), "chamfer": abs(float(ch) - float(ch0)) <= 1e-4 * abs(float(ch0)) + 1e-30}
            # the fused calls (no indices asked for)
            if V: print('  fused', flush=True)
            chf = pcu.chamfer_distance(q, r); st_chf = pcu.last_stats(); hf = pcu.hausdorff_distance(q, r); st_hf = pcu.last_stats()
            parts["chamfer_fused"] = abs(float(chf) - float(ch0)) <= 1e-4 * abs(float(ch0)) + 1e-30
            parts["hausdorff_fused"] = hf == h0[0]
            ok = all(parts.values())
            if not ok: tag += f" FAILED {[k_ for k_, v_ in parts.items() if not v_]} h={h} h0={h0} ch={float(ch)!r} chf={float(chf)!r} ch0={float(ch0)!r} hf={hf!r} stats_chf={st_chf} stats_hf={st_hf}"
    except Exception as e:
        ok = False; tag += f" EXC {e!r}"
    if not ok:
        bad += 1; print("MISMATCH", tag, flush=True)
print(f"fuzz: {ncases} cases, {bad} mismatches, {time.time() - t0:.1f} s", flush=True)
sys.exit(1 if bad else 0)
