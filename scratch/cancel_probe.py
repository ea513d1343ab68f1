"""How long do single long calls take, and how fast does pcu.cancel() end them? (round 6: the test that cancels ONE call in flight)"""
import sys, threading, time
import numpy as np
sys.path.insert(0, ".")
import point_cloud_utils_amd as pcu

def line(n, dtype, seed):
    rng = np.random.default_rng(seed)
    t = rng.random(n).astype(dtype)
    return np.ascontiguousarray(np.stack([t, t * dtype(0.5), t * dtype(0.25)], axis=1))

def probe(name, fn, delay):
    fn()                                   # warm (arena, graphs)
    t0 = time.perf_counter(); fn(); full = time.perf_counter() - t0
    req = [None]
    def canceller():
        time.sleep(delay); req[0] = time.perf_counter(); pcu.cancel()
    th = threading.Thread(target=canceller); th.start()
    t0 = time.perf_counter()
    try:
        fn(); out = "finished"
    except KeyboardInterrupt:
        out = "KeyboardInterrupt"
    t1 = time.perf_counter(); th.join()
    print(f"{name}: full {full*1e3:.1f} ms; cancel at +{delay*1e3:.0f} ms -> {out} after {(t1 - t0)*1e3:.1f} ms, {(t1 - req[0])*1e3:.1f} ms after the request", flush=True)
    fn()

rng = np.random.default_rng(3)
q = rng.random((237_000, 3)); r = line(267_000, np.float64, 4)
probe("237k f64 queries k=16 vs 267k-point LINE", lambda: pcu.k_nearest_neighbors(q, r, 16), 0.2)
q2 = rng.random((200_000, 3), dtype=np.float32); r2 = rng.random((200_000, 3), dtype=np.float32)
probe("200k vs 200k f32, k=200 (kd traversal for every query)", lambda: pcu.k_nearest_neighbors(q2, r2, 200), 0.05)
dup = np.repeat(rng.random((300_000, 3), dtype=np.float32), 3, axis=0)
probe("900k-point triplicated cloud against itself, k=4 (every query tied)", lambda: pcu.k_nearest_neighbors(dup, dup, 4), 0.05)
probe("same through hausdorff(return_index)", lambda: pcu.hausdorff_distance(dup, dup[::-1].copy(), return_index=True), 0.02)
probe("same through chamfer(return_index)", lambda: pcu.chamfer_distance(dup, dup[::-1].copy(), return_index=True), 0.02)
