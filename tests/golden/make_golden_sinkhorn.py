#!/usr/bin/env python3
"""Generates tests/golden/sinkhorn.npz from the reference's own pure-numpy module (/root/reference/point_cloud_utils/
_sinkhorn.py, loaded by file path -- the package itself cannot be imported without its compiled extension). Run where
/root/reference exists:  python tests/golden/make_golden_sinkhorn.py"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
import oracle  # noqa: E402


def main():
    ref = oracle.reference_sinkhorn_module()
    assert ref is not None, "needs /root/reference"
    rng = np.random.default_rng(20250924)
    out = {}
    for tag, dt in (("f32", np.float32), ("f64", np.float64)):
        a = rng.random((96, 3)).astype(dt); b = rng.random((80, 3)).astype(dt)
        out[f"a_{tag}"] = a; out[f"b_{tag}"] = b
        for p in (None, 1, np.inf, 3):
            out[f"M_{tag}_p{p}"] = ref.pairwise_distances(a, b, p)
        M = ref.pairwise_distances(a, b)
        wa = np.full(96, 1.0 / 96, dt); wb = np.full(80, 1.0 / 80, dt)
        out[f"P_{tag}"] = ref.sinkhorn(wa, wb, M, eps=1e-2, max_iters=60)
        ab = rng.random((3, 40, 5)).astype(dt); bb = rng.random((3, 50, 5)).astype(dt)          # batched, d = 5, non-uniform weights
        Mb = ref.pairwise_distances(ab, bb)
        wab = rng.random((3, 40)).astype(dt) + dt(0.5); wab /= wab.sum(1, keepdims=True)
        wbb = rng.random((3, 50)).astype(dt) + dt(0.5); wbb /= wbb.sum(1, keepdims=True)
        out[f"ab_{tag}"] = ab; out[f"bb_{tag}"] = bb; out[f"wab_{tag}"] = wab; out[f"wbb_{tag}"] = wbb; out[f"Mb_{tag}"] = Mb
        out[f"Pb_{tag}"] = ref.sinkhorn(wab, wbb, Mb, eps=5e-2, max_iters=100, stop_thresh=1e-4)
    p = rng.random((64, 3)); q = rng.random((48, 3))
    emd, P = ref.earth_movers_distance(p, q, eps=1e-2)
    out["emd_p"] = p; out["emd_q"] = q; out["emd"] = np.float64(emd); out["emd_P"] = P
    np.savez_compressed(os.path.join(HERE, "sinkhorn.npz"), **out)
    print("wrote sinkhorn.npz:", len(out), "arrays")


if __name__ == "__main__":
    main()
