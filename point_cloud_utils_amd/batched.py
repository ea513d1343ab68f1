"""Batches of independent (x, y) cloud pairs sharded over the GPUs of one node (BASELINE config 4).

One process per GPU (`torch.distributed`; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests). Pair p is owned by
rank p mod world_size; ranks never exchange point data. The only collective is ONE all_gather of the per-pair scalars
(3 doubles per pair for Hausdorff: d, i, j; 1 for Chamfer) at the end -- a few KB over xGMI, latency-bound.

Inside a rank the pairs go through the library's batch entry points (`pcu_hip_hausdorff_batch_*` /
`pcu_hip_chamfer_batch_*`, include/pcu_hip.h): one C call per chunk of pairs, several pairs in flight on internal streams,
no Python threads. The reference has no batched call (a user loops over `hausdorff_distance`); per-pair results are
identical to the single-pair functions.
"""
import ctypes

import numpy as np

CHUNK = 64          # pairs handed to one library call (bounds how many input clouds are alive at once)


def shard_pairs(n_pairs, rank, world_size):
    """Indices of the pairs owned by `rank` (round-robin, so uneven pair sizes spread evenly)."""
    return list(range(rank, n_pairs, world_size))


def _gather_rows(local_rows, n_pairs, width, group=None):
    """local_rows: (n_local, width) float64 in shard order -> (n_pairs, width) in pair order on every rank."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()):
        return np.asarray(local_rows, dtype=np.float64).reshape(n_pairs, width)
    world, rank = dist.get_world_size(group), dist.get_rank(group)
    per = (n_pairs + world - 1) // world
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    buf = torch.full((per, width), float("nan"), dtype=torch.float64, device=dev)
    if len(local_rows):
        buf[:len(local_rows)] = torch.as_tensor(np.asarray(local_rows, dtype=np.float64).reshape(-1, width), device=dev)
    out = [torch.empty_like(buf) for _ in range(world)]
    dist.all_gather(out, buf, group=group)              # the job's only collective
    res = np.empty((n_pairs, width), dtype=np.float64)
    for r in range(world):
        idx = shard_pairs(n_pairs, r, world)
        res[idx] = out[r][:len(idx)].cpu().numpy()
    return res


def _rank_world(group):
    import torch.distributed as dist
    if dist.is_available() and dist.is_initialized():
        return dist.get_rank(group), dist.get_world_size(group)
    return 0, 1


def _run_chunk(kind, pairs, lanes, squared=False, p_norm=2.0, max_points_per_leaf=10):
    """One library call for a list of (x, y) pairs. Returns a list of result rows."""
    from . import _lib, _check_pair, _Dev, _KNN_ZERO, _KNN_DIM, _HD_ZERO, _HD_DIM, _record, Stats
    devs = []
    for x, y in pairs:
        if kind == "hausdorff":
            _check_pair(x, y, "source", "target", _HD_ZERO, _HD_DIM)
        else:
            _check_pair(x, y, "query_points", "dataset_points", _KNN_ZERO, _KNN_DIM)
        devs.append(_Dev(x, y))          # contiguous buffers (kept alive until the call returns), pointers, device, stream
    d0 = devs[0]
    for d in devs:
        if d.suffix != d0.suffix or d.torch != d0.torch or d.device != d0.device:
            raise ValueError("all pairs of a batch must share dtype, device and array kind (numpy / torch)")
    n = len(devs)
    vp, i64 = ctypes.c_void_p, ctypes.c_int64
    xs = (vp * n)(*[d.pa for d in devs]); ys = (vp * n)(*[d.pb for d in devs])
    nxs = (i64 * n)(*[int(d.a.shape[0]) for d in devs]); nys = (i64 * n)(*[int(d.b.shape[0]) for d in devs])
    L = _lib.lib()
    L.pcu_hip_ctx_set_batch_lanes(d0.ctx, int(lanes))
    st = Stats()
    flags = d0.flags | (_lib.SQUARED if squared else 0)
    if kind == "hausdorff":
        od = np.zeros((n, 2), dtype=d0.np_dtype); oi = np.zeros((n, 2), dtype=np.int64); oj = np.zeros((n, 2), dtype=np.int64)
        rc = getattr(L, "pcu_hip_hausdorff_batch_" + d0.suffix)(d0.ctx, n, xs, nxs, ys, nys, int(max_points_per_leaf), od.ctypes.data,
                                                                oi.ctypes.data, oj.ctypes.data, flags, d0.stream, ctypes.addressof(st))
        _lib.check(rc)
        _record(st)
        rows = []
        for p in range(n):       # point_cloud_utils/__init__.py:69-81 on Python floats, exactly as hausdorff_distance does
            hxy, ix1, iy1 = float(od[p, 0]), int(oi[p, 0]), int(oj[p, 0])
            hyx, iy2, ix2 = float(od[p, 1]), int(oi[p, 1]), int(oj[p, 1])
            rows.append((max(hxy, hyx), float(ix1), float(iy1)) if hxy > hyx else (max(hxy, hyx), float(ix2), float(iy2)))
        return rows
    means = np.zeros((n, 2), dtype=np.float64)
    rc = getattr(L, "pcu_hip_chamfer_batch_" + d0.suffix)(d0.ctx, n, xs, nxs, ys, nys, float(p_norm), int(max_points_per_leaf),
                                                          means.ctypes.data, flags, d0.stream, ctypes.addressof(st))
    _lib.check(rc)
    _record(st)
    # __init__.py:112-115: both means are scalars of the input dtype; their sum, in that dtype, is the result
    return [(float(d0.np_dtype(means[p, 1]) + d0.np_dtype(means[p, 0])),) for p in range(n)]


def _map_chunks(kind, get_pair, mine, lanes, **kw):
    rows = []
    for c0 in range(0, len(mine), CHUNK):
        pairs = [get_pair(p) for p in mine[c0:c0 + CHUNK]]
        rows.extend(_run_chunk(kind, pairs, lanes, **kw))
    return rows


def batched_hausdorff(get_pair, n_pairs, squared_distances=False, max_points_per_leaf=10, op=None, group=None, workers=4):
    """Two-sided Hausdorff distance of `n_pairs` independent pairs. `get_pair(p)` returns (x, y) for pair p and is only
    called for the pairs this rank owns. Returns an (n_pairs, 3) float64 array of (d, i, j) rows -- what
    hausdorff_distance(x, y, return_index=True) returns for each pair -- identical on all ranks.
    `workers`: pairs kept in flight on the GPU. `op`: a stand-in for hausdorff_distance (the CPU tests of the sharding
    logic inject one); by default the pairs go through the library's batch entry point."""
    rank, world = _rank_world(group)
    mine = shard_pairs(n_pairs, rank, world)
    if op is not None:
        rows = []
        for p in mine:
            x, y = get_pair(p)
            d, i, j = op(x, y, return_index=True, squared_distances=squared_distances, max_points_per_leaf=max_points_per_leaf)
            rows.append((float(d), float(i), float(j)))
    else:
        rows = _map_chunks("hausdorff", get_pair, mine, workers, squared=squared_distances, max_points_per_leaf=max_points_per_leaf)
    return _gather_rows(rows, n_pairs, 3, group)


def batched_chamfer(get_pair, n_pairs, p_norm=2, max_points_per_leaf=10, op=None, group=None, workers=4):
    """Chamfer distance of `n_pairs` independent pairs -> (n_pairs,) float64, identical on all ranks."""
    rank, world = _rank_world(group)
    mine = shard_pairs(n_pairs, rank, world)
    if op is not None:
        rows = []
        for p in mine:
            x, y = get_pair(p)
            rows.append((float(op(x, y, p_norm=p_norm, max_points_per_leaf=max_points_per_leaf)),))
    else:
        rows = _map_chunks("chamfer", get_pair, mine, workers, p_norm=p_norm, max_points_per_leaf=max_points_per_leaf)
    return _gather_rows(rows, n_pairs, 1, group)[:, 0]
