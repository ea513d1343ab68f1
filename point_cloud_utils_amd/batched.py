"""Batches of independent (x, y) cloud pairs sharded over the GPUs of one node (BASELINE config 4).

One process per GPU (`torch.distributed`; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests). Pair p is owned by
rank p mod world_size; ranks never exchange point data. The only collective is ONE all_gather of the per-pair scalars
(3 doubles per pair for Hausdorff: d, i, j; 1 for Chamfer) at the end -- a few KB over xGMI, latency-bound.
"""
import numpy as np


_POOLS = {}


def shard_pairs(n_pairs, rank, world_size):
    """Indices of the pairs owned by `rank` (round-robin, so uneven pair sizes spread evenly)."""
    return list(range(rank, n_pairs, world_size))


def _map_pairs(fn, pairs, workers):
    """fn(p) for the pairs this rank owns. With workers > 1 the calls run in a thread pool: every thread owns a
    context (stream + workspace) of its own and ctypes drops the GIL during the call, so the many short kernels
    of independent pairs interleave on the GPU instead of queueing behind each other."""
    if workers <= 1 or len(pairs) <= 1:
        return [fn(p) for p in pairs]
    from concurrent.futures import ThreadPoolExecutor
    from . import _lib
    try:
        import torch
        if torch.cuda.is_available():
            torch.cuda.synchronize()        # inputs produced on torch streams are complete before workers start
            dev = torch.cuda.current_device()
        else:
            dev = None
    except Exception:
        dev = None

    def init():
        _lib.private_streams(True)
        if dev is not None:
            import torch
            torch.cuda.set_device(dev)

    # one long-lived pool per worker count: its threads (and therefore their GPU contexts / workspaces) are reused
    ex = _POOLS.get(workers)
    if ex is None:
        ex = _POOLS[workers] = ThreadPoolExecutor(max_workers=workers, initializer=init)
    return list(ex.map(fn, pairs))


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


def batched_hausdorff(get_pair, n_pairs, squared_distances=False, max_points_per_leaf=10, op=None, group=None, workers=4):
    """Two-sided Hausdorff distance of `n_pairs` independent pairs. `get_pair(p)` returns (x, y) for pair p and is only
    called for the pairs this rank owns. Returns an (n_pairs, 3) float64 array of (d, i, j) rows, identical on all ranks.
    `op` defaults to point_cloud_utils_amd.hausdorff_distance (tests inject a CPU stand-in)."""
    import torch.distributed as dist
    if op is None:
        from . import hausdorff_distance as op
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_available() and dist.is_initialized() else (0, 1)
    def one(p):
        x, y = get_pair(p)
        d, i, j = op(x, y, return_index=True, squared_distances=squared_distances, max_points_per_leaf=max_points_per_leaf)
        return (float(d), float(i), float(j))

    rows = _map_pairs(one, shard_pairs(n_pairs, rank, world), workers)
    return _gather_rows(rows, n_pairs, 3, group)


def batched_chamfer(get_pair, n_pairs, p_norm=2, max_points_per_leaf=10, op=None, group=None, workers=4):
    """Chamfer distance of `n_pairs` independent pairs -> (n_pairs,) float64, identical on all ranks."""
    import torch.distributed as dist
    if op is None:
        from . import chamfer_distance as op
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_available() and dist.is_initialized() else (0, 1)
    def one(p):
        x, y = get_pair(p)
        return (float(op(x, y, p_norm=p_norm, max_points_per_leaf=max_points_per_leaf)),)

    rows = _map_pairs(one, shard_pairs(n_pairs, rank, world), workers)
    return _gather_rows(rows, n_pairs, 1, group)[:, 0]
