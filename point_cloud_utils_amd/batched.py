"""Batches of independent (x, y) cloud pairs sharded over the GPUs of one node (BASELINE config 4).

One process per GPU (`torch.distributed`; backend "nccl" = RCCL on ROCm, "gloo" in the CPU tests). Pair p is owned by
rank p mod world_size; ranks never exchange point data. The only collective is ONE all_gather of the per-pair scalars
(3 doubles per pair for Hausdorff: d, i, j; 1 for Chamfer) at the end -- a few KB over xGMI, latency-bound.
"""
import numpy as np


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


def batched_hausdorff(get_pair, n_pairs, squared_distances=False, max_points_per_leaf=10, op=None, group=None):
    """Two-sided Hausdorff distance of `n_pairs` independent pairs. `get_pair(p)` returns (x, y) for pair p and is only
    called for the pairs this rank owns. Returns an (n_pairs, 3) float64 array of (d, i, j) rows, identical on all ranks.
    `op` defaults to point_cloud_utils_amd.hausdorff_distance (tests inject a CPU stand-in)."""
    import torch.distributed as dist
    if op is None:
        from . import hausdorff_distance as op
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_available() and dist.is_initialized() else (0, 1)
    rows = []
    for p in shard_pairs(n_pairs, rank, world):
        x, y = get_pair(p)
        d, i, j = op(x, y, return_index=True, squared_distances=squared_distances, max_points_per_leaf=max_points_per_leaf)
        rows.append((float(d), float(i), float(j)))
    return _gather_rows(rows, n_pairs, 3, group)


def batched_chamfer(get_pair, n_pairs, p_norm=2, max_points_per_leaf=10, op=None, group=None):
    """Chamfer distance of `n_pairs` independent pairs -> (n_pairs,) float64, identical on all ranks."""
    import torch.distributed as dist
    if op is None:
        from . import chamfer_distance as op
    rank, world = (dist.get_rank(group), dist.get_world_size(group)) if dist.is_available() and dist.is_initialized() else (0, 1)
    rows = []
    for p in shard_pairs(n_pairs, rank, world):
        x, y = get_pair(p)
        rows.append((float(op(x, y, p_norm=p_norm, max_points_per_leaf=max_points_per_leaf)),))
    return _gather_rows(rows, n_pairs, 1, group)[:, 0]
