"""CPU test of the N>1 path: world_size-2 gloo processes shard a batch of independent pairs and gather the scalars
with one all_gather (point_cloud_utils_amd/batched.py). The per-pair operator is the oracle here (no GPU on this
box); on the GPU box the same driver calls the HIP path (tests/test_gpu_parity.py::test_batched_single_rank)."""
import os
import socket
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_pairs, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    import oracle
    from point_cloud_utils_amd import batched
    dist.init_process_group("gloo", rank=rank, world_size=world)
    called = []

    def get_pair(p):
        called.append(p)
        rng = np.random.default_rng(1000 + 2 * p), np.random.default_rng(1001 + 2 * p)
        return rng[0].random((300 + 10 * p, 3), dtype=np.float32), rng[1].random((250, 3), dtype=np.float32)

    hd = batched.batched_hausdorff(get_pair, n_pairs, op=lambda x, y, **kw: oracle.hausdorff_distance(x, y, **kw))
    ch = batched.batched_chamfer(get_pair, n_pairs, op=lambda x, y, **kw: oracle.chamfer_distance(x, y, **kw))
    q.put((rank, sorted(set(called)), hd, ch))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("n_pairs", [5, 8])
def test_gloo_world2_shards_and_gathers(n_pairs):
    import torch.multiprocessing as mp
    import oracle
    oracle.build()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs: p.join(60)
    res.sort(key=lambda t: t[0])
    # every pair computed exactly once, round-robin
    assert res[0][1] == list(range(0, n_pairs, 2)) and res[1][1] == list(range(1, n_pairs, 2))
    # both ranks hold the same, complete, pair-ordered result, equal to a serial run
    assert np.array_equal(res[0][2], res[1][2]) and np.array_equal(res[0][3], res[1][3])
    for p in range(n_pairs):
        x = np.random.default_rng(1000 + 2 * p).random((300 + 10 * p, 3), dtype=np.float32)
        y = np.random.default_rng(1001 + 2 * p).random((250, 3), dtype=np.float32)
        d, i, j = oracle.hausdorff_distance(x, y, return_index=True)
        assert tuple(res[0][2][p]) == (d, i, j)
        assert res[0][3][p] == float(oracle.chamfer_distance(x, y))


def test_shard_pairs_partition():
    from point_cloud_utils_amd.batched import shard_pairs
    for n in (1, 7, 256):
        for w in (1, 2, 4, 8):
            allp = sorted(p for r in range(w) for p in shard_pairs(n, r, w))
            assert allp == list(range(n))
            assert max(len(shard_pairs(n, r, w)) for r in range(w)) - min(len(shard_pairs(n, r, w)) for r in range(w)) <= 1
