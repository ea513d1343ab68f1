"""Multi-rank readiness on ONE GPU (-m gpu): two processes (torch.distributed, gloo rendezvous on 127.0.0.1) share cuda:0 and run
the batched drivers through the HIP batch entry points (pcu_hip_hausdorff_batch_* / pcu_hip_chamfer_batch_*) -- no `op=`
stand-in: sharding (pair p -> rank p mod 2), the HIP path per pair, and the one all_gather of the scalars run together.
RCCL refuses two ranks on one device, so the gather goes over gloo here; with one rank per GPU `bench.py --gpus N` uses the
same driver over backend "nccl" (= RCCL). No scaling curve has been measured yet (the driver had no multi-GPU node)."""
import os
import socket
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _pair(p, n):
    return (np.random.default_rng(1000 + 2 * p).random((n, 3), dtype=np.float32),
            np.random.default_rng(1001 + 2 * p).random((n - 1000 * (p % 3), 3), dtype=np.float32))


def _worker(rank, world, port, n_pairs, n, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      HSA_ENABLE_IPC_MODE_LEGACY="0")
    try:
        import torch
        import torch.distributed as dist
        import point_cloud_utils_amd as pcu
        from point_cloud_utils_amd import batched
        torch.cuda.set_device(0)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        called = []

        def host_pair(p):
            called.append(p)
            return _pair(p, n)

        dev_pairs = {}

        def dev_pair(p):             # device-resident pairs (the benchmarked mode): uploaded by the owning rank only
            if p not in dev_pairs:
                x, y = _pair(p, n)
                dev_pairs[p] = (torch.from_numpy(x).cuda(), torch.from_numpy(y).cuda())
            return dev_pairs[p]

        hd = batched.batched_hausdorff(host_pair, n_pairs)
        ch = batched.batched_chamfer(host_pair, n_pairs)
        hd2 = batched.batched_hausdorff(dev_pair, n_pairs)
        ch2 = batched.batched_chamfer(dev_pair, n_pairs)
        single = {p: pcu.hausdorff_distance(*_pair(p, n), return_index=True) for p in sorted(set(called))[:2]}
        q.put((rank, sorted(set(called)), hd, ch, hd2, ch2, single, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception as e:          # surfaced by the parent
        import traceback
        q.put((rank, [], None, None, None, None, None, traceback.format_exc()))


def test_two_ranks_share_one_gpu_through_the_hip_batch_entry_points():
    import torch.multiprocessing as mp
    import oracle
    oracle.build()
    kind = "ref" if oracle.have_ref() else "port"
    n_pairs, n = 7, 60_000
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, n_pairs, n, q)) for r in range(2)]
    for p in procs: p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs: p.join(120)
    res.sort(key=lambda t: t[0])
    for r in res:
        assert r[7] is None, r[7]
    assert res[0][1] == list(range(0, n_pairs, 2)) and res[1][1] == list(range(1, n_pairs, 2))      # every pair computed once
    for a, b in zip(res[0][2:6], res[1][2:6]):
        assert np.array_equal(a, b)                                                                 # rank-identical, complete
    assert np.array_equal(res[0][2], res[0][4]) and np.array_equal(res[0][3], res[0][5])            # host pairs == device pairs
    for p in range(n_pairs):
        x, y = _pair(p, n)
        assert tuple(res[0][2][p]) == tuple(float(v) for v in oracle.hausdorff_distance(x, y, return_index=True, kind=kind))
        ch0 = float(oracle.chamfer_distance(x, y, kind=kind))
        assert abs(res[0][3][p] - ch0) <= 1e-4 * ch0
    for r in res:
        for p, h in r[6].items():
            assert tuple(float(v) for v in h) == tuple(res[0][2][p])                                # batch entry point == single call


@pytest.mark.parametrize("config", ["headline", "c4"])
def test_bench_multi_rank_launch_rehearsal(config):
    """bench.py exactly as the driver launches it for N = 2 (`python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr
    127.0.0.1 ... bench.py --gpus 2`), except that both ranks share the test box's one GPU and the scalar collectives go over gloo
    (PCU_BENCH_SHARE_GPU=1; RCCL refuses two ranks on one device). Checks the contract of the one JSON line rank 0 prints: n_gpus, whole-job
    value = units of all ranks / max-over-ranks time, weak scaling, parity of the timed pairs."""
    import json
    import subprocess
    small = ["--points", "200000"] if config == "headline" else []
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "6", "--warmup", "2",
           "--no-cpu-baseline", "--config", config] + small
    r = subprocess.run(cmd, env=dict(os.environ, PCU_BENCH_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0"), capture_output=True, text=True,
                       timeout=300, cwd=ROOT)
    assert r.returncode == 0, (r.stdout[-2000:], r.stderr[-3000:])
    lines = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 only
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 6 and d["warmup"] == 2 and d["scaling"] == "weak" and d["higher_is_better"] is True
    assert "REHEARSAL" in d["config"]["collectives"]
    units = 2 * 200000 * 2 if config == "headline" else 32 * 2 * 262144 * 2          # query-points per step over both ranks
    assert abs(d["value"] - units / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]
    par = d["parity"]
    assert par and all(v is not False for v in par.values()), par
