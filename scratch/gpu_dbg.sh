#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
cat > /tmp/lanes.py <<'PY'
import sys, time, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import point_cloud_utils_amd as pcu
from point_cloud_utils_amd import batched
from conftest import cloud
n, npairs = 262144, 32
pairs = [(torch.from_numpy(cloud(1000 + 2 * p, n, np.float32)).cuda(), torch.from_numpy(cloud(1001 + 2 * p, n, np.float32)).cuda()) for p in range(npairs)]
ref = None
for lanes in (1, 2, 3, 4):
    f = lambda: batched._map_chunks("hausdorff", lambda p: pairs[p], list(range(npairs)), lanes)
    r = f(); f(); torch.cuda.synchronize()
    if ref is None: ref = r
    assert r == ref
    ts = []
    for _ in range(7):
        t0 = time.perf_counter(); f(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    t = float(np.median(ts))
    print("lanes/thread %2d: %.3f ms total, %.1f us per pair, %.3g q-pts/s" % (lanes, t * 1e3, t * 1e6 / npairs, npairs * 2 * n / t), flush=True)
PY
for th in 1 2 3 4; do echo "== threads $th"; PCU_HIP_BATCH_THREADS=$th python /tmp/lanes.py 2>&1 | grep -v amdgpu; done
timeout 300 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "config4 or batched" 2>&1 | tail -2
