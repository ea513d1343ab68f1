#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
python bench.py --steps 50 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-400
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/c3t -- python $ROOT/bench.py --config c3 --steps 3 --warmup 2 --no-parity > /tmp/c3t.log 2>&1
db=$(find /tmp/c3t -name "*results.db" | head -1)
python $ROOT/scratch/timeline.py $db k_bbox_partial > $ROOT/gpurun_out/c3_timeline.txt 2>&1
tail -30 $ROOT/gpurun_out/c3_timeline.txt
cd $ROOT
python - <<'PY'
import numpy as np, torch, point_cloud_utils_amd as pcu
n=4000000
x=torch.from_numpy(np.random.default_rng(1).random((n,3),dtype=np.float32)).cuda(); y=torch.from_numpy(np.random.default_rng(2).random((n,3),dtype=np.float32)).cuda()
pcu.k_nearest_neighbors(x,y,16); pcu.set_timing(2); pcu.k_nearest_neighbors(x,y,16); print(pcu.last_stats())
PY
