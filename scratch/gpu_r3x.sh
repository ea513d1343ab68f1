#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
cd /tmp
timeout 200 rocprofv3 --kernel-trace -d /tmp/clt -- python $ROOT/bench.py --config cluster --steps 3 --warmup 2 --no-parity > /tmp/clt.log 2>&1
db=$(find /tmp/clt -name "*results.db" | head -1)
python $ROOT/scratch/timeline.py $db k_bbox_partial > $ROOT/gpurun_out/cluster_timeline.txt 2>&1
cd $ROOT
PCU_HIP_DEBUG_SKEW=1 timeout 100 python bench.py --config cluster --steps 1 --warmup 1 --no-parity 2>&1 | grep -v "^{" | tail -14
