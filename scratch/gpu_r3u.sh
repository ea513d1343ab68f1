#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -4
cd /tmp
timeout 300 rocprofv3 --kernel-trace -d /tmp/c3t -- python $ROOT/bench.py --config c3 --steps 3 --warmup 2 --no-parity > /tmp/c3t.log 2>&1
db=$(find /tmp/c3t -name "*results.db" | head -1)
python $ROOT/scratch/timeline.py $db k_bbox_partial > $ROOT/gpurun_out/c3_timeline.txt 2>&1
tail -3 $ROOT/gpurun_out/c3_timeline.txt
