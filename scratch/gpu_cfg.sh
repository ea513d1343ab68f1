#!/bin/bash
# one config of bench.py: the line, then a rocprofv3 kernel trace + timeline of the last call: gpu_cfg.sh <tag> <config> [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=$1; CFG=$2; shift; shift
for kv in "$@"; do export "$kv"; done
python bench.py --config $CFG --steps 6 --warmup 2 --no-parity --no-cpu-baseline 2>&1 | tail -1 | cut -c1-400 | tee gpurun_out/${TAG}_line.txt
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python $GRAFT_REPO_ROOT/bench.py --config $CFG --steps 3 --warmup 2 --no-parity --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/timeline.py $(find gpurun_out/${TAG}_trace -name "*results.db" | head -1) > gpurun_out/${TAG}_timeline.txt
rm -rf gpurun_out/${TAG}_trace
tail -${TL_LINES:-90} gpurun_out/${TAG}_timeline.txt | cut -c1-110
