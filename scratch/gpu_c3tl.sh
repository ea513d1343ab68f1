#!/bin/bash
# config 3 kernel timeline of one call (profiles/r04_c3_timeline.txt is made from this)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/c3tl_trace -- python $GRAFT_REPO_ROOT/bench.py --config c3 --steps 3 --warmup 2 --no-parity --no-cpu-baseline --no-kernel-events > $GRAFT_REPO_ROOT/gpurun_out/c3tl.log 2>&1
cd $GRAFT_REPO_ROOT
python scratch/timeline.py $(find gpurun_out/c3tl_trace -name "*results.db" | head -1) > gpurun_out/c3tl_timeline.txt
rm -rf gpurun_out/c3tl_trace
tail -5 gpurun_out/c3tl_timeline.txt
