#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-t}
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(ls gpurun_out/${TAG}_trace/*/*.db | head -1) | cut -c1-160
