#!/bin/bash
# rocprofv3 kernel trace of bench.py + profiles/summarize_rocprof.py: gpu_trace.sh <tag> [ENV=VALUE ...] (the summary only is kept)
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-t}; shift
for kv in "$@"; do export "$kv"; done
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-parity --no-configs > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/${TAG}_trace -name "*results.db" | head -1) | cut -c1-170 > gpurun_out/${TAG}_trace_summary.txt
rm -rf gpurun_out/${TAG}_trace
cat gpurun_out/${TAG}_trace_summary.txt
