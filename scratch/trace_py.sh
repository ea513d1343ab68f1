#!/bin/bash
# rocprofv3 kernel trace of an arbitrary python script, per-kernel summary only: trace_py.sh <tag> <script> [args...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=$1; shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace -- python "$@" > $GRAFT_REPO_ROOT/gpurun_out/${TAG}_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(find gpurun_out/${TAG}_trace -name "*results.db" | head -1) | cut -c1-170 | head -${TRACE_LINES:-25} > gpurun_out/${TAG}_trace_summary.txt
rm -rf gpurun_out/${TAG}_trace
cat gpurun_out/${TAG}_trace_summary.txt
