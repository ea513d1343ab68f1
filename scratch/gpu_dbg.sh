#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
cd /tmp && export TMPDIR=/tmp
PCU_HIP_NO_FUSE=1 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/t5_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-parity > $GRAFT_REPO_ROOT/gpurun_out/t5_trace.log 2>&1
cd $GRAFT_REPO_ROOT
python profiles/summarize_rocprof.py $(ls gpurun_out/t5_trace/*/*.db | head -1) | cut -c1-160 | head -14
