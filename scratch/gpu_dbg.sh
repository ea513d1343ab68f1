#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
python -c "import torch"
for c in gauss_s0.05 outlier_bbox; do
 (cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/skew2_$c -- python $GRAFT_REPO_ROOT/scratch/skew.py $c > $GRAFT_REPO_ROOT/gpurun_out/skew2_$c.log 2>&1)
 tail -1 gpurun_out/skew2_$c.log
done
