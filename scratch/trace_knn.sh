#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
cd /tmp && export TMPDIR=/tmp
for k in 8; do
rm -rf $R/gpurun_out/tk$k
timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/tk$k -- python $R/scratch/knn_k.py 1000000 $k 2>&1 | grep "n=1000000" | cut -c1-60
python $R/profiles/summarize_rocprof.py $R/gpurun_out/tk$k/*/*_results.db | cut -c1-130 | sed -n 3,24p
done
