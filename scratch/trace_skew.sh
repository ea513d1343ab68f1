#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}; R=$PWD
cd /tmp && export TMPDIR=/tmp
for c in gauss_s0.05 sphere_surface; do
rm -rf $R/gpurun_out/ts_$c
PCU_HIP_DEBUG_SKEW=1 timeout 200 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/ts_$c -- python $R/scratch/skew.py $c 2>&1 | grep "$c\|skew\]" | tail -3
python $R/profiles/summarize_rocprof.py $R/gpurun_out/ts_$c/*/*_results.db | cut -c1-135 | sed -n 3,14p
done
