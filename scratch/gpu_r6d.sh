#!/bin/bash
# round 6: counters of the staged (brick) pass against k_search1_flat on the same shared grid, and of the flat kernel on per-cloud grids
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
bash scratch/gpu_sq.sh r6d_brick k_search1 PCU_HIP_BRICK=1 > /dev/null 2>&1
bash scratch/gpu_sq.sh r6d_flat_shared k_search1 > /dev/null 2>&1
bash scratch/gpu_sq.sh r6d_flat_own k_search1 PCU_HIP_NO_SHARED_GRID=1 > /dev/null 2>&1
for t in r6d_brick r6d_flat_shared r6d_flat_own; do echo "== $t"; cat gpurun_out/${t}_sq.txt; done
bash scratch/ab.sh 3 shared= own=PCU_HIP_NO_SHARED_GRID=1 | tee gpurun_out/r6d_ab.txt
