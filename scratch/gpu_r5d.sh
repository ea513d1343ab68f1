#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TL_LINES=40 bash scratch/gpu_cfg.sh d_normals normals > gpurun_out/d_normals.txt 2>&1
TL_LINES=40 bash scratch/gpu_cfg.sh d_voxel voxel > gpurun_out/d_voxel.txt 2>&1
TL_LINES=60 bash scratch/gpu_cfg.sh d_c4 c4 > gpurun_out/d_c4.txt 2>&1
cat gpurun_out/d_normals.txt gpurun_out/d_voxel.txt; tail -70 gpurun_out/d_c4.txt
