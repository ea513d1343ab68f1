#!/bin/bash
# round 6: k_search_runs on the coordinates-only stream (c3 A/B against the Pt4 build), shared grid for k = 1 k_nearest_neighbors (c2 A/B), parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py tests/test_gpu_configs.py -m gpu -x -q -k "not switch" 2>&1 | tail -4 ) | tee gpurun_out/r6h_tests.log
STEPS=8 bash scratch/gpu_cfg_ab.sh c3 3 xyz= pt4=PCU_HIP_LIBRARY=$GRAFT_REPO_ROOT/scratch/variants/libpcu_hip_runs_pt4.so | tee gpurun_out/r6h_ab.txt
STEPS=40 bash scratch/gpu_cfg_ab.sh c2 3 shared= own=PCU_HIP_NO_SHARED_GRID=1 | tee -a gpurun_out/r6h_ab.txt
