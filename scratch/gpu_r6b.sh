#!/bin/bash
# round 6, call 2: single-call cancel tests (piecewise traversal launch), normals thin-neighbourhood tests, sweep with ROUND = 6 seeds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time timeout 600 python scratch/cancel_probe.py ) > gpurun_out/r6b_probe.log 2>&1
( time timeout 1200 python -m pytest tests/test_gpu_cancel.py tests/test_gpu_normals.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -15 ) > gpurun_out/r6b_tests.log 2>&1
cat gpurun_out/r6b_probe.log; tail -15 gpurun_out/r6b_tests.log
