#!/bin/bash
# round 6, call 1: cancellation rework (generation counter, error codes through the ABI, SIGINT rule) + probe of single long calls
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time timeout 600 python scratch/cancel_probe.py ) > gpurun_out/r6a_probe.log 2>&1
( time timeout 900 python -m pytest tests/test_gpu_cancel.py tests/test_gpu_parity.py -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r6a_tests.log 2>&1
( timeout 300 python bench.py --no-configs 2>&1 | tail -2 ) > gpurun_out/r6a_bench.log 2>&1
cat gpurun_out/r6a_probe.log; tail -8 gpurun_out/r6a_tests.log; tail -c 1500 gpurun_out/r6a_bench.log
