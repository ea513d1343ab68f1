#!/bin/bash
# round 6 mid-round check: full -m gpu suite + 700 fuzz cases on the build with shared grids, handed-down layouts, in-kernel cancellation
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -8 ) > gpurun_out/r6j_tests.log 2>&1
( time timeout 900 python scratch/fuzz.py 601 350 ) > gpurun_out/r6j_fuzz_601.log 2>&1
( time timeout 900 python scratch/fuzz.py 602 350 ) > gpurun_out/r6j_fuzz_602.log 2>&1
tail -8 gpurun_out/r6j_tests.log; tail -4 gpurun_out/r6j_fuzz_601.log gpurun_out/r6j_fuzz_602.log
