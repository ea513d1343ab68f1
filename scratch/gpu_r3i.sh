#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
timeout 240 python -m pytest tests/test_gpu_parity.py -m gpu -q -x --durations=12 2>&1 | tail -22
