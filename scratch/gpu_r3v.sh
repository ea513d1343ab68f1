#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_multirank.py -m gpu -q -x 2>&1 | tail -15
python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-300
