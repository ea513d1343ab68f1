#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 100 python scratch/case283.py 2>&1 | grep "GPU\|equal"
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x 2>&1 | tail -2
timeout 60 python bench.py --config gauss --steps 10 --warmup 3 2>/dev/null | cut -c1-230
