#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 120 python scratch/case5.py 5 2>&1 | grep "GPU"
timeout 120 python scratch/case5.py 7 2>&1 | grep "GPU\|case"
timeout 60 python bench.py --config cluster --steps 10 --warmup 3 --no-parity 2>/dev/null | cut -c1-260
