#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "unbalanced or refit or REFIT or cluster or uneven or DEBUG_SKEW or NO_RESCALE or shifted" 2>&1 | tail -5
for c in cluster outlier gauss; do python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | cut -c1-330; done
PCU_HIP_DEBUG_SKEW=1 python bench.py --config cluster --steps 1 --warmup 1 --no-parity 2>&1 | grep -v "^{" | tail -12
