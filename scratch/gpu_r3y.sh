#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x 2>&1 | tail -4
for c in cluster gauss c4; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | cut -c1-260; done
