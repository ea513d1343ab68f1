#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "unbalanced or refit or cluster or uneven or shifted or slab or golden or poisoned or beyond" 2>&1 | tail -4
for c in cluster gauss outlier; do timeout 200 python bench.py --config $c --steps 10 --warmup 3 2>/dev/null | cut -c1-260; done
timeout 100 python bench.py --steps 30 --warmup 5 --no-cpu-baseline 2>/dev/null | cut -c1-220
