#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "balanced or headline" 2>&1 | tail -5
for i in 1 2; do timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | cut -c1-220; done
