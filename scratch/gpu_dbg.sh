#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
for v in 4 0 4 5 3; do
PCU_HIP_KD_SPEC_PAIRS=$v timeout 200 python bench.py --config c3 --steps 8 --warmup 2 --no-parity --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('pairs=$v c3 %.3f ms' % d['ms_per_step'])"
done
