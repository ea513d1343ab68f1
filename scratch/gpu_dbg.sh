#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
for v in 0 1 0 1; do
if [ $v = 1 ]; then export PCU_HIP_FUSED_GRID=1; else unset PCU_HIP_FUSED_GRID; fi
timeout 200 python bench.py --config c4 --steps 20 --warmup 3 --no-parity --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('FUSED_GRID=$v c4 %.3f ms per 32 pairs = %.1f us per pair' % (d['ms_per_step'], d['ms_per_step']*1000/32))"
done
