#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 600 python -m pytest tests/test_gpu_sinkhorn.py -x -q -m gpu 2>&1 | tail -2
for r in 4 8 16; do
  PCU_HIP_SINK_ROWS=$r timeout 300 python bench.py --config sinkhorn --steps 10 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('rows $r: %.3f ms' % d['ms_per_step'], d.get('parity'))"
done
