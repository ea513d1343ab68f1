#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 200 python bench.py --config c3 --steps 6 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('c3 %.3f ms' % d['ms_per_step'], json.dumps(d.get('parity'))[:200])"
timeout 200 python bench.py --steps 60 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('headline %.4f ms' % d['ms_per_step'], d.get('parity'))"
timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "lattice or tie or sweep or unbalanced" 2>&1 | tail -2
