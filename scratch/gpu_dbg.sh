#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 200 python bench.py --config c3 --steps 8 --warmup 2 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('c3 %.3f ms' % d['ms_per_step'], json.dumps(d.get('parity'))[:60])"
timeout 250 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "lattice or tie or kd or dup or speculative or beyond" 2>&1 | tail -2
