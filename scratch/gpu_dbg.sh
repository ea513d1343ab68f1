#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for cfg in "" "--config c3" "--config normals"; do timeout 300 python bench.py $cfg --steps 60 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$cfg %.4g %s %.4f ms' % (d['value'], d['unit'], d['ms_per_step']))"; done
