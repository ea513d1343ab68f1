#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 300 python scratch/skew.py 2>&1 | grep -v amdgpu.ids | tail -9
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
for cfg in "--config c2" "--config c3" "--config c4" "--config normals"; do timeout 300 python bench.py $cfg --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | cut -c1-230; done
