#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 120 python scratch/dbg_normals.py 2>&1 | grep -v amdgpu.ids
timeout 200 python bench.py --config c5 --steps 10 --warmup 3 2>/dev/null | cut -c1-330
