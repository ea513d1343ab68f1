#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 300 python scratch/skew.py 2>&1 | grep -v amdgpu.ids | tail -9
timeout 900 python -m pytest tests -x -q -m gpu 2>&1 | tail -4
timeout 300 python scratch/fuzz.py 2>&1 | tail -3
