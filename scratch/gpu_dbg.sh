#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 200 python scratch/skew.py mix_10pct_cluster 2>&1 | grep -v amdgpu | tail -1
timeout 250 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unbalanced or refit" 2>&1 | tail -1
timeout 100 python scratch/fuzz.py 9 40 2>&1 | tail -1
