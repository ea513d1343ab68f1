#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 300 python scratch/skew.py 2>&1 | grep -v amdgpu.ids
PCU_HIP_DEBUG_SKEW=1 timeout 100 python scratch/skew.py mix_10pct_cluster 2>&1 | grep -v amdgpu.ids | tail -6
