#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
for v in 4 2; do echo "COARSE_AT=$v"; PCU_HIP_COARSE_AT=$v timeout 200 python scratch/skew.py gauss_s0.05 outlier_bbox 2>&1 | grep -v amdgpu | tail -2; done
