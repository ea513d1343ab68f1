#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 200 python scratch/skew.py mix_10pct_cluster gauss_s0.05 outlier_bbox mesh_samples 2>&1 | grep -v amdgpu | tail -5
timeout 250 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -x -q -m gpu -k "unbalanced or refit or overflow or config5 or surface" 2>&1 | tail -2
timeout 100 python scratch/fuzz.py 7 40 2>&1 | tail -1
