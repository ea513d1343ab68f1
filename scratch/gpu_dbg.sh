#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 300 python scratch/skew.py 2>&1 | tail -6
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "overflow" 2>&1 | tail -5
