#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 600 python -u -m pytest tests/test_gpu_sinkhorn.py -x -q -m gpu 2>&1 | tail -30
