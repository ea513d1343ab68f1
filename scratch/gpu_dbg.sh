#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
echo "=== only unbalanced"; timeout 120 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "unbalanced" 2>&1 | grep -v "^  File \"/usr" | tail -12
echo "=== ties + unbalanced"; timeout 120 python -u -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "ties or unbalanced" 2>&1 | grep -v "^  File \"/usr" | tail -12
