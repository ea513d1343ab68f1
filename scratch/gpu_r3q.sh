#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 300 python scratch/occ_headline.py
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "golden or metrics or consistency" 2>&1 | tail -3
