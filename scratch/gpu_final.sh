#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
timeout 900 python -m pytest tests -m gpu -q -x --durations=6 2>&1 | tail -12
bash profiles/collect.sh r04 2>&1 | tail -16 | cut -c1-250
