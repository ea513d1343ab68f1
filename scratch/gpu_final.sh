#!/bin/bash
# Round artefacts: full GPU test suite (timed), then profiles/collect.sh.
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
python -c "import torch"
TAG=${1:-r02}
( time timeout 1500 python -m pytest tests/ -x -q -m gpu ) > gpurun_out/${TAG}_pytest_gpu.log 2>&1
tail -6 gpurun_out/${TAG}_pytest_gpu.log
timeout 1500 bash profiles/collect.sh $TAG > gpurun_out/${TAG}_collect.log 2>&1
tail -c 3000 gpurun_out/${TAG}_collect.log
