#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 900 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "headline or fused or config4 or config2 or dense or unbalanced or golden or metrics or ties or sweep or dataset_index" 2>&1 | tail -3
for v in 0 1; do
  if [ $v = 1 ]; then export PCU_HIP_SPLIT_GRID=1; fi
  for i in 1 2; do timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('split=$v  %.4g q/s %.4f ms' % (d['value'], d['ms_per_step']), d['roofline']['avg_launch_ms'])"; done
done
unset PCU_HIP_SPLIT_GRID
bash scratch/gpu_trace.sh t6 | head -30
