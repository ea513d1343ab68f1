#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 900 python -u -m pytest tests/test_gpu_configs.py tests/test_gpu_parity.py -x -q -m gpu -k "headline or fused or config4 or config2 or dense or unbalanced or golden or metrics or ties or sweep or dataset_index" 2>&1 | tail -5
timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/c4_bench.json 2>/dev/null
python - <<'PY'
import json
for l in open('gpurun_out/c4_bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print('%.4g q/s  %.4f ms/step' % (d['value'], d['ms_per_step']), d['roofline']['avg_launch_ms'], d['device_ms_per_step'])
PY
