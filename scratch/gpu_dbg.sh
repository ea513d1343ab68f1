#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
run() { timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$1  %.4g q/s %.4f ms' % (d['value'], d['ms_per_step']), 'flat %.4f' % d['roofline']['avg_launch_ms'])"; }
unset PCU_HIP_HYB; run "default      "
export PCU_HIP_HYB=1
run "hyb 3/24     "
for lib in h2_24 h4_24 h3_32 h3_16; do PCU_HIP_LIBRARY=$GRAFT_REPO_ROOT/point_cloud_utils_amd/libpcu_hip_$lib.so run "hyb $lib   "; done
timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('parity', d.get('parity'))"
timeout 150 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or lattice or tie or sweep or hausdorff or chamfer" 2>&1 | tail -2
