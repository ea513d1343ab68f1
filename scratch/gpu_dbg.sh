#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
for lib in libpcu_hip.so libpcu_hip_bk4.so libpcu_hip_bk2.so libpcu_hip_bk16.so; do
  export PCU_HIP_LIBRARY=$GRAFT_REPO_ROOT/point_cloud_utils_amd/$lib
  for cfg in "" "" "--config c4"; do timeout 300 python bench.py $cfg --steps 100 --warmup 5 --no-cpu-baseline --no-parity 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    if l.startswith('{'): d=json.loads(l); print('$lib $cfg %.4g %s %.4f ms' % (d['value'], d['unit'], d['ms_per_step']), d.get('device_ms_per_step',{}).get('index_build'))"; done
done
