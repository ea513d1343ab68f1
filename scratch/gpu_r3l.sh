#!/bin/bash
# round 3, call L: k > 1 lane kernel: pipelined row loop, occupancy variants (config 3)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in "" _p0 _w5 _w6; do
  export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip$v.so
  timeout 300 python bench.py --config c3 --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3l_c3$v.json; python -c "
import json; d=json.load(open('$OUT/r3l_c3$v.json')); r=d['roofline']; print('lib$v c3', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_live_hip_events'), {k:v for k,v in d['parity'].items() if k!='stats'})"
done
unset PCU_HIP_LIBRARY
timeout 600 python -m pytest tests/test_gpu_parity.py tests/test_gpu_normals.py -m gpu -q -x -k "seeded or golden or normals or sweep" 2>&1 | tail -3
timeout 300 python bench.py --config normals --steps 10 --warmup 2 2>/dev/null | grep '^{' | cut -c1-300
