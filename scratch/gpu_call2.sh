#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
timeout 300 python -m pytest tests/test_gpu_configs.py -x -q -m gpu -k "headline or fused or config4" > gpurun_out/c2_tests.log 2>&1; echo "rc=$?" >> gpurun_out/c2_tests.log
for i in 1 2; do
  timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/c2_bench_A$i.json 2>/dev/null
  PCU_HIP_LIBRARY=$GRAFT_REPO_ROOT/point_cloud_utils_amd/libpcu_hip_w8.so timeout 200 python bench.py --steps 100 --warmup 5 --no-cpu-baseline --no-parity > gpurun_out/c2_bench_B$i.json 2>/dev/null
done
PCU_HIP_PROF_BUILD=1 timeout 100 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-parity > /dev/null 2> gpurun_out/c2_prof_build.log
timeout 200 python bench.py --config c4 --steps 10 --warmup 2 > gpurun_out/c2_c4.json 2>/dev/null
tail -3 gpurun_out/c2_tests.log
for f in gpurun_out/c2_bench_*.json gpurun_out/c2_c4.json; do python - $f <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); print(sys.argv[1], '%.4g q/s  %.4f ms/step' % (d['value'], d['ms_per_step']), d.get('roofline',{}).get('avg_launch_ms'), d.get('device_ms_per_step'))
PY
done
tail -4 gpurun_out/c2_prof_build.log
