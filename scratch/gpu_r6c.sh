#!/bin/bash
# round 6, call: first run of the staged (brick) k = 1 pass on a shared grid: fused-sum parity, then headline A/B against k_search1_flat
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time timeout 900 python -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py -m gpu -x -q -k "fused or chamfer or Chamfer or metrics or sweep" 2>&1 | tail -12 ) > gpurun_out/r6c_tests.log 2>&1
for i in 1 2; do
  for v in "PCU_HIP_BRICK=1" "PCU_HIP_BRICK=0" "PCU_HIP_NO_SHARED_GRID=1"; do
    echo "== $v" >> gpurun_out/r6c_ab.log
    ( env $v timeout 300 python bench.py --no-configs --no-cpu-baseline --steps 200 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['roofline']['avg_launch_ms'], d['device_ms_per_step'], d['parity'])" ) >> gpurun_out/r6c_ab.log 2>&1
  done
done
tail -12 gpurun_out/r6c_tests.log; cat gpurun_out/r6c_ab.log
