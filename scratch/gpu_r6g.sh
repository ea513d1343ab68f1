#!/bin/bash
# round 6: scatter block size by cloud size (PTS): A/B at the headline and config 4, parity subset
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -4 ) | tee gpurun_out/r6g_tests.log
bash scratch/ab.sh 3 auto= pts8=PCU_HIP_BUILD_PTS=8 pts4=PCU_HIP_BUILD_PTS=4 pts2=PCU_HIP_BUILD_PTS=2 | tee gpurun_out/r6g_ab.txt
for v in "X=1" "PCU_HIP_BUILD_PTS=8" "PCU_HIP_BUILD_PTS=4" "PCU_HIP_BUILD_PTS=2"; do
  for i in 1 2; do ( env $v timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 $v', d['ms_per_step'], d['parity'])" ) | tee -a gpurun_out/r6g_ab.txt; done
done
PCU_HIP_PROF_BUILD2=1 python scratch/build_prof.py 2>&1 | grep prof | sed -n '3,4p;11,12p' | tee -a gpurun_out/r6g_ab.txt
