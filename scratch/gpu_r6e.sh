#!/bin/bash
# round 6: the layout handed down between calls (GridGeo): parity, staleness, A/B
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_sweep.py -m gpu -x -q 2>&1 | tail -12 ) > gpurun_out/r6e_tests.log 2>&1
tail -12 gpurun_out/r6e_tests.log
bash scratch/ab.sh 3 geo= nogeo=PCU_HIP_NO_GEO_CACHE=1 | tee gpurun_out/r6e_ab.txt
PCU_HIP_PROF_BUILD2=1 python scratch/build_prof.py 2>&1 | grep prof | sed -n '3,4p;11,12p' | tee -a gpurun_out/r6e_ab.txt
( timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4', d['ms_per_step'], d['parity'])" ) | tee -a gpurun_out/r6e_ab.txt
( PCU_HIP_NO_GEO_CACHE=1 timeout 300 python bench.py --config c4 --steps 10 --warmup 3 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('c4 nogeo', d['ms_per_step'], d['parity'])" ) | tee -a gpurun_out/r6e_ab.txt
