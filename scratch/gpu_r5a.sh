#!/bin/bash
# round 5, call A: new sweep + cancel tests; config 1 under the small-cloud thresholds
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
nproc > gpurun_out/a_nproc.txt
( time python -m pytest tests/test_gpu_sweep.py tests/test_gpu_cancel.py -x -q --durations=8 2>&1 | tail -25 ) > gpurun_out/a_tests.log 2>&1
STEPS=30 bash scratch/gpu_cfg_ab.sh c1 2 base= wob=PCU_HIP_WAVE_ONLY_BELOW=2048 bmin=PCU_HIP_BUCKET_MIN=2048 both=PCU_HIP_WAVE_ONLY_BELOW=2048,PCU_HIP_BUCKET_MIN=2048 > gpurun_out/a_c1_ab.log 2>&1
PCU_HIP_WAVE_ONLY_BELOW=2048 PCU_HIP_BUCKET_MIN=2048 python bench.py --config c1 --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-900 > gpurun_out/a_c1_both_line.txt
PCU_HIP_WAVE_ONLY_BELOW=2048 PCU_HIP_BUCKET_MIN=2048 python bench.py --config c5 --steps 20 --warmup 3 2>&1 | tail -1 | cut -c1-900 > gpurun_out/a_c5_both_line.txt
python bench.py --config c5 --steps 20 --warmup 3 --no-parity --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 > gpurun_out/a_c5_base_line.txt
cat gpurun_out/a_nproc.txt gpurun_out/a_tests.log gpurun_out/a_c1_ab.log gpurun_out/a_c1_both_line.txt gpurun_out/a_c5_both_line.txt gpurun_out/a_c5_base_line.txt
