#!/bin/bash
# round 5, call C: k_search_runs (k > 1 lane pass on the run list): parity tests that run k > 1, config 3 A/B against k_search
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time python -m pytest tests/test_gpu_sweep.py tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -x -q -k "sweep or offset or seeded or golden_knn or config3 or shifted or near_ties or max_points or dataset_index or nonfinite_inputs or KSEARCH or kd_tree or k_beyond" 2>&1 | tail -12 ) > gpurun_out/c_tests.log 2>&1
STEPS=8 bash scratch/gpu_cfg_ab.sh c3 2 runs= v1=PCU_HIP_KSEARCH_V1=1 > gpurun_out/c_c3_ab.log 2>&1
python bench.py --config c3 --steps 8 --warmup 2 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], {k: v for k, v in d['parity'].items() if k != 'stats'}, d['parity'].get('stats'), d['roofline']['dominant_kernel'].get('ms_live_hip_events'))" > gpurun_out/c_c3.txt 2>&1
python bench.py --config normals --steps 8 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300 > gpurun_out/c_normals.txt
cat gpurun_out/c_tests.log gpurun_out/c_c3_ab.log gpurun_out/c_c3.txt gpurun_out/c_normals.txt
