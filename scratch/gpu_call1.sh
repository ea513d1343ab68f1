#!/bin/bash
# GPU call 1 of round 2: new full-size config tests, a slice of the parity suite over the changed paths, bench, profiles.
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
export PYTHONUNBUFFERED=1
timeout 900 python -m pytest tests/test_gpu_configs.py -x -q -m gpu > gpurun_out/c1_configs.log 2>&1; echo "configs rc=$?" >> gpurun_out/c1_configs.log
timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "golden or metrics or ties or unbalanced or batched or bodies or sweep or shifted or timing or dense" > gpurun_out/c1_parity.log 2>&1; echo "parity rc=$?" >> gpurun_out/c1_parity.log
timeout 900 bash profiles/collect.sh r02a > gpurun_out/c1_collect.log 2>&1
for c in c2 c3 c4 c5; do timeout 300 python bench.py --config $c --steps 10 --warmup 2 >> gpurun_out/c1_configs_bench.log 2>&1; done
tail -5 gpurun_out/c1_configs.log gpurun_out/c1_parity.log; tail -c 1500 gpurun_out/c1_collect.log; cat gpurun_out/c1_configs_bench.log | cut -c1-400
