#!/bin/bash
# round 5, call B: the full -m gpu suite at the new small-cloud thresholds (+ durations), c1 line with parity
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( time python -m pytest tests -m gpu -x -q --durations=12 2>&1 | tail -30 ) > gpurun_out/b_tests.log 2>&1
python bench.py --config c1 --steps 20 --warmup 3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['ms_per_step'], d['parity'], d.get('cpu_baseline',{}).get('value'))" > gpurun_out/b_c1.txt 2>&1
cat gpurun_out/b_tests.log gpurun_out/b_c1.txt
