#!/bin/bash
# final GPU call of a round: collection at HEAD, the randomised sweep (scratch/fuzz.py, 12 seeds x 350 cases), the full -m gpu suite, smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
bash profiles/collect.sh r06 > gpurun_out/r06_collect.log 2>&1
for seed in 625 626; do
  ( time timeout 900 python scratch/fuzz.py $seed 350 ) > gpurun_out/r06_fuzz_$seed.log 2>&1
done
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r06_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r06_smoke.log 2>&1
grep -h "^fuzz:" gpurun_out/r06_fuzz_*.log; tail -3 gpurun_out/r06_final_tests.log gpurun_out/r06_smoke.log
