#!/bin/bash
# final GPU call of a round: collection at HEAD, the randomised sweep (scratch/fuzz.py), the full -m gpu suite, smoke
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
bash profiles/collect.sh r05 > gpurun_out/r05_collect.log 2>&1
( time timeout 900 python scratch/fuzz.py 505 350 ) > gpurun_out/r05_fuzz_505.log 2>&1
( time timeout 900 python scratch/fuzz.py 506 350 ) > gpurun_out/r05_fuzz_506.log 2>&1
( time python -m pytest tests -m gpu -x -q 2>&1 | tail -6 ) > gpurun_out/r05_final_tests.log 2>&1
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" > gpurun_out/r05_smoke.log 2>&1
tail -3 gpurun_out/r05_fuzz_505.log gpurun_out/r05_fuzz_506.log gpurun_out/r05_final_tests.log gpurun_out/r05_smoke.log
