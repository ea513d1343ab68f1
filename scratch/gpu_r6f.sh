#!/bin/bash
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
( PCU_HIP_DEBUG_SKEW=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k layout_handed 2>&1 | grep -v "^\[pair\|^\[finish\|\[rescale" | tail -40 ) > gpurun_out/r6f_test.log 2>&1
tail -40 gpurun_out/r6f_test.log
PCU_HIP_PROF_BUILD2=1 python scratch/build_prof.py 2>&1 | grep prof | sed -n '3,4p;11,12p'
bash scratch/ab.sh 3 geo= nogeo=PCU_HIP_NO_GEO_CACHE=1
