#!/bin/bash
# round 3, call A: full -m gpu suite on the pruned build + new parity tests, baseline bench, VALU issue-rate micro-benchmark with counters
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -25 > $OUT/r3a_pytest.log
tail -5 $OUT/r3a_pytest.log
timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline > $OUT/r3a_bench.json 2> $OUT/r3a_bench.err
cut -c1-400 $OUT/r3a_bench.json
cd /tmp && hipcc --offload-arch=gfx950 -O3 -o valu_rate $ROOT/profiles/ubench/valu_rate.hip 2>/dev/null && ./valu_rate > $OUT/r3a_valu_rate.txt 2>&1
rocprofv3 --kernel-trace --pmc SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_WAVES GRBM_GUI_ACTIVE --output-format csv -d $OUT/r3a_valu_pmc -- ./valu_rate > $OUT/r3a_valu_pmc.log 2>&1
cat $OUT/r3a_valu_rate.txt
