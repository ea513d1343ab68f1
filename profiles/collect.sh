#!/bin/bash
# Collects the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_bench.json         bench.py line (with cpu_baseline)
#   gpurun_out/<tag>_trace/             rocprofv3 --kernel-trace --stats of the same command
#   gpurun_out/<tag>_pmc_fetch|write/   rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, csv)
# then `python profiles/postprocess.py <tag>` (CPU side) turns them into the tracked summaries under profiles/.
TAG=${1:-r01}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
python $ROOT/bench.py --steps 30 --warmup 3 > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $OUT/${TAG}_trace.log 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_pmc_fetch -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_pmc_write -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_write.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY GRBM_GUI_ACTIVE --output-format csv -d $OUT/${TAG}_pmc_sq -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_sq.log 2>&1
rocprofv3 --kernel-trace --pmc TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum --output-format csv -d $OUT/${TAG}_pmc_ta -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_ta.log 2>&1
rocprofv3 --kernel-trace --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $OUT/${TAG}_pmc_tcp -- python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline > $OUT/${TAG}_pmc_tcp.log 2>&1
tail -c 1500 $OUT/${TAG}_bench.json
