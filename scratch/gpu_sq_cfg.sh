#!/bin/bash
# SQ counter passes of one bench.py --config under an environment variant: gpu_sq_cfg.sh <tag> <config> <kernel pattern> [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=$1; CFG=$2; PAT=$3; shift; shift; shift
for kv in "$@"; do export "$kv"; done
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --config $CFG --steps 3 --warmup 1 --no-cpu-baseline --no-parity --no-kernel-events"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- $B > $OUT/${TAG}_pmc_$name.log 2>&1; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES GRBM_GUI_ACTIVE
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
cd $ROOT
python scratch/pmc_sum.py $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2 | grep "$PAT" > $OUT/${TAG}_sq.txt
rm -rf $OUT/${TAG}_pmc_sq $OUT/${TAG}_pmc_sq2
cat $OUT/${TAG}_sq.txt
