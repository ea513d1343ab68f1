#!/bin/bash
# ad-hoc counter passes for the search kernel (scratch): bash scratch/gpu_pmc.sh <tag> [env assignments...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-p}; shift
for e in "$@"; do export "$e"; done
python -c "import torch"
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-configs"
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- $B > $OUT/${TAG}_pmc_$name.log 2>&1; }
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS GRBM_GUI_ACTIVE
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
pass lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS SQ_LDS_ADDR_CONFLICT
cd $ROOT
