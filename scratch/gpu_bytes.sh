#!/bin/bash
# HBM-byte counter passes (FETCH_SIZE, WRITE_SIZE; separate runs) of bench.py, per-kernel means: gpu_bytes.sh <tag> [ENV=VALUE ...]
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out; export PYTHONUNBUFFERED=1
TAG=${1:-b}; shift
for kv in "$@"; do export "$kv"; done
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/gpurun_out
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-configs ${PCU_BENCH_ARGS:-}"
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/${TAG}_pmc_$c -- $B > $OUT/${TAG}_pmc_$c.log 2>&1
done
cd $ROOT
python scratch/pmc_sum.py $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE > $OUT/${TAG}_bytes.txt
rm -rf $OUT/${TAG}_pmc_FETCH_SIZE $OUT/${TAG}_pmc_WRITE_SIZE
cat $OUT/${TAG}_bytes.txt
