#!/bin/bash
# A/B of one bench.py config under environment variants, interleaved: gpu_cfg_ab.sh <config> <rounds> name=ENV=V,ENV=V name2= ...
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
CFG=$1; R=$2; shift; shift
for r in $(seq $R); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    ( IFS=,; for kv in $envs; do [ -n "$kv" ] && export "$kv"; done
      python bench.py --config $CFG --steps ${STEPS:-10} --warmup 3 --no-parity --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('$name', round(d['ms_per_step'],4))" )
  done
done
