#!/bin/bash
# Alternating A/B of bench.py between the built library and variant libraries: ab.sh <rounds> <name=ENV1=V1,ENV2=V2 ...>   ("base=" = no env)
cd $GRAFT_REPO_ROOT
R=$1; shift
P='import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print("%.4f %.4f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"]))'
for r in $(seq $R); do
  for spec in "$@"; do
    name=${spec%%=*}; envs=${spec#*=}
    out=$(env $(echo $envs | tr ',' ' ') python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-parity --no-configs 2>/dev/null | python -c "$P")
    echo "$name $out"
  done
done | sort | awk '{n[$1]++; s[$1]+=$2; k[$1]+=$3; if(!($1 in mn)||$2<mn[$1])mn[$1]=$2} END{for(v in n) printf "%-12s runs %d  ms_per_step mean %.4f min %.4f  flat kernel mean %.4f\n", v, n[v], s[v]/n[v], mn[v], k[v]/n[v]}'
