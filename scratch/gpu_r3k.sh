#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd /tmp; export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity"
for v in a b; do
  if [ $v = b ]; then export PCU_HIP_GRID_KERNEL=1; fi
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/r3k_w_$v -- $B > $OUT/r3k_w_$v.log 2>&1
  rocprofv3 --kernel-trace --stats -d $OUT/r3k_t_$v -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3k_t_$v.log 2>&1
  python $ROOT/profiles/summarize_rocprof.py $(find $OUT/r3k_t_$v -name "*results.db" | head -1) | head -10; rm -rf $OUT/r3k_t_$v
  python - <<PY
import csv,glob,collections
f=glob.glob("$OUT/r3k_w_$v/*/*counter_collection.csv")[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    agg[r['Kernel_Name'].split('(')[0][:40]].append(float(r['Counter_Value']))
for k,v in agg.items(): print("$v", k, round(sum(v)/len(v),1))
PY
done
