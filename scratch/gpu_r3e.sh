#!/bin/bash
# round 3, call E: robust grid range + in-kernel escalation: full gpu suite, headline, uneven configs, traces
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
timeout 1500 python -m pytest tests -m gpu -q -x 2>&1 | tail -15 > $OUT/r3e_pytest.log; tail -6 $OUT/r3e_pytest.log
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3e_bench.json 2> $OUT/r3e_bench.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/r3e_bench.json"))
    print("headline ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal")))
except Exception as e:
    print("FAILED", e); print(open("$OUT/r3e_bench.err").read()[-1500:])
PY
for c in gauss cluster outlier c5 c2; do timeout 300 python bench.py --config $c --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3e_$c.json; python -c "
import json; d=json.load(open('$OUT/r3e_$c.json')); print('$c', 'ms_per_step %.4f' % d['ms_per_step'], {k:v for k,v in d['parity'].items() if k!='stats'}, d['parity'].get('stats'))"; done
for c in gauss outlier; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3e_trace_$c -- python $ROOT/bench.py --config $c --steps 4 --warmup 2 --no-parity > $OUT/r3e_trace_$c.log 2>&1)
done
