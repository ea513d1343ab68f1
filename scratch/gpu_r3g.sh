#!/bin/bash
# round 3, call G: block-level exact accumulation in the wave pass, parallel k_bucket_large; wave grid A/B; traces
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in "" _wb2048; do
  export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip$v.so
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3g_bench$v.json 2> $OUT/r3g_bench$v.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/r3g_bench$v.json"))
    print("lib$v headline ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal")))
except Exception as e:
    print("lib$v FAILED", e); print(open("$OUT/r3g_bench$v.err").read()[-1500:])
PY
  for c in gauss cluster; do timeout 300 python bench.py --config $c --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3g_$c$v.json; python -c "
import json; d=json.load(open('$OUT/r3g_$c$v.json')); print('lib$v $c', 'ms_per_step %.4f' % d['ms_per_step'], {k:v for k,v in d['parity'].items() if k!='stats'})"; done
done
unset PCU_HIP_LIBRARY
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3g_trace_gauss -- python $ROOT/bench.py --config gauss --steps 4 --warmup 2 --no-parity > $OUT/r3g_trace_gauss.log 2>&1)
python scratch/timeline.py $(find $OUT/r3g_trace_gauss -name "*results.db" | head -1) | tail -24
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3g_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3g_trace.log 2>&1)
python profiles/summarize_rocprof.py $(find $OUT/r3g_trace -name "*results.db" | head -1) 2>/dev/null | head -10
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "not switch" 2>&1 | tail -4
