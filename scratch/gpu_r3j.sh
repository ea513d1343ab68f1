#!/bin/bash
# round 3, call J: leaner layout + inner certification bound (headline), early-exit insertion chains (c3), parity
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for rep in 1 2; do
timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3j_bench_$rep.json 2> $OUT/r3j_bench_$rep.err
python - <<PY
import json
try:
    d=json.load(open("$OUT/r3j_bench_$rep.json"))
    print("headline ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal")))
except Exception as e:
    print("FAILED", e); print(open("$OUT/r3j_bench_$rep.err").read()[-1500:])
PY
done
for v in "" _kb20; do
  export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip$v.so
  for c in c3 c2; do timeout 300 python bench.py --config $c --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3j_$c$v.json; python -c "
import json; d=json.load(open('$OUT/r3j_$c$v.json')); print('lib$v $c', 'ms_per_step %.4f' % d['ms_per_step'], {k:v for k,v in d['parity'].items() if k!='stats'})"; done
done
unset PCU_HIP_LIBRARY
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3j_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3j_trace.log 2>&1)
python profiles/summarize_rocprof.py $(find $OUT/r3j_trace -name "*results.db" | head -1) 2>/dev/null | head -10
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3j_trace_c3 -- python $ROOT/bench.py --config c3 --steps 4 --warmup 2 --no-parity > $OUT/r3j_trace_c3.log 2>&1)
python profiles/summarize_rocprof.py $(find $OUT/r3j_trace_c3 -name "*results.db" | head -1) 2>/dev/null | head -12
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py tests/test_gpu_normals.py -m gpu -q -x -k "not switch" 2>&1 | tail -4
