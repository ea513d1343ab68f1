#!/bin/bash
# round 3, call B: A/B of the k = 1 lane-pass variants (PCU_FLAT_XYZ x PCU_FLAT_SORT) -- bench line (parity inside), kernel trace
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in 00 10 01 11; do
  export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip_v$v.so
  for rep in 1 2; do
    timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3b_bench_v${v}_$rep.json 2> $OUT/r3b_bench_v${v}_$rep.err
    python - <<PY
import json
try:
    d=json.load(open("$OUT/r3b_bench_v${v}_$rep.json"))
    print("v$v rep$rep ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal")))
except Exception as e:
    print("v$v rep$rep FAILED", e); print(open("$OUT/r3b_bench_v${v}_$rep.err").read()[-1500:])
PY
  done
done
for v in 00 11; do
  export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip_v$v.so
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3b_trace_v$v -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3b_trace_v$v.log 2>&1)
  python profiles/summarize_rocprof.py $(find $OUT/r3b_trace_v$v -name "*results.db" | head -1) 2>/dev/null | head -14
done
export PCU_HIP_LIBRARY=$ROOT/point_cloud_utils_amd/libpcu_hip_v11.so
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "not switch" 2>&1 | tail -6
