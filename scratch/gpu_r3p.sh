#!/bin/bash
# round 3, call P: tail fold inside the wave kernel's last block
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export PCU_HIP_TAIL_KERNEL=1; fi
  for rep in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3p_bench_${v}_$rep.json 2> $OUT/r3p_bench_${v}_$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/r3p_bench_${v}_$rep.json"))
    print("tail_kernel=$v headline ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s chamfer %.10g" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal"), d["chamfer"]))
except Exception as e:
    print("FAILED", e); print(open("$OUT/r3p_bench_${v}_$rep.err").read()[-1500:])
PY
  done
  timeout 300 python bench.py --config c4 --steps 10 --warmup 2 2>/dev/null | grep '^{' | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('tail_kernel=$v c4 ms %.4f'%d['ms_per_step'], d['parity'])"
done
unset PCU_HIP_TAIL_KERNEL
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3p_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3p_trace.log 2>&1)
python profiles/summarize_rocprof.py $(find $OUT/r3p_trace -name "*results.db" | head -1) 2>/dev/null | head -10; rm -rf $OUT/r3p_trace
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_configs.py -m gpu -q -x -k "not switch or TAIL" 2>&1 | tail -4
