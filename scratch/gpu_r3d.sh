#!/bin/bash
# round 3, call D: fast grid layout in the one-pass kernel; traces of the uneven-cloud configs; counters of the k = 16 lane kernel
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for rep in 1 2; do
  timeout 300 python bench.py --steps 60 --warmup 5 --no-cpu-baseline > $OUT/r3d_bench_$rep.json 2> $OUT/r3d_bench_$rep.err
  python - <<PY
import json
try:
    d=json.load(open("$OUT/r3d_bench_$rep.json"))
    print("rep$rep ms_per_step %.4f search_kernel_ms %.4f idx_ms %.4f parity %s e2e %.3f" % (d["ms_per_step"], d["roofline"]["avg_launch_ms"], d["device_ms_per_step"]["index_build"], d.get("parity",{}).get("idx_equal"), d["end_to_end_numpy"]["ms_per_call"]))
except Exception as e:
    print("rep$rep FAILED", e); print(open("$OUT/r3d_bench_$rep.err").read()[-1500:])
PY
done
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3d_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity > $OUT/r3d_trace.log 2>&1)
python profiles/summarize_rocprof.py $(find $OUT/r3d_trace -name "*results.db" | head -1) 2>/dev/null | head -10
for c in gauss cluster outlier; do
  (cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3d_trace_$c -- python $ROOT/bench.py --config $c --steps 4 --warmup 2 --no-parity > $OUT/r3d_trace_$c.log 2>&1)
done
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY --output-format csv -d $OUT/r3d_pmc_c3 -- python $ROOT/bench.py --config c3 --steps 2 --warmup 1 --no-parity > $OUT/r3d_pmc_c3.log 2>&1)
(cd /tmp && rocprofv3 --kernel-trace --pmc SQ_WAVES SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY TA_TA_BUSY_sum --output-format csv -d $OUT/r3d_pmc2_c3 -- python $ROOT/bench.py --config c3 --steps 2 --warmup 1 --no-parity > $OUT/r3d_pmc2_c3.log 2>&1)
ls $OUT | grep r3d
