#!/bin/bash
# A/B of the bucketed index build and the k=1 group-wise search kernel (run on the GPU box through gpurun)
cd ${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p gpurun_out
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > gpurun_out/ab_pytest.txt
for v in default; do
  case $v in
    default) E="" ;;
    split_sort) E="PCU_HIP_SPLIT_SORT=1" ;;
  esac
  echo "== $v" >> gpurun_out/ab_bench.txt
  env $E timeout 300 python bench.py --steps 40 --warmup 5 --no-cpu-baseline >> gpurun_out/ab_bench.txt 2>&1
done
timeout 600 python scratch/skew.py > gpurun_out/ab_skew.txt 2>&1

cd /tmp && export TMPDIR=/tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/ab_trace -- python $GRAFT_REPO_ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/ab_trace.log 2>&1
cd $GRAFT_REPO_ROOT
cat gpurun_out/ab_pytest.txt; python - <<'PY'
import json
for l in open('gpurun_out/ab_bench.txt'):
    l=l.strip()
    if l.startswith('=='): print(l)
    elif l.startswith('{'):
        d=json.loads(l); print(round(d['ms_per_step'],4), d['device_ms_per_step'], round(d['roofline']['avg_launch_ms'],4), d['chamfer'])
    else: print(l[:300])
PY
cat gpurun_out/ab_skew.txt
python profiles/summarize_rocprof.py gpurun_out/ab_trace/runc/*_results.db 2>&1 | cut -c1-140 | head -50
