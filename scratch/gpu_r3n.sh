#!/bin/bash
# round 3, call N: speculative kd top enqueued after the searches (config 3, normals)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export PCU_HIP_NO_KD_SPEC=1; fi
  timeout 300 python bench.py --config c3 --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3n_c3_$v.json; python -c "
import json; d=json.load(open('$OUT/r3n_c3_$v.json')); r=d['roofline']; print('nospec=$v c3', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_live_hip_events'), {k:v for k,v in d['parity'].items() if k!='stats'})"
  timeout 300 python bench.py --config normals --steps 10 --warmup 2 2>/dev/null | grep '^{' | cut -c100-250
done
unset PCU_HIP_NO_KD_SPEC
timeout 600 python -m pytest tests/test_gpu_configs.py -m gpu -q -x -k "speculative or config3 or beyond" 2>&1 | tail -3
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3n_trace_c3 -- python $ROOT/bench.py --config c3 --steps 4 --warmup 2 --no-parity > $OUT/r3n_trace_c3.log 2>&1)
python scratch/timeline.py $(find $OUT/r3n_trace_c3 -name "*results.db" | head -1) > $OUT/r3n_c3_timeline.txt; rm -rf $OUT/r3n_trace_c3
n=$(grep -n "k_bbox_partial" $OUT/r3n_c3_timeline.txt | tail -1 | cut -d: -f1); sed -n "${n},\$p" $OUT/r3n_c3_timeline.txt | awk '{ if ($2+0 > 60 || /k_search|k_bbox|kd_search|subtree|roi|result_block/) print }' | head -40
