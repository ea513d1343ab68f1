#!/bin/bash
# round 3, call M: fill radius in the k > 1 lane pass (config 3, normals), parity
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
for v in 0 1; do
  if [ $v = 1 ]; then export PCU_HIP_NO_FILL_RADIUS=1; fi
  timeout 300 python bench.py --config c3 --steps 10 --warmup 2 2>/dev/null | grep '^{' > $OUT/r3m_c3_$v.json; python -c "
import json; d=json.load(open('$OUT/r3m_c3_$v.json')); r=d['roofline']; print('nofill=$v c3', 'ms_per_step %.4f' % d['ms_per_step'], 'kernel_ms', r.get('kernel_ms_live_hip_events'), {k:v for k,v in d['parity'].items() if k!='stats'}, d['parity']['stats']['n_escalated'])"
  timeout 300 python bench.py --config normals --steps 10 --warmup 2 2>/dev/null | grep '^{' | cut -c1-280
done
unset PCU_HIP_NO_FILL_RADIUS
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_normals.py tests/test_gpu_configs.py -m gpu -q -x -k "not switch" 2>&1 | tail -3
(cd /tmp && rocprofv3 --kernel-trace --stats -d $OUT/r3m_trace_c3 -- python $ROOT/bench.py --config c3 --steps 4 --warmup 2 --no-parity > $OUT/r3m_trace_c3.log 2>&1)
python scratch/timeline.py $(find $OUT/r3m_trace_c3 -name "*results.db" | head -1) > $OUT/r3m_c3_timeline.txt; tail -14 $OUT/r3m_c3_timeline.txt; rm -rf $OUT/r3m_trace_c3
