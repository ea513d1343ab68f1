#!/bin/bash
cd $GRAFT_REPO_ROOT; export PYTHONUNBUFFERED=1
python -c "import torch"
timeout 300 python -u -m pytest tests/test_gpu_sinkhorn.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --config sinkhorn --steps 10 --warmup 2 2>&1 | grep -E "^\{" | cut -c1-700
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/sk_trace -- python $GRAFT_REPO_ROOT/bench.py --config sinkhorn --steps 3 --warmup 1 --no-parity > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
python - <<'PY'
import sqlite3, glob
db = sorted(glob.glob('gpurun_out/sk_trace/*/*.db'))[-1]
cur = sqlite3.connect(db).cursor()
for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
    print(f"{name[:90]:90s} {calls:6d} {tot:12.1f} {avg:10.2f} {pct:6.2f}")
PY
