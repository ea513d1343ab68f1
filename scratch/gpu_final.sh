#!/bin/bash
# the round's closing run: full `-m gpu` suite, smoke(), one bench line per config (profiles/<tag>_configs.jsonl is refreshed from it)
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp; TAG=${1:-r04}
timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 2>&1 | tail -16
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
: > gpurun_out/${TAG}_configs.jsonl
for c in c1 c2 c3 c4 c5 gauss cluster outlier normals morton voxel sinkhorn; do
  timeout 400 python bench.py --config $c --steps 10 --warmup 2 2>/dev/null | grep '^{' >> gpurun_out/${TAG}_configs.jsonl
done
python bench.py --steps 50 --warmup 5 > gpurun_out/${TAG}_bench.json 2>/dev/null
cut -c1-200 gpurun_out/${TAG}_configs.jsonl; tail -c 600 gpurun_out/${TAG}_bench.json
