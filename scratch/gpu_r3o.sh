#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; OUT=$ROOT/gpurun_out; mkdir -p $OUT; cd $ROOT
export TMPDIR=/tmp
PCU_HIP_PROF_BUILD=1 timeout 120 python - <<'PY' 2>&1 | grep -E "prof" | tail -6
import numpy as np, torch, point_cloud_utils_amd as pcu
x = torch.rand((1000000, 3), device="cuda"); y = torch.rand((1000000, 3), device="cuda")
for _ in range(6): pcu.chamfer_distance(x, y)
PY
