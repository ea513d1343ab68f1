#!/bin/bash
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}; cd $ROOT; export TMPDIR=/tmp
for v in scratch/variants/*.so; do
  cp $v point_cloud_utils_amd/libpcu_hip.so
  echo "== $v"
  timeout 120 python - <<'PY' 2>&1 | tail -3
import numpy as np, torch, point_cloud_utils_amd as pcu
x=torch.rand(100000,3).cuda(); y=torch.rand(100000,3).cuda()
print("chamfer", float(pcu.chamfer_distance(x,y)))
d,i=pcu.k_nearest_neighbors(x,y,16); print("knn16", float(d.sum()))
PY
done
