import sys, os, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import point_cloud_utils_amd as pcu
from conftest import cloud
n = 1000000
x, y = cloud(1000, n, np.float32), cloud(1001, n, np.float32)
dxy, c1 = pcu.k_nearest_neighbors(x, y, 1)
dyx, c2 = pcu.k_nearest_neighbors(y, x, 1)
h, i, j = pcu.hausdorff_distance(x, y, return_index=True)
print("hausdorff", h, i, j, "expected", float(max(dxy.max(), dyx.max())), pcu.last_stats())
if h != float(max(dxy.max(), dyx.max())):
    for nm, d, c, a, b in (("x->y", dxy, c1, x, y), ("y->x", dyx, c2, y, x)):
        for ii, jj in ((i, j), (j, i)):
            if ii < len(d):
                print(nm, "row", ii, "knn d", d[ii], "c", c[ii], "| claimed partner", jj, "dist", float(np.linalg.norm(a[ii] - b[jj])))
ch = pcu.chamfer_distance(x, y)
print("chamfer", float(ch), "from rows", float(np.float32(dxy.astype(np.float64).mean()) + np.float32(dyx.astype(np.float64).mean())), pcu.last_stats())
