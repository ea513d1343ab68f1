#!/usr/bin/env python3
"""Compact timeline of the last call in a rocprofv3 kernel-trace db: scratch/timeline.py results.db [first-kernel-substring]"""
import sqlite3, sys, collections
db = sys.argv[1]; first = sys.argv[2] if len(sys.argv) > 2 else "k_bbox_partial"
cur = sqlite3.connect(db).cursor()
rows = list(cur.execute("select name,start,end,queue_id from kernels order by start"))
idx = [i for i, r in enumerate(rows) if first in r[0]]
# a call may contain several bbox launches (restarts, coarse grids): split calls by large host gaps instead
calls = []; cur_call = [rows[0]]
for r in rows[1:]:
    if r[1] - max(x[2] for x in cur_call) > 150_000: calls.append(cur_call); cur_call = [r]
    else: cur_call.append(r)
calls.append(cur_call)
print("calls:", len(calls), "spans(us):", [round((max(x[2] for x in c) - c[0][1]) / 1e3) for c in calls])
c = calls[-1]; t0 = c[0][1]; prev_end = t0
agg = collections.OrderedDict()
for r in c:
    nm = r[0].split('(')[0].replace('void pcu::', '')[:44]
    gap = (r[1] - prev_end) / 1e3
    print(f"{(r[1]-t0)/1e3:9.1f} {(r[2]-r[1])/1e3:8.1f} q{r[3]} {nm}" + (f"   <-- gap {gap:.1f}" if gap > 8 else ""))
    prev_end = max(prev_end, r[2])
    agg[nm] = agg.get(nm, 0) + (r[2] - r[1]) / 1e3
print("total span", (prev_end - t0) / 1e3, "sum of kernels", sum(agg.values()))
for k, v in sorted(agg.items(), key=lambda kv: -kv[1])[:12]: print(f"   {v:9.1f}  {k}")
