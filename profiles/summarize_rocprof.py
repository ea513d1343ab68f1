#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd sqlite database: per-kernel calls / total / average /
percentage, plus the kernel timeline of two consecutive bench steps from the middle of the timed region (steps without HIP
events: bench.py brackets the search launch on every 8th step only, and its last three steps carry phase events).
Usage: summarize_rocprof.py results.db [--json out.json]   (--json: per-kernel calls / total_us / avg_us, for profiles/postprocess.py;
the databases themselves are ~14 MB each and stay on the GPU box)"""
import sqlite3
import sys


def main(path, json_out=None):
    cur = sqlite3.connect(path).cursor()
    if json_out:
        import json
        json.dump([{"name": n, "calls": c, "total_us": t, "avg_us": a} for n, c, t, a in
                   cur.execute("select name,total_calls,total_duration,average from top_kernels")], open(json_out, "w"))
    print(f"# source: {path}")
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:100]:100s} {calls:6d} {tot/1:12.1f} {avg:10.2f} {pct:6.2f}")
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    starts = [i for i, r in enumerate(rows) if "k_bbox_partial" in r[0] or "k_bucket_onepass3" in r[0]]          # first kernel of every step
    if len(starts) >= 8:
        a = starts[len(starts) // 2 + 1]
        b = starts[len(starts) // 2 + 3]
        t0 = rows[a][1]
        print("\n# timeline of two consecutive steps from the middle of the run (us since the first kernel; profiled clocks)")
        prev_end = None
        for r in rows[a:b]:
            gap = "" if prev_end is None else f"  gap {(r[1] - prev_end) / 1000:6.1f}"
            print(f"{(r[1]-t0)/1000:9.1f}  dur {(r[2]-r[1])/1000:8.1f}  grid {r[3]:>8}  {r[0][:90]}{gap}")
            prev_end = r[2]


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[3] if len(sys.argv) > 3 and sys.argv[2] == "--json" else None)
