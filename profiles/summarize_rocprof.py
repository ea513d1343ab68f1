#!/usr/bin/env python3
"""Summarise a rocprofv3 (--kernel-trace --stats) rocpd sqlite database: per-kernel calls / total / average /
percentage, plus the kernel timeline of the last complete bench step. Usage: summarize_rocprof.py results.db"""
import sqlite3
import sys


def main(path):
    cur = sqlite3.connect(path).cursor()
    print(f"# source: {path}")
    print(f"{'kernel':100s} {'calls':>6s} {'total_us':>12s} {'avg_us':>10s} {'pct':>6s}")
    for name, calls, tot, avg, pct in cur.execute("select name,total_calls,total_duration,average,percentage from top_kernels"):
        print(f"{name[:100]:100s} {calls:6d} {tot/1:12.1f} {avg:10.2f} {pct:6.2f}")
    rows = list(cur.execute("select name, start, end, grid_x from kernels order by start"))
    marks = [r for r in rows if "k_bbox" in r[0] or "k_index_prep" in r[0]]
    if len(marks) >= 4:
        s = marks[-4][1] if len(marks) >= 4 else marks[0][1]
        # one step = two index builds: start at the 2nd-to-last pair
        s = marks[-2][1]
        prev = [r for r in rows if r[1] < s]
        # walk back to the first kernel of the step (the build before)
        s = marks[-2 - 0][1]
        print("\n# timeline of the last step (us since its first index-build kernel; profiled clocks)")
        first = None
        for r in rows:
            if r[1] >= marks[-2][1] - 1 and first is None:
                first = r[1]
        start_idx = rows.index(marks[-2])
        # include the preceding init kernel(s) of that build
        while start_idx > 0 and rows[start_idx][1] - rows[start_idx - 1][2] < 20000 and "k_search" not in rows[start_idx - 1][0] \
                and "pnorm" not in rows[start_idx - 1][0] and "sum_final" not in rows[start_idx - 1][0] and "copyBuffer" not in rows[start_idx - 1][0]:
            start_idx -= 1
        t0 = rows[start_idx][1]
        for r in rows[start_idx:]:
            print(f"{(r[1]-t0)/1000:9.1f}  dur {(r[2]-r[1])/1000:8.1f}  grid {r[3]:>8}  {r[0][:90]}")


if __name__ == "__main__":
    main(sys.argv[1])
