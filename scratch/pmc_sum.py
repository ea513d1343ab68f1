"""Per-kernel means of rocprofv3 --pmc csv output: pmc_sum.py <dir> [<dir> ...]  (FETCH_SIZE / WRITE_SIZE are printed in MB as reported;
on gfx950 FETCH_SIZE counts half the bytes of wide reads -- profiles/r03_calib.txt)."""
import csv, glob, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for d in sys.argv[1:]:
    for f in glob.glob(f"{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].replace("void pcu::", "").split("(")[0]
            acc[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc):
    parts = []
    for c in sorted(acc[k]):
        v = sum(acc[k][c]) / len(acc[k][c])
        parts.append(f"{c}={v / 1024:.2f} MB" if c.endswith("_SIZE") else f"{c}={v:.4g}")
    print(f"{k[:60]:60s} n={len(next(iter(acc[k].values()))):3d}  " + "  ".join(parts))
