import csv, glob, sys, collections
tag = sys.argv[1]; pat = sys.argv[2] if len(sys.argv) > 2 else "k_search1_flat"
acc = collections.defaultdict(list)
for f in glob.glob(f"gpurun_out/{tag}_pmc_*/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if pat in r["Kernel_Name"]: acc[r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(acc): print(f"{k:28s} n={len(acc[k]):3d} mean={sum(acc[k])/len(acc[k]):.4g}")
