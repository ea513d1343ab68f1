#!/usr/bin/env python3
"""Turns gpurun_out/<tag>_* (written by profiles/collect.sh on the GPU box) into tracked summaries:
   profiles/<tag>_bench.json, profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt, profiles/hbm_traffic.json"""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r01"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
line = [l for l in open(os.path.join(G, f"{tag}_bench.json")) if l.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench.json"), "w").write(line)
db = glob.glob(os.path.join(G, f"{tag}_trace", "*", "*.db"))[0]
txt = subprocess.run([sys.executable, os.path.join(P, "summarize_rocprof.py"), db], capture_output=True, text=True).stdout
open(os.path.join(P, f"{tag}_kernel_stats.txt"), "w").write(txt.replace(ROOT + "/", ""))
def pmc(d):
    f = glob.glob(os.path.join(G, f"{tag}_pmc_{d}", "*", "*_counter_collection.csv"))[0]
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].replace("void pcu::", "").split("(")[0].replace("pcu::", "")
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg
out = ["# rocprofv3 --pmc passes (separate runs of `bench.py --steps 5 --warmup 2`), mean per dispatch",
       "# FETCH_SIZE / WRITE_SIZE are in KB. gfx950 note (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts 64 B per 128 B request,",
       "# i.e. reports 1/2 of the bytes of coalesced reads (k_bbox_partial reads 12.0 MB: ~5,870 KB reported); WRITE_SIZE",
       "# matched a known coalesced 16.0 MB write. hbm_bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024."]
fetch, write, sq = pmc("fetch"), pmc("write"), pmc("sq")
for extra_pass in ("ta", "tcp"):          # texture-addresser / L1 counters of the search kernels (optional passes)
    try:
        for k, cs in pmc(extra_pass).items():
            for c, v in cs.items(): sq[k][c] = v
    except Exception:
        pass
traffic = {}
for k in fetch:
    f = sum(fetch[k]["FETCH_SIZE"]) / len(fetch[k]["FETCH_SIZE"]); w = sum(write[k]["WRITE_SIZE"]) / len(write[k]["WRITE_SIZE"]) if k in write else 0
    traffic[k] = (2 * f + w) * 1024
    extra = "  ".join(f"{c}={sum(v)/len(v):.4g}" for c, v in sq.get(k, {}).items())
    out.append(f"{k:34s} n={len(fetch[k]['FETCH_SIZE']):3d} FETCH_SIZE={f:10.1f} WRITE_SIZE={w:10.1f} hbm_bytes={traffic[k]:.4g}  {extra}")
open(os.path.join(P, f"{tag}_pmc.txt"), "w").write("\n".join(out) + "\n")
flat = [v for k, v in traffic.items() if k.startswith("k_search1_flat<float")]
json.dump({"source": f"profiles/{tag}_pmc.txt", "k_search1_flat_f32_bytes_per_launch": flat[0] if flat else None,
           "note": "one launch = both directions of the 1M-vs-1M Chamfer step; 2*FETCH_SIZE + WRITE_SIZE, see the header of the source file"},
          open(os.path.join(P, "hbm_traffic.json"), "w"))
print(line[:600]); print("\n".join(out[4:12]))
