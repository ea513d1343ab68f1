#!/usr/bin/env python3
"""Turns gpurun_out/<tag>_* (written by profiles/collect.sh on the GPU box) into tracked summaries:
   profiles/<tag>_bench.json, profiles/<tag>_kernel_stats.txt, profiles/<tag>_pmc.txt, profiles/<tag>_calib.txt,
   profiles/hbm_traffic.json (read by bench.py: measured HBM bytes per launch + issue-side fractions of the dominant kernel)."""
import collections, csv, glob, json, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else "r05"
G = os.path.join(ROOT, "gpurun_out"); P = os.path.join(ROOT, "profiles")
# the commit the collection ran at: profiles/.collect_commit is written next to the gpurun call (the GPU box has no .git) and travels with the snapshot;
# collect.sh copies it into the output directory
try:
    COMMIT = open(os.path.join(G, f"{tag}_commit.txt")).read().strip()
except OSError:
    COMMIT = None
N_SIMD, N_CU = 1024, 256          # MI355X: 256 CUs x 4 SIMD-32
line = [l for l in open(os.path.join(G, f"{tag}_bench.json")) if l.startswith("{")][-1]
open(os.path.join(P, f"{tag}_bench.json"), "w").write(line)
# (the traces are summarised on the GPU box by collect.sh -- profiles/summarize_rocprof.py -- because the databases are too large to bring back)
txt = open(os.path.join(G, f"{tag}_trace_summary.txt")).read()
open(os.path.join(P, f"{tag}_kernel_stats.txt"), "w").write(txt.replace(ROOT + "/", "").replace("/root/repo/", ""))


def pmc(prefix, d):
    fs = glob.glob(os.path.join(G, f"{tag}_{prefix}_{d}", "*", "*_counter_collection.csv"))
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    if not fs:
        return agg
    for r in csv.DictReader(open(max(fs, key=os.path.getmtime))):
        k = r["Kernel_Name"].replace("void pcu::", "").split("(")[0].replace("pcu::", "")
        if k.startswith("k_"): agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
    return agg


def mean(v):
    return sum(v) / len(v) if v else 0.0


# ---- calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (profiles/calib/calib_gather.hip: 1 GiB per kernel)
GIB = float(1 << 30)
cal = ["# profiles/calib/calib_gather: every kernel touches each 16-byte record of a 1 GiB array exactly once (true HBM bytes: 1 GiB)",
       "# reported = counter x 1024 (FETCH_SIZE / WRITE_SIZE are in KB); factor = true bytes / reported bytes"]
cf, cw, ct = pmc("calib", "fetch"), pmc("calib", "write"), pmc("calib", "tcc")
factor = {}
for k in sorted(set(cf) | set(cw)):
    f = mean(cf[k].get("FETCH_SIZE", [])) * 1024; w = mean(cw[k].get("WRITE_SIZE", [])) * 1024
    raw = "  ".join(f"{c}={mean(v):.4g}" for c, v in ct.get(k, {}).items())
    if k.startswith("k_w"):
        factor[k] = GIB / w if w else None
        cal.append(f"{k:14s} WRITE_SIZE -> {w / GIB:6.3f} GiB reported, factor {factor[k]}   (FETCH_SIZE {f / GIB:.3f} GiB)  {raw}")
    else:
        factor[k] = GIB / f if f else None
        cal.append(f"{k:14s} FETCH_SIZE -> {f / GIB:6.3f} GiB reported, factor {factor[k]}  {raw}")
open(os.path.join(P, f"{tag}_calib.txt"), "w").write("\n".join(cal) + "\n")
f_stream = factor.get("k_stream16") or 2.0          # coalesced reads (index-build passes)
f_gather = factor.get("k_gather12") or factor.get("k_gather16") or f_stream      # per-lane record gathers (search kernels)
w_stream = factor.get("k_wstream16") or 1.0
w_scatter = factor.get("k_wscatter16") or w_stream

out = ["# rocprofv3 --pmc passes (separate runs of `bench.py --steps 5 --warmup 2`), mean per dispatch",
       f"# FETCH_SIZE / WRITE_SIZE are in KB. Calibrated on this box (profiles/{tag}_calib.txt): true bytes = reported x factor,",
       f"#   coalesced reads x{f_stream:.3f}, 12/16-byte per-lane gathers x{f_gather:.3f}, coalesced writes x{w_stream:.3f}, scattered 16-byte writes x{w_scatter:.3f}.",
       "# hbm_bytes = FETCH_SIZE*1024*f_read + WRITE_SIZE*1024*f_write with the gather factors for k_search*, the streaming ones otherwise."]
fetch, write = pmc("pmc", "fetch"), pmc("pmc", "write")
sq = pmc("pmc", "sq")
for extra_pass in ("sq2", "tcc", "ta", "tcp"):
    for k, cs in pmc("pmc", extra_pass).items():
        for c, v in cs.items(): sq[k][c] = v
traffic = {}
for k in fetch:
    f = mean(fetch[k]["FETCH_SIZE"]); w = mean(write[k]["WRITE_SIZE"]) if k in write else 0
    gather = k.startswith("k_search")
    traffic[k] = f * 1024 * (f_gather if gather else f_stream) + w * 1024 * (w_scatter if k.startswith("k_bucket_scatter") else w_stream)
    extra = "  ".join(f"{c}={mean(v):.4g}" for c, v in sorted(sq.get(k, {}).items()))
    out.append(f"{k:40s} n={len(fetch[k]['FETCH_SIZE']):3d} FETCH_SIZE={f:10.1f} WRITE_SIZE={w:10.1f} hbm_bytes={traffic[k]:.4g}  {extra}")
open(os.path.join(P, f"{tag}_pmc.txt"), "w").write("\n".join(out) + "\n")

dom = [k for k in traffic if k.startswith("k_search1_flat<float")]
rep = {"commit": COMMIT, "source": f"profiles/{tag}_pmc.txt", "counters_source": f"profiles/{tag}_pmc.txt + profiles/{tag}_calib.txt",
       "k_search1_flat_f32_bytes_per_launch": traffic[dom[0]] if dom else None,
       "index_build_bytes_per_step": sum(v for k, v in traffic.items() if k.startswith(("k_bbox", "k_make_grid", "k_bucket"))),      # every pass is one launch for both clouds
       "note": "one launch = both directions of the 1M-vs-1M Chamfer step; calibrated FETCH_SIZE + WRITE_SIZE, see the header of the source file"}
if dom:
    c = {k: mean(v) for k, v in sq[dom[0]].items()}
    # kernel cycles: SQ_BUSY_CYCLES is summed over the 32 shader engines (each SQ counts the cycles it was busy); GRBM_GUI_ACTIVE is
    # summed over the 8 XCDs and includes the profiler's per-dispatch idle time, so it overstates short kernels
    cyc = c.get("SQ_BUSY_CYCLES", 0.0) / 32.0 or c.get("GRBM_GUI_ACTIVE", 0.0) / 8.0
    rep["kernel_cycles"] = cyc
    if cyc and c.get("SQ_INSTS_VALU"):
        rep["valu_insts_per_wave"] = c["SQ_INSTS_VALU"] / c.get("SQ_WAVES", 1.0)
        rep["salu_insts_per_wave"] = c.get("SQ_INSTS_SALU", 0.0) / c.get("SQ_WAVES", 1.0)
        # "2 cycles per wave64 VALU instruction" (MI355X_MICROARCH.md, wave scheduling) is the best case (see valu_busy_frac below):
        rep["valu_issue_frac_at_2_cycles"] = 2.0 * c["SQ_INSTS_VALU"] / (N_SIMD * cyc)
        rep["salu_issue_frac"] = c.get("SQ_INSTS_SALU", 0.0) / (N_CU * cyc)     # one scalar instruction per cycle per CU
    if c.get("SQ_ACTIVE_INST_VALU"):
        rep["active_lane_frac"] = c.get("SQ_THREAD_CYCLES_VALU", 0.0) / (64.0 * c["SQ_ACTIVE_INST_VALU"])
    if cyc and c.get("SQ_INSTS_VALU"):
        # SQ_ACTIVE_INST_VALU counts INSTRUCTIONS (it equals SQ_INSTS_VALU for every instruction kind, profiles/r03_valu_rate.txt), not busy
        # cycles: round 2's "x 4" reading was wrong. Measured issue cost per wave64 instruction per SIMD: 2.2-2.4 cycles for the fast class
        # (fma / add / mul / mov / and), 4.1 for the slow class (min / max / min3, compares, v_cndmask_e64, packed f32, lshl_add, or3).
        # The VALU's busy fraction therefore lies between the two bounds; `valu_busy_frac` is the estimate for the kernel's static mix
        # (45 % slow-class instructions: 0.55 x 2.3 + 0.45 x 4.1 = 3.1 cycles).
        per_simd = c["SQ_INSTS_VALU"] / N_SIMD
        rep["valu_busy_frac_bounds"] = [2.25 * per_simd / cyc, 4.1 * per_simd / cyc]
        rep["valu_busy_frac"] = 3.1 * per_simd / cyc
        rep["valu_cycles_per_inst_source"] = "profiles/r03_valu_rate.txt"
    if c.get("TA_BUSY_avr") and cyc:
        rep["ta_busy_frac"] = c["TA_BUSY_avr"] / cyc
    if c.get("SQ_INSTS_VMEM_RD") and c.get("SQ_WAVES"):
        rep["vmem_insts_per_wave"] = c["SQ_INSTS_VMEM_RD"] / c["SQ_WAVES"]
json.dump(rep, open(os.path.join(P, "hbm_traffic.json"), "w"), indent=1)
# ---- the other configs: per-config kernel statistics, HBM bytes per launch, dominant kernel (read by bench.py --config)
cfgk = {}
for c_ in ("c1", "c2", "c3", "c4", "c5", "gauss", "cluster", "outlier"):
    kj = os.path.join(G, f"{tag}_{c_}_trace_kernels.json")
    if not os.path.exists(kj):
        continue
    txt_c = open(os.path.join(G, f"{tag}_{c_}_trace_summary.txt")).read().split("\n\n")[0].replace("/root/repo/", "")
    rows = [(r["name"], r["calls"], r["total_us"], r["avg_us"]) for r in json.load(open(kj))]
    n_calls = 7                                                   # bench.py --config c --steps 4 --warmup 2 --no-parity --no-kernel-events: 1 initial + 2 warm-up + 4 timed calls
    def short(n): return n.replace("void pcu::", "").replace("pcu::", "").split("(")[0]
    fetch_c, write_c = pmc(f"{c_}_pmc", "fetch"), pmc(f"{c_}_pmc", "write")
    ks = {}
    for name, calls, tot, avg in rows:
        k = short(name)
        if not k.startswith("k_"): continue
        f = mean(fetch_c.get(k, {}).get("FETCH_SIZE", [])); w = mean(write_c.get(k, {}).get("WRITE_SIZE", []))
        gather = k.startswith(("k_search", "k_kd_search"))
        ks[k] = {"calls": calls, "avg_us": avg, "total_us": tot, "launches_per_call": calls / n_calls,          # (top_kernels is in microseconds)
                 "hbm_bytes_per_launch": f * 1024 * (f_gather if gather else f_stream) + w * 1024 * w_stream if (f or w) else None}
    if not ks: continue
    dom_c = max(ks, key=lambda k: ks[k]["total_us"])
    cfgk[c_] = {"commit": COMMIT, "dominant": dom_c, "dominant_avg_us": ks[dom_c]["avg_us"], "dominant_launches_per_call": ks[dom_c]["launches_per_call"],
                "dominant_hbm_bytes_per_launch": ks[dom_c]["hbm_bytes_per_launch"],
                "gpu_us_per_call": sum(v["total_us"] for v in ks.values()) / n_calls,
                "hbm_bytes_per_call": sum((v["hbm_bytes_per_launch"] or 0) * v["launches_per_call"] for v in ks.values()),
                "source": f"profiles/{tag}_{c_}_kernel_stats.txt, profiles/{tag}_{c_}_pmc.txt"}
    open(os.path.join(P, f"{tag}_{c_}_kernel_stats.txt"), "w").write(
        f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {c_} --steps 4 --warmup 2 --no-parity --no-kernel-events ({n_calls} calls of the operator)\n" + txt_c.replace(ROOT + "/", "") + "\n")
    lines_c = [f"# rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes) of `bench.py --config {c_} --steps 4 --warmup 2 --no-parity --no-kernel-events`, mean per dispatch; KB, calibrated as in {tag}_pmc.txt",
               f"# per call of the operator: GPU time {cfgk[c_]['gpu_us_per_call']:.1f} us, HBM traffic {cfgk[c_]['hbm_bytes_per_call'] / 1e6:.1f} MB; dominant kernel: {dom_c}"]
    for k, v in sorted(ks.items(), key=lambda kv: -kv[1]["total_us"]):
        f = mean(fetch_c.get(k, {}).get("FETCH_SIZE", [])); w = mean(write_c.get(k, {}).get("WRITE_SIZE", []))
        hb = v["hbm_bytes_per_launch"]
        lines_c.append(f"{k:44s} launches/call {v['launches_per_call']:6.2f} avg_us {v['avg_us']:9.2f} FETCH_SIZE={f:11.1f} WRITE_SIZE={w:11.1f} hbm_bytes/launch={(hb if hb is not None else float('nan')):.4g}")
    open(os.path.join(P, f"{tag}_{c_}_pmc.txt"), "w").write("\n".join(lines_c) + "\n")
json.dump(cfgk, open(os.path.join(P, "config_kernels.json"), "w"), indent=1)
# ---- the widened rows: kernel statistics only
for c_ in ("normals", "voxel", "sinkhorn", "morton"):
    sm = os.path.join(G, f"{tag}_{c_}_trace_summary.txt")
    if os.path.exists(sm):
        txt_c = open(sm).read().split("\n\n")[0].replace("/root/repo/", "").replace(ROOT + "/", "")
        open(os.path.join(P, f"{tag}_{c_}_kernel_stats.txt"), "w").write(
            f"# rocprofv3 --kernel-trace --stats -- python bench.py --config {c_} --steps 4 --warmup 2 --no-parity --no-kernel-events (7 calls of the operator; commit {COMMIT})\n" + txt_c + "\n")
cfg = os.path.join(G, f"{tag}_configs.jsonl")
if os.path.exists(cfg):
    open(os.path.join(P, f"{tag}_configs.jsonl"), "w").write(open(cfg).read())
print(line[:600]); print("\n".join(cal)); print("\n".join(out[4:14])); print(json.dumps(rep, indent=1))
