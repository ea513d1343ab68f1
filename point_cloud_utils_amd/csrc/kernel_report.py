#!/usr/bin/env python3
"""Print per-kernel VGPR/SGPR/scratch/LDS from a gfx950 code object or .s file (uses llvm-readelf --notes).
Also counts v_fma/v_fmac/v_mad in each kernel body when given the .s (the search kernels must have none)."""
import re, subprocess, sys

def demangle(n):
    try:
        return subprocess.run(["c++filt", n], capture_output=True, text=True).stdout.strip() or n
    except FileNotFoundError:
        return n

def report_s(path):
    s = open(path).read()
    rows = []
    for m in re.finditer(r"^(\S+):\s*; @\1\n(.*?)\n\s*s_endpgm", s, re.S | re.M):
        name, body = m.group(1), m.group(2)
        fma = len(re.findall(r"\bv_(fma|fmac|mad|pk_fma)\w*_f(32|64)", body))
        rows.append((name, fma))
    meta = s[s.find("amdhsa.kernels"):]
    info = {}
    for blk in meta.split("- .agpr_count:")[1:]:
        g = lambda k: (re.search(r"\." + k + r":\s+(\S+)", blk) or [None, "?"])[1]
        info[g("name")] = (g("vgpr_count"), blk.split()[0], g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
    for name, fma in rows:
        v = info.get(name, ("?",) * 5)
        print(f"{demangle(name)[:90]:90s} vgpr={v[0]:>3} agpr={v[1]:>3} sgpr={v[2]:>3} scratch={v[3]:>4} lds={v[4]:>5} fma_ops={fma}")

if __name__ == "__main__":
    report_s(sys.argv[1])
