"""The no-FMA contract, checked on the shipped binary (CPU test): d2 = ((dx*dx) + (dy*dy)) + (dz*dz) with separate multiplies
and adds (L2_Simple_Adaptor::evalMetric, nanoflann.hpp:496-507, built by the reference with -msse3: no FMA). The gfx950 code object
inside libpcu_hip.so is extracted and every search kernel disassembled; a floating-point fused multiply-add (v_fma / v_fmac / v_mad /
v_pk_fma on f16 / f32 / f64) is only allowed inside the expansion of the correctly rounded sqrt of the epilogue (the Newton fix-up
steps of v_sqrt_f32 / v_rsq_f64), which is checked by count per kernel. Integer v_mad_* (address arithmetic) are not floating
point and are ignored."""
import os
import re
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LIB = os.path.join(ROOT, "point_cloud_utils_amd", "libpcu_hip.so")
LLVM = "/opt/rocm/lib/llvm/bin"
SUF = r"(_e32|_e64|_dpp|_sdwa)?\b"
FMA = re.compile(r"\bv_(pk_)?(fma|fmac|mad|madak|madmk|mac)(_mix\w*|_legacy)?_f(16|32|64)" + SUF)
# The correctly rounded sqrt expands to v_sqrt_f32 + 2 v_fma_f32 (f32) / v_rsq_f64 + 3 v_fma_f64 + 4 v_fmac_f64 (f64) on gfx950
# (hipcc 7.2): a kernel whose only fused multiply-adds are those has exactly 2 per v_sqrt_f32 and 7 per v_rsq_f64; a contracted
# distance, bound or cell computation adds to the count.
PER_SQRT_F32, PER_RSQ_F64 = 2, 7


def _code_object(tmp_path):
    for t in ("llvm-objcopy", "clang-offload-bundler", "llvm-objdump", "llvm-readelf"):
        if not os.path.exists(os.path.join(LLVM, t)):
            pytest.skip(f"{t} not in this image")
    if not os.path.exists(LIB):
        pytest.skip("libpcu_hip.so not built")
    # the library is linked from two translation units (pcu_hip.hip, search_kernels.hip): .hip_fatbin holds one offload bundle per unit
    fat = str(tmp_path / "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "-O", "binary", "--only-section=.hip_fatbin", LIB, fat], check=True)
    blob, magic = open(fat, "rb").read(), b"__CLANG_OFFLOAD_BUNDLE__"
    starts = [m.start() for m in re.finditer(re.escape(magic), blob)]
    assert starts, "no offload bundle in .hip_fatbin"
    cos = []
    for i, a in enumerate(starts):
        part, co = str(tmp_path / f"fat{i}.bin"), str(tmp_path / f"co{i}.gfx950")
        open(part, "wb").write(blob[a:starts[i + 1] if i + 1 < len(starts) else len(blob)])
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={part}",
                        "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", f"--output={co}"], check=True)
        cos.append(co)
    return cos


def test_search_kernels_have_no_fused_multiply_add(tmp_path):
    counts, all_names = {}, []
    for co in _code_object(tmp_path):
        syms = subprocess.run([f"{LLVM}/llvm-readelf", "-s", "--wide", co], capture_output=True, text=True, check=True).stdout
        names = sorted({ln.split()[-1] for ln in syms.splitlines() if " FUNC " in ln and re.search(r"k_search|k_kd_search", ln)})
        all_names += names
        _count_fma(co, names, counts)
    names = all_names
    assert len(names) >= 20, names                    # k_search1_flat x {f32, f64} x fuse modes, k_search<K>, k_search_wave<K>, kd traversals
    assert any("k_search1_flat" in n for n in names) and any("k_search_wave" in n for n in names)
    assert len(counts) == len(names)
    total = 0
    for k, c in counts.items():
        assert not c["other"], (k, c["other"][:4])
        assert c["fma32"] == PER_SQRT_F32 * c["sqrt32"], (k, c)
        assert c["fma64"] == PER_RSQ_F64 * c["rsq64"], (k, c)
        total += c["fma32"] + c["fma64"]
    assert total > 0                                  # the guard sees the expansions it accounts for (the patterns match this ISA)


def _count_fma(co, names, counts):
    for i in range(0, len(names), 8):
        out = subprocess.run([f"{LLVM}/llvm-objdump", "-d", "--no-show-raw-insn", "--disassemble-symbols=" + ",".join(names[i:i + 8]), co],
                             capture_output=True, text=True, check=True).stdout
        cur = None
        for ln in out.splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
                counts[cur] = {"sqrt32": 0, "rsq64": 0, "fma32": 0, "fma64": 0, "other": []}
                continue
            if cur is None:
                continue
            ins = ln.split("//")[0].strip()
            c = counts[cur]
            if re.search(r"\bv_sqrt_f32" + SUF, ins): c["sqrt32"] += 1
            if re.search(r"\bv_rsq_f64" + SUF, ins): c["rsq64"] += 1
            m = FMA.search(ins)
            if m:
                if m.group(1) or m.group(4) == "16": c["other"].append(ins)
                elif m.group(4) == "32": c["fma32"] += 1
                else: c["fma64"] += 1


def test_every_declared_kernel_is_gfx950(tmp_path):
    cos = _code_object(tmp_path)
    assert len(cos) == 2                              # pcu_hip.hip + search_kernels.hip
    for co in cos:
        notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
        assert "amdgcn-amd-amdhsa--gfx950" in notes
