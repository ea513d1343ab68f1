#!/bin/bash
# Disassemble the gfx950 code objects of the built library into /tmp/pcu_co{0,1}.s and list kernel code sizes: scratch/disasm.sh [pattern]
L=/opt/rocm/lib/llvm/bin; LIB=point_cloud_utils_amd/libpcu_hip.so
$L/llvm-objcopy -O binary --only-section=.hip_fatbin $LIB /tmp/fat.bin
python3 - <<'PY'
import re,subprocess
blob=open('/tmp/fat.bin','rb').read(); magic=b"__CLANG_OFFLOAD_BUNDLE__"
st=[m.start() for m in re.finditer(re.escape(magic),blob)]
for i,a in enumerate(st):
    open(f'/tmp/fat{i}.bin','wb').write(blob[a:st[i+1] if i+1<len(st) else len(blob)])
    subprocess.run(["/opt/rocm/lib/llvm/bin/clang-offload-bundler","--unbundle","--type=o",f"--input=/tmp/fat{i}.bin","--targets=hipv4-amdgcn-amd-amdhsa--gfx950",f"--output=/tmp/pcu_co{i}.co"],check=True)
    open(f'/tmp/pcu_co{i}.s','w').write(subprocess.run(["/opt/rocm/lib/llvm/bin/llvm-objdump","-d",f"/tmp/pcu_co{i}.co"],capture_output=True,text=True).stdout)
PY
for i in 0 1; do $L/llvm-readelf -s --wide /tmp/pcu_co$i.co | awk '$4=="FUNC"{print $3, $8}' | grep -i "${1:-k_}" | sort -n | tail -40; done
