#!/bin/bash
cd ${GRAFT_REPO_ROOT:-/root/repo}
R=$PWD
mkdir -p gpurun_out/pmc
cd /tmp && export TMPDIR=/tmp
rocprofv3 -L > $R/gpurun_out/pmc/avail.txt 2>&1
for grp in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" \
           "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_BUSY_CYCLES SQ_INST_CYCLES_VMEM SQ_WAVE_CYCLES GRBM_GUI_ACTIVE" \
           "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum TA_BUSY_avr" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TA_TCP_STATE_READ_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_GATE_EN1_sum TCP_GATE_EN2_sum"; do
  tag=$(echo $grp | cut -d' ' -f1)
  timeout 200 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $R/gpurun_out/pmc/$tag -- python $R/bench.py --steps 4 --warmup 2 --no-cpu-baseline > $R/gpurun_out/pmc/$tag.log 2>&1
done
cd $R
python - <<'PY'
import csv, glob, collections
for d in sorted(glob.glob('gpurun_out/pmc/*/')):
    fs = glob.glob(d + '**/*counter_collection.csv', recursive=True)
    if not fs: print(d, 'no csv'); continue
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for row in csv.DictReader(open(fs[0])):
        k = row['Kernel_Name'][:40]
        acc[k][row['Counter_Name']].append(float(row['Counter_Value']))
    for k in acc:
        if 'k_search1_flat' in k or 'k_bucket_count' in k:
            print(k, {c: round(sum(v)/len(v)) for c, v in acc[k].items()}, 'n=', len(next(iter(acc[k].values()))))
PY
grep -c . gpurun_out/pmc/avail.txt
