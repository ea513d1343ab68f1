#!/bin/bash
# Collects the round's measurement artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/<tag>_bench.json                 headline bench.py line (with cpu_baseline + parity)
#   gpurun_out/<tag>_trace_summary.txt, _trace_kernels.json   rocprofv3 --kernel-trace --stats of the headline command, summarised on the box
#   gpurun_out/<tag>_pmc_<pass>/                rocprofv3 --pmc passes of the headline (separate runs, csv): fetch, write, tcc, sq, sq2, ta, tcp
#   gpurun_out/<tag>_<cfg>_trace/, _<cfg>_pmc_fetch/, _<cfg>_pmc_write/   the same trace + HBM-byte passes for every other BASELINE config
#                                               (c1 c2 c3 c4 c5) and the uneven-cloud cases (gauss cluster outlier)
#   gpurun_out/<tag>_calib_<pass>/              the TCC passes over profiles/calib/calib_gather (known byte counts)
#   gpurun_out/<tag>_configs.jsonl              one bench.py --config line per config (parity checked inside the line)
# then `python profiles/postprocess.py <tag>` (CPU side) turns them into the tracked summaries under profiles/.
# Counter passes never carry trace domains other than --kernel-trace (gpurun refuses --pmc with sys/hip/hsa traces).
TAG=${1:-r05}
ROOT=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$ROOT/gpurun_out
mkdir -p $OUT
# the commit of this snapshot (written by the caller before gpurun: `git describe --always --dirty > profiles/.collect_commit`; no .git on the box)
cp $ROOT/profiles/.collect_commit $OUT/${TAG}_commit.txt 2>/dev/null || echo unknown > $OUT/${TAG}_commit.txt
cd /tmp && export TMPDIR=/tmp
B="python $ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-parity --no-configs"
# (PCU_COLLECT_LINES / PCU_COLLECT_CONFIGS: subsets of the config lists below, for a quick check of the pipeline)
python $ROOT/bench.py --steps 50 --warmup 5 ${PCU_COLLECT_BENCH_FLAGS:-} > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
# (a rocprofv3 database is ~14 MB and gpurun brings back 64 MiB at most: every trace is summarised here and its database dropped)
summarise() { db=$(find $OUT/$1 -name "*results.db" | head -1); python $ROOT/profiles/summarize_rocprof.py $db --json $OUT/$1_kernels.json > $OUT/$1_summary.txt 2>/dev/null; rm -rf $OUT/$1; }
rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_trace -- python $ROOT/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-parity --no-configs > $OUT/${TAG}_trace.log 2>&1
summarise ${TAG}_trace
pass() { name=$1; shift; rocprofv3 --kernel-trace --pmc "$@" --output-format csv -d $OUT/${TAG}_pmc_$name -- $B > $OUT/${TAG}_pmc_$name.log 2>&1; }
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum TCC_HIT_sum
pass sq SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU SQ_BUSY_CYCLES SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_SCA GRBM_GUI_ACTIVE
pass sq2 SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_INSTS_VMEM_RD SQ_INSTS_SMEM SQ_INSTS_LDS SQ_ACTIVE_INST_VMEM GRBM_GUI_ACTIVE
pass ta TA_TA_BUSY_sum TA_BUSY_avr TA_FLAT_READ_WAVEFRONTS_sum
pass tcp TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum
if [ -x $ROOT/profiles/calib/calib_gather ]; then
  C=$ROOT/profiles/calib/calib_gather
  rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_calib_fetch -- $C > $OUT/${TAG}_calib_fetch.log 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_calib_write -- $C > $OUT/${TAG}_calib_write.log 2>&1
  rocprofv3 --kernel-trace --pmc TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum TCC_BUBBLE_sum --output-format csv -d $OUT/${TAG}_calib_tcc -- $C > $OUT/${TAG}_calib_tcc.log 2>&1
fi
tail -c 1500 $OUT/${TAG}_bench.json
# the other BASELINE configs, the uneven clouds and the widened rows: one bench line each (parity inside), and for the
# BASELINE configs + uneven clouds a kernel trace and the two HBM-byte counter passes of the same command
: > $OUT/${TAG}_configs.jsonl
for c in ${PCU_COLLECT_LINES:-c1 c2 c3 c4 c5 gauss cluster outlier normals morton voxel sinkhorn}; do
  timeout 400 python $ROOT/bench.py --config $c --steps 10 --warmup 2 2>/dev/null | grep '^{' >> $OUT/${TAG}_configs.jsonl
done
for c in ${PCU_COLLECT_CONFIGS:-c1 c2 c3 c4 c5 gauss cluster outlier}; do
  CB="python $ROOT/bench.py --config $c --steps 4 --warmup 2 --no-parity --no-kernel-events"
  timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${c}_trace -- $CB > $OUT/${TAG}_${c}_trace.log 2>&1
  summarise ${TAG}_${c}_trace
  timeout 400 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $OUT/${TAG}_${c}_pmc_fetch -- $CB > $OUT/${TAG}_${c}_pmc_fetch.log 2>&1
  timeout 400 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $OUT/${TAG}_${c}_pmc_write -- $CB > $OUT/${TAG}_${c}_pmc_write.log 2>&1
done
# the widened rows: kernel trace only
for c in ${PCU_COLLECT_FROWS:-normals voxel sinkhorn morton}; do
  CB="python $ROOT/bench.py --config $c --steps 4 --warmup 2 --no-parity --no-kernel-events"
  timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_${c}_trace -- $CB > $OUT/${TAG}_${c}_trace.log 2>&1
  summarise ${TAG}_${c}_trace
done
cut -c1-260 $OUT/${TAG}_configs.jsonl
