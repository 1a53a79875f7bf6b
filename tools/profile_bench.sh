#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel trace (+stats) of the default command in one run; HBM / L2 / SQ counters each in their own run (PMC runs carry
#   --kernel-trace only) of the headline configuration alone (--no-variants); FETCH / WRITE once more on the unplanned merger,
#   whose merge kernel moves an exactly known byte count (the calibration of the FETCH_SIZE x 2 rule).
# tools/profile_report.py turns the databases into <tag>_kernel_stats.md, <tag>_pmc.md, traffic.json under gpurun_out/profiles
# (only gpurun_out/ travels back, and the raw databases exceed what is copied): usage  bash tools/profile_bench.sh r02
set -u
TAG=${1:-r03}
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof
rm -rf $OUT; mkdir -p $OUT
BENCH="python /root/repo/bench.py --steps 3 --warmup 1 --repeats 1 --ramp-max-ms 600 --no-cpu-baseline --no-secondary"
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $BENCH > $OUT/trace.log 2>&1
# PTB_PROFILE_LIGHT=1: HBM traffic only (FETCH_SIZE, WRITE_SIZE; a --pmc pass costs ~2 minutes of box time whatever it runs)
if [ "${PTB_PROFILE_LIGHT:-0}" = "2" ]; then      # kernel trace only (profiles/traffic.json keeps the last PMC passes and their date)
  :
elif [ "${PTB_PROFILE_LIGHT:-0}" = "1" ]; then
  for C in "FETCH_SIZE" "WRITE_SIZE"; do
    timeout 300 rocprofv3 --pmc $C --kernel-trace -d $OUT/$C -o run -- $BENCH --no-variants > $OUT/$C.log 2>&1
  done
else
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"; do
  T=$(echo $C | cut -d" " -f1)
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d $OUT/$T -o run -- $BENCH --no-variants > $OUT/$T.log 2>&1
done
for C in "FETCH_SIZE" "WRITE_SIZE"; do
  timeout 400 rocprofv3 --pmc $C --kernel-trace -d $OUT/${C}_cal -o run -- $BENCH --no-variants --unplanned > $OUT/${C}_cal.log 2>&1
done
fi
grep -h '"metric"' $OUT/trace.log | tail -1 > /root/repo/gpurun_out/bench_under_rocprof.json
PTB_PROFILE_OUT=/root/repo/gpurun_out/profiles python /root/repo/tools/profile_report.py $TAG > /root/repo/gpurun_out/profile_report.log 2>&1
du -sh $OUT | tail -1
rm -rf $OUT
