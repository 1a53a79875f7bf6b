#!/bin/bash
# Collect the rocprofv3 evidence for bench.py on the GPU box (run through gpurun from the repo root):
#   kernel trace (+stats) in one run, HBM / L2 / SQ counters each in their own run (PMC runs carry --kernel-trace only).
# Raw sqlite databases land in gpurun_out/prof/<tag>/; tools/profile_report.py turns them into profiles/*.md + traffic.json.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof
mkdir -p $OUT
BENCH="python /root/repo/bench.py --steps 3 --warmup 1 --no-cpu-baseline"
rocprofv3 --kernel-trace --stats -d $OUT/trace -o run -- $BENCH > $OUT/trace.log 2>&1
for C in "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_LDS_UNALIGNED_STALL"; do
  TAG=$(echo $C | cut -d" " -f1)
  rocprofv3 --pmc $C --kernel-trace -d $OUT/$TAG -o run -- $BENCH > $OUT/$TAG.log 2>&1
done
grep -h '"metric"' $OUT/trace.log | tail -1
