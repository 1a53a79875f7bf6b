#!/bin/bash
# The placement experiment of tools/placement_pmc.sh with the ADDRESS-TRANSLATION counters of the vector L1 (UTCL1 = its TLB; misses go
# to the UTCL2): does a slow pool translate worse?   bash tools/placement_tlb.sh [K] [ROUNDS] [IMAGES]  -> gpurun_out/placement_tlb/
set -u
K=${1:-6}; R=${2:-2}; I=${3:-6}
ROOT=/root/repo
O=$ROOT/gpurun_out/placement_tlb
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/tools/placement_pmc.py $K 3 8 2>&1 | grep -v amdgpu > $O/plain.txt
n=0
for G in "TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" "TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum TCP_UTCL1_STALL_INFLIGHT_MAX_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum" "TCP_UTCL1_STALL_MULTI_MISS_sum TCP_UTCL1_SERIALIZATION_STALL_sum TCP_UTCL1_THRASHING_STALL_sum TCP_UTCL1_LFIFO_FULL_sum" "GRBM_UTCL2_BUSY GRBM_GUI_ACTIVE TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  rm -rf /tmp/pt_$n
  timeout 500 rocprofv3 --pmc $G --kernel-trace -d /tmp/pt_$n -o run -- python $ROOT/tools/placement_pmc.py $K $R $I > $O/pass_$n.log 2>&1
  db=$(find /tmp/pt_$n -name "*.db" | head -1)
  { echo "## counters: $G"; grep -E "^(round|va|K=)" $O/pass_$n.log; [ -n "$db" ] && timeout 120 python $ROOT/tools/placement_pmc_report.py $db $K $R $I; echo; } > $O/pass_$n.txt 2>&1
  rm -rf /tmp/pt_$n
  n=$((n+1))
done
cat $O/plain.txt $O/pass_*.txt > $O/summary.txt
