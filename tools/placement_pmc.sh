#!/bin/bash
# The placement experiment on the GPU box: one plain run (clean HIP-event times per pool) + one rocprofv3 --pmc pass per counter
# group of tools/placement_pmc.py; summaries land in gpurun_out/placement/.   bash tools/placement_pmc.sh [K] [ROUNDS] [IMAGES]
set -u
K=${1:-6}; R=${2:-2}; I=${3:-6}
ROOT=/root/repo
O=$ROOT/gpurun_out/placement
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 300 python $ROOT/tools/placement_pmc.py $K 3 8 2>&1 | grep -v amdgpu > $O/plain.txt
n=0
for G in "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" "TCC_EA0_WRREQ TCC_EA0_WRREQ_LEVEL TCC_EA0_WRREQ_STALL TCC_TOO_MANY_EA_WRREQS_STALL" "TCC_BUSY TCC_TAG_STALL TCC_REQ GRBM_GUI_ACTIVE" "TCC_EA0_RDREQ_32B TCC_BUBBLE TCC_MISS TCC_HIT"; do
  rm -rf /tmp/pp_$n
  timeout 500 rocprofv3 --pmc $G --kernel-trace -d /tmp/pp_$n -o run -- python $ROOT/tools/placement_pmc.py $K $R $I > $O/pass_$n.log 2>&1
  db=$(find /tmp/pp_$n -name "*.db" | head -1)
  { echo "## counters: $G"; grep -E "^(round|va|K=)" $O/pass_$n.log; [ -n "$db" ] && timeout 120 python $ROOT/tools/placement_pmc_report.py $db $K $R $I; echo; } > $O/pass_$n.txt 2>&1
  rm -rf /tmp/pp_$n
  n=$((n+1))
done
cat $O/plain.txt $O/pass_*.txt > $O/summary.txt
