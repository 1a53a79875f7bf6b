#!/bin/bash
# tools/placement_pmc.sh, per TCC instance: bash tools/placement_channels.sh [K] [ROUNDS] [IMAGES]  -> gpurun_out/placement_channels/
set -u
K=${1:-6}; R=${2:-2}; I=${3:-6}
ROOT=/root/repo
O=$ROOT/gpurun_out/placement_channels
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
n=0
for G in "TCC_EA0_RDREQ TCC_EA0_RDREQ_LEVEL TCC_REQ" "TCC_BUSY TCC_TAG_STALL TCC_EA0_RDREQ_DRAM_CREDIT_STALL" "TCC_MISS TCC_HIT TCC_EA0_WRREQ"; do
  rm -rf /tmp/pc_$n
  timeout 500 rocprofv3 --pmc $G --kernel-trace -d /tmp/pc_$n -o run -- python $ROOT/tools/placement_pmc.py $K $R $I > $O/pass_$n.log 2>&1
  db=$(find /tmp/pc_$n -name "*.db" | head -1)
  { echo "# counters: $G"; grep -E "^(round|va|K=)" $O/pass_$n.log; echo; [ -n "$db" ] && timeout 200 python $ROOT/tools/placement_channels.py $db $K $R $I; echo; } > $O/pass_$n.txt 2>&1
  rm -rf /tmp/pc_$n
  n=$((n+1))
done
cat $O/pass_*.txt > $O/summary.txt
