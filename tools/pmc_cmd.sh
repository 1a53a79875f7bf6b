#!/bin/bash
# rocprofv3 PMC passes (each counter group in its own run, kernel trace only) for an arbitrary command:
#   tools/pmc_cmd.sh <tag> "<counters group 1>" ["<group 2>" ...] -- <command...>
# Raw databases land in gpurun_out/pmc/<tag>_<n>/; tools/rocpd_summary.py prints per-kernel averages.
set -u
TAG=$1; shift
GROUPS_=()
while [ "$1" != "--" ]; do GROUPS_+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/pmc
mkdir -p $OUT
i=0
for G in "${GROUPS_[@]}"; do
  timeout 400 rocprofv3 --pmc $G --kernel-trace -d $OUT/${TAG}_$i -o run -- "$@" > $OUT/${TAG}_$i.log 2>&1
  i=$((i+1))
done
