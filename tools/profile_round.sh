#!/bin/bash
# Secondary evidence of a round, collected on the GPU box (only gpurun_out/ travels back, so every database is summarised there):
#   loss kernels at the cfg4 shape (rocprofv3 kernel stats), Lovasz, cfg5 one-pass multiscale, per-rank cost of the sharded merge,
#   microbenchmarks.   usage: bash tools/profile_round.sh r02
set -u
TAG=${1:-r03}
R=/root/repo
O=$R/gpurun_out/profiles
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_losses -o run -- python $R/tools/prof_losses.py > /tmp/prof_losses.log 2>&1
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/prof_losses.py  (every loss at [32,16,512,512], forward and forward+backward; Lovasz at [4,16,512,512])"; echo; timeout 60 python $R/tools/rocprof_summary.py /tmp/prof_losses/run_results.db; } > $O/${TAG}_losses_kernel_stats.md
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_lov -o run -- python $R/tools/prof_lovasz.py > /tmp/prof_lov.log 2>&1
{ echo "# $TAG: rocprofv3 --kernel-trace --stats -- python tools/prof_lovasz.py  (LovaszLoss [4,16,512,512] + BinaryLovaszLoss [4,512,512], 20 x forward+backward, 20 x forward)"; echo; timeout 60 python $R/tools/rocprof_summary.py /tmp/prof_lov/run_results.db | head -14; echo; grep "LovaszLoss forward" /tmp/prof_lov.log; } > $O/${TAG}_lovasz_kernel_stats.md
timeout 400 rocprofv3 --kernel-trace --stats -d /tmp/prof_cfg5 -o run -- python $R/tools/bench_cfg5.py > /tmp/prof_cfg5.log 2>&1
{ echo "# $TAG: BASELINE configs[4] (multiscale 0.75 / 1.0 / 1.25 + fliplr, 4096 x 4096, C = 4): python tools/bench_cfg5.py"; echo; timeout 120 python $R/tools/bench_cfg5.py 2>/dev/null | grep -v amdgpu; echo; echo "rocprofv3 kernel stats of the same script:"; echo; timeout 60 python $R/tools/rocprof_summary.py /tmp/prof_cfg5/run_results.db | head -9; } > $O/${TAG}_cfg5.md
cd $R
{ for w in 8 4 2; do timeout 200 python tools/shard_sim.py --world $w 2>/dev/null | grep -v amdgpu; echo; done; echo "--- incremental accumulate + exchange path (round 1) for comparison"; timeout 200 python tools/shard_sim.py --world 8 --no-defer 2>/dev/null | grep -v amdgpu; } > $O/${TAG}_shard_sim.txt
timeout 200 python tools/bench_losses.py 2>/dev/null | grep -v amdgpu > $O/${TAG}_losses_microbench.txt
tools/build/valu_probe > $O/${TAG}_valu_probe.txt 2>&1
[ "${PTB_PROFILE_LIGHT:-0}" = "1" ] && { ls -la $O; exit 0; }      # (no --pmc passes: ~2 minutes of box time each)
# PMC (own passes, kernel trace only): vector-ALU occupancy of the two loss kernels section 3.4 of DESIGN.md discusses
for w in "fused" "cefocal bwd"; do t=$(echo $w | tr " " "_"); bash tools/pmc_cmd.sh $t "VALUBusy" "SQ_INSTS_VALU SQ_WAVES SQ_INSTS_SALU" "MemUnitBusy" -- python $R/tools/prof_one_loss.py $w; done
{ echo "# $TAG: rocprofv3 --pmc <group> --kernel-trace -- python tools/prof_one_loss.py {fused | cefocal bwd}  (cfg4 shape; one counter group per run)"; echo;
  for f in $(find $R/gpurun_out/pmc -name "*.db" | sort); do python tools/pmc_summary.py $f ptb:: ; done; } > $O/${TAG}_losses_pmc.md 2>&1
rm -rf $R/gpurun_out/pmc
ls -la $O
