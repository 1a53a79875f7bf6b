set -u
cd /tmp && export TMPDIR=/tmp
OUT=/root/repo/gpurun_out/prof_lds
rm -rf $OUT; mkdir -p $OUT
BENCH="python /root/repo/bench.py --steps 3 --warmup 1 --repeats 1 --ramp-max-ms 600 --no-cpu-baseline --no-secondary --no-variants --placement-tries 1"
timeout 400 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS --kernel-trace -d $OUT/lds -o run -- $BENCH > $OUT/lds.log 2>&1
python /root/repo/tools/pmc_summary.py $(find $OUT/lds -name "*.db" | head -1) band_plan > /root/repo/gpurun_out/r05_lds_pmc.txt 2>&1
rm -rf $OUT
