#!/bin/bash
# Does the process-to-process spread of the headline follow the TLB?  Each run: bench.py under rocprofv3 --pmc (UTCL1 translation
# misses of the band kernel) with a different amount of memory allocated first; prints ms per image next to the miss count.
cd /tmp && export TMPDIR=/tmp
for pad in $*; do
  rm -rf /tmp/tlbp
  PTB_BENCH_PAD_MB=$pad rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum --kernel-trace -d /tmp/tlbp -o run -- \
      python /root/repo/bench.py --steps 5 --warmup 2 --repeats 1 --ramp-max-ms 800 --no-cpu-baseline --no-variants > /tmp/tlb.log 2>&1
  MS=$(grep -h '"metric"' /tmp/tlb.log | tail -1 | python -c "import json,sys; print(json.loads(sys.stdin.read())['ms_per_step'])")
  MISS=$(python /root/repo/tools/pmc_summary.py $(find /tmp/tlbp -name "*.db") band_plan | grep MISS | awk '{print $(NF-4)}')
  echo "pad ${pad} MB: ${MS} ms per image, UTCL1 translation misses per band launch: ${MISS}"
done
