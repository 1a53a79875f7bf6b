#!/bin/bash
# Same-box A/B of non-temporal vs plain result stores: the default library against the -DPTB_NT_OUT=0 build
# (EXTRA="-DPTB_NT_OUT=0" tools/build_variant.sh plainstores).  Every command is bounded.
PLAIN=pytorch_toolbelt_amd/lib/alt/plainstores/libptb_hip.so
for r in 1 2; do
for L in default $PLAIN; do
  if [ $L = default ]; then unset PTB_HIP_LIB; else export PTB_HIP_LIB=$L; fi
  echo "=== round $r lib=$L"
  timeout 200 python bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-secondary < /dev/null 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('headline', d['value'], d['ms_per_step'], d['roofline']['avg_launch_ms'], d['roofline']['frac'])"
  timeout 200 python tools/bench_cfg5.py < /dev/null 2>&1 | grep -E "cfg5|alone"
  timeout 200 python tools/bench_losses.py < /dev/null 2>&1 | tail -12
done; done
unset PTB_HIP_LIB
timeout 120 python tools/ab_bwd_nt.py < /dev/null 2>&1 | tail -6
