#!/bin/bash
# Interleaved A/B of bench.py variants on one box: tools/ab_bench.sh "<flags A>" "<flags B>" [rounds]
A="$1"; B="$2"; R=${3:-3}
for i in $(seq $R); do for F in "$A" "$B"; do
python bench.py --steps 20 --warmup 3 --no-cpu-baseline $F 2>&1 | grep metric | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('[%s]' % '$F', d['value'], d['ms_per_step'], d['config']['region_hbm_frac'], d['roofline']['achieved'], d['roofline']['avg_launch_ms'])"
done; done
