#!/bin/bash
# Headline bench under different values of one environment variable:  bash tools/ab_env.sh VAR "<v0> <v1> ..." [rounds]
VAR=$1; VALS=$2; ROUNDS=${3:-2}
for r in $(seq $ROUNDS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-variants --no-cpu-baseline --repeats 3 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$VAR" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(f"{sys.argv[1]}={sys.argv[2]}: {d['ms_per_step']} ms  frac {d['roofline']['frac']}  of box read ceiling {d['roofline'].get('frac_of_box_read_ceiling')}")
PY
  done
done
