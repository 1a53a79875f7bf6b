#!/bin/bash
# Interleaved A/B of one ptb_set_tunable key on the headline bench:  bash tools/ab_tunable.sh <key> "<v0> <v1> ..." [rounds]
KEY=$1; VALS=$2; ROUNDS=${3:-3}
for r in $(seq $ROUNDS); do
  for v in $VALS; do
    python bench.py --no-variants --no-cpu-baseline --repeats 3 --tunable $KEY=$v 2>/dev/null | tail -1 > /tmp/ab_line.json
    python - "$KEY" "$v" <<'PY'
import json, sys
d = json.loads(open("/tmp/ab_line.json").read())
print(f"tunable {sys.argv[1]}={sys.argv[2]}: {d['ms_per_step']} ms  frac {d['roofline']['frac']}  of box read ceiling {d['roofline'].get('frac_of_box_read_ceiling')}")
PY
  done
done
