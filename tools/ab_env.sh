#!/bin/bash
# Dev tool (GPU box): same-box A/B of one environment switch on the headline bench leg.
# usage: tools/ab_env.sh VAR "v1 v2 ..." [reps] [kernel tags to print...]
# Runs bench.py (timed region only) with VAR set to each value in turn, `reps` rounds, and prints ms/step + the tags' ms.
VAR=$1; VALS=$2; REPS=${3:-2}; shift 3
TAGS="$*"
for r in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --no-cpu-baseline --no-extra-legs --no-exclusive-pass --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
k=d['kernel_ms_per_step']
print('$VAR=$v', round(d['ms_per_step'],3), {t:k[t]['ms_per_step'] for t in '$TAGS'.split() if t in k})
"
  done
done
