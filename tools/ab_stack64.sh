#!/bin/bash
# Dev tool (GPU box): same-box A/B of one environment switch on the 64-crystal 4-layer step (eager and hipGraph replay).
# usage: tools/ab_stack64.sh VAR "v1 v2 ..." [reps]
VAR=$1; VALS=$2; REPS=${3:-2}
for r in $(seq 1 $REPS); do
  for v in $VALS; do
    env $VAR=$v python bench.py --workload stack --graphs 64 --hipgraph --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); lb=d['launch_bound']
print('$VAR=$v stack64 eager', round(d['ms_per_step'],3), 'graph', lb['hipgraph']['ms_per_step'], 'launches', lb['library_kernel_launches_per_step'])
"
  done
done
