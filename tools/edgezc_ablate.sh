#!/bin/bash
# Dev tool (GPU box): rebuilds csrc/edgez.hip with each EZC_ABL value (1 no Z stores, 2 no gathers, 4 no matrix instructions;
# timing only, wrong results) and prints the per-edge forward kernel's time in the headline step
cd "$(dirname "$0")/.."
for abl in "$@" 0; do
  touch cgat_amd/csrc/edgez.hip
  CGAT_HIPCC_FLAGS="-DCGAT_DEV_ABLATIONS -DEZC_ABL=$abl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed for $abl"; continue; }
  python bench.py --steps 5 --warmup 2 --no-extra-legs --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EZC_ABL=$abl edge_z ms', d['kernel_ms_per_step']['edge_z']['ms_per_step'], 'step', round(d['ms_per_step'],2))"
done
# leave the product build behind (build_lib.sh rebuilds when the flags it was built with change)
bash cgat_amd/build_lib.sh > /dev/null 2>&1
