#!/bin/bash
# Dev tool (GPU box): timing-only ablations of the forward contraction kernel (csrc/bilinear.hip, -DRC_ABL=bits: 1 no LDS-DMA,
# 2 no fragment reads, 4 no matrix instructions, 8 no flush, 16 no wait + barrier), library rebuilt on the box per variant;
# prints the kernel's ms per step on the headline leg (4 launches).  Results are wrong by construction.
cd $GRAFT_REPO_ROOT
for abl in 0 1 2 4 8 16 3 5 6 12 20 7 15 31 0; do
  touch cgat_amd/csrc/bilinear.hip
  CGAT_HIPCC_FLAGS="-DCGAT_DEV_ABLATIONS -DRC_ABL=$abl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "RC_ABL=$abl build failed"; continue; }
  timeout 300 python bench.py --no-cpu-baseline --no-extra-legs --no-exclusive-pass --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
try:
    d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
    print('RC_ABL=$abl', 'bilinear_rows', k['bilinear_rows']['ms_per_step'], 'ms/step;', 'step', round(d['ms_per_step'],2))
except Exception as e: print('RC_ABL=$abl failed', e)
"
done
# leave the product build behind (build_lib.sh rebuilds when the flags it was built with change)
bash cgat_amd/build_lib.sh > /dev/null 2>&1
