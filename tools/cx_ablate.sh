#!/bin/bash
# Dev tool (GPU box): timing-only ablations of the dense-layer chain kernel (csrc/chain.hip, -DCX_ABL=bits: 1 no epilogue loads,
# 2 no stores), library rebuilt on the box per variant; ms per step of the chains on the headline leg.  Results are wrong by construction.
cd $GRAFT_REPO_ROOT
for abl in 0 1 2 3 0; do
  touch cgat_amd/csrc/chain.hip
  CGAT_HIPCC_FLAGS="-DCGAT_DEV_ABLATIONS -DCX_ABL=$abl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed $abl"; continue; }
  python bench.py --no-cpu-baseline --no-extra-legs --no-exclusive-pass --steps 10 --warmup 3 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('CX_ABL=$abl mlp_chain', k['mlp_chain']['ms_per_step'], 'step', round(d['ms_per_step'],2))
"
done
# leave the product build behind (build_lib.sh rebuilds when the flags it was built with change)
bash cgat_amd/build_lib.sh > /dev/null 2>&1
