#!/bin/bash
# Dev tool (run on the GPU box): rebuilds csrc/wgradc.hip with each timing-only ablation (WGC_ABL bits: 1 no q loads,
# 2 no split arithmetic, 4 no LDS fragment reads, 8 no LDS-DMA, 16 no matrix instructions) and times the f16x3c weight
# gradient at 83 340 rows.  Results with an ablation are WRONG by construction; the last build is the real one.
cd "$(dirname "$0")/.."
for abl in "$@" 0; do
  touch cgat_amd/csrc/wgradc.hip
  CGAT_HIPCC_FLAGS="-DCGAT_DEV_ABLATIONS -DWGC_ABL=$abl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed for $abl"; continue; }
  echo "== WGC_ABL=$abl"
  python tools/wgrad_probe.py 83340 2>&1 | grep "unit  f16x3c"
done
# leave the product build behind (build_lib.sh rebuilds when the flags it was built with change)
bash cgat_amd/build_lib.sh > /dev/null 2>&1
