#!/bin/bash
# Dev tool (GPU box): phase stamps of the f16x3c weight-gradient kernel (tools/wgrad_stamps.py) for each WGC_ABL value given
cd "$(dirname "$0")/.."
for abl in "$@"; do
  touch cgat_amd/csrc/wgradc.hip
  CGAT_HIPCC_FLAGS="-DCGAT_DEV_ABLATIONS -DWGC_STAMPS -DWGC_ABL=$abl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed for $abl"; continue; }
  echo "== WGC_ABL=$abl"
  python tools/wgrad_stamps.py 2>&1 | sed -n 11,19p
done
touch cgat_amd/csrc/wgradc.hip; bash cgat_amd/build_lib.sh > /dev/null 2>&1
