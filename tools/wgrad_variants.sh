#!/bin/bash
# Dev tool (GPU box): rebuilds csrc/wgradc.hip with each given set of -D flags (one quoted argument per variant) and times
# the f16x3c weight gradient at 83 340 rows; the last build is the plain one.
cd "$(dirname "$0")/.."
for fl in "$@" ""; do
  touch cgat_amd/csrc/wgradc.hip
  CGAT_HIPCC_FLAGS="$fl" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed for $fl"; continue; }
  echo "== flags: $fl"
  python tools/wgrad_probe.py 83340 2>&1 | grep -E "unit  f16x3c|unit  f16x3 "
done
