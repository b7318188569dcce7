#!/bin/bash
# Dev tool (GPU box): A/B of a compile-time macro of one csrc file on the headline leg: rebuilds the library on the box.
# usage: tools/ab_build_flag.sh <file stem> "<flags A>" "<flags B>" [kernel tags...]
cd $GRAFT_REPO_ROOT
F=$1; A=$2; B=$3; shift 3; TAGS="$*"
run() {
  touch cgat_amd/csrc/$F.hip
  CGAT_HIPCC_FLAGS="$1" bash cgat_amd/build_lib.sh > /dev/null 2>&1 || { echo "build failed: $1"; return; }
  for r in 1 2; do
  python bench.py --no-cpu-baseline --no-extra-legs --no-exclusive-pass --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernel_ms_per_step']
print('[$1]', round(d['ms_per_step'],3), {t:k[t]['ms_per_step'] for t in '$TAGS'.split() if t in k})
"
  done
}
run "$A"; run "$B"; run "$A"
# leave the product build behind, whatever A was: build_lib.sh rebuilds when its flags change
bash cgat_amd/build_lib.sh > /dev/null 2>&1
