#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3i}
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
run() { name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 $Q > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print("$name:", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("bilinear_wgrad","edge_seg_bwd","edge_gj","edge_ge","edge_gw","rows_ge","rows_gw","rows_dw")})
PY
}
run split128 CGAT_SEG_BWD_SPLIT=1
run split128_hi CGAT_SIDE_PRIORITY=-1
run split256_hi CGAT_SIDE_PRIORITY=-1 CGAT_SIDE_WGRAD_WGS=256
run split192_hi CGAT_SIDE_PRIORITY=-1 CGAT_SIDE_WGRAD_WGS=192
run split96 CGAT_SIDE_WGRAD_WGS=96
run split64 CGAT_SIDE_WGRAD_WGS=64
run split256_lo CGAT_SIDE_PRIORITY=1 CGAT_SIDE_WGRAD_WGS=256
run mono256_hi CGAT_SEG_BWD_SPLIT=0 CGAT_SIDE_PRIORITY=-1 CGAT_SIDE_WGRAD_WGS=256
