#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3h}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_golden.py tests/test_chunked.py -m gpu -q -p no:cacheprovider -x -k "nodes_layer or determinism or side_stream or million or ragged or irregular or config5 or rebuilt or base" > $O/tests.log 2>&1
tail -5 $O/tests.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
run() { # name, env...
  name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 $Q > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print("$name:", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("bilinear_wgrad","edge_seg_bwd","edge_gj","edge_ge","edge_gw","rows_ge","rows_gw","rows_dw")})
PY
}
run mono128 CGAT_SEG_BWD_SPLIT=0
run split128 CGAT_SEG_BWD_SPLIT=1
run split256 CGAT_SEG_BWD_SPLIT=1 CGAT_SIDE_WGRAD_WGS=256
run mono256 CGAT_SEG_BWD_SPLIT=0 CGAT_SIDE_WGRAD_WGS=256
run split_serial CGAT_SEG_BWD_SPLIT=1 CGAT_OVERLAP_WGRAD=0
run mono_serial CGAT_SEG_BWD_SPLIT=0 CGAT_OVERLAP_WGRAD=0
