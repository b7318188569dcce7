#!/bin/bash
# Dev tool (GPU box): the weight-gradient side stream's knobs on the headline bench leg, same box.
# usage: bash tools/side_stream_sweep.sh [out-dir under gpurun_out] [set: base | units]
cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-side_sweep}; mkdir -p $O
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
d=json.loads(open("$O/b_$n.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$n", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("bilinear_wgrad","edge_seg_bwd","edge_gj","edge_ge","edge_gw","rows_dw")})
PY
}
S="CGAT_OVERLAP_WGRAD=1"
if [ "${2:-base}" = base ]; then
run serial CGAT_OVERLAP_WGRAD=0
run side128 $S
run side256 $S CGAT_SIDE_WGRAD_WGS=256
run side256_prio $S CGAT_SIDE_WGRAD_WGS=256 CGAT_SIDE_PRIORITY=-1
run side128_prio $S CGAT_SIDE_PRIORITY=-1
run side256_dw $S CGAT_SIDE_WGRAD_WGS=256 CGAT_SIDE_DW=1
run side128_dw $S CGAT_SIDE_DW=1
run side256_noearly $S CGAT_SIDE_WGRAD_WGS=256 CGAT_SIDE_EARLY_PREP=0
run serial2 CGAT_OVERLAP_WGRAD=0
run side128b $S
else
run serial CGAT_OVERLAP_WGRAD=0
run serial_u512 CGAT_OVERLAP_WGRAD=0 CGAT_WGC_UNITS=512
run side128_dw_prio $S CGAT_SIDE_DW=1 CGAT_SIDE_PRIORITY=-1
run u512_w128_dw $S CGAT_WGC_UNITS=512 CGAT_SIDE_WGRAD_WGS=128 CGAT_SIDE_DW=1
run u512_w171_dw $S CGAT_WGC_UNITS=512 CGAT_SIDE_WGRAD_WGS=171 CGAT_SIDE_DW=1
run u512_w171 $S CGAT_WGC_UNITS=512 CGAT_SIDE_WGRAD_WGS=171
run u512_w171_dw_prio $S CGAT_WGC_UNITS=512 CGAT_SIDE_WGRAD_WGS=171 CGAT_SIDE_DW=1 CGAT_SIDE_PRIORITY=-1
run u512_w205_dw $S CGAT_WGC_UNITS=512 CGAT_SIDE_WGRAD_WGS=205 CGAT_SIDE_DW=1
run u768_w154_dw $S CGAT_WGC_UNITS=768 CGAT_SIDE_WGRAD_WGS=154 CGAT_SIDE_DW=1
run u768_w192_dw $S CGAT_WGC_UNITS=768 CGAT_SIDE_WGRAD_WGS=192 CGAT_SIDE_DW=1
run side128_dw $S CGAT_SIDE_DW=1
run serial2 CGAT_OVERLAP_WGRAD=0
fi
