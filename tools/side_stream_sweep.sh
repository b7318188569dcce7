cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2t}; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_golden.py -q -m gpu -x -k "full_gradient or nodes_layer or config1" > $O/t_g.log 2>&1; echo "golden subset rc=$?"; tail -n 4 $O/t_g.log | cut -c1-300
run() { # name, env...
  n=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/b_$n.json 2> $O/b_$n.err
  python - <<PY
import json
d=json.loads(open("$O/b_$n.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("$n", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("bilinear_wgrad","edge_seg_bwd","edge_ge","edge_gw","rows_dw")})
PY
}
run default X=1
run dw_side CGAT_SIDE_DW=1
run wgs256 CGAT_SIDE_WGRAD_WGS=256
run wgs192 CGAT_SIDE_WGRAD_WGS=192
run wgs160 CGAT_SIDE_WGRAD_WGS=160
run wgs256_dwside CGAT_SIDE_WGRAD_WGS=256 CGAT_SIDE_DW=1
run serial CGAT_OVERLAP_WGRAD=0
run default2 X=1
