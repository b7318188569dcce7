cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2h}; mkdir -p $O
for w in 64 88 128 168 256; do
  CGAT_SIDE_WGRAD_WGS=$w python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/bench_w$w.json 2> $O/bench_w$w.err
  python - <<PY
import json
d=json.loads(open("$O/bench_w$w.json").read().strip().splitlines()[-1])
print($w, round(d["ms_per_step"],3), {k:v["ms_per_step"] for k,v in d["kernel_ms_per_step"].items() if k in ("bilinear_wgrad","edge_seg_bwd","edge_ge","edge_gw","gemm_f32")})
PY
done
