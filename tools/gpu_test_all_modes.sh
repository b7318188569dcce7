cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2ac}; mkdir -p $O
timeout 2400 python -m pytest tests -q -m gpu > $O/t_f16x3.log 2>&1; echo "f16x3 rc=$?"; tail -n 5 $O/t_f16x3.log | cut -c1-300
CGAT_BILINEAR_MODE=bf16x6 timeout 2400 python -m pytest tests -q -m gpu > $O/t_bf16x6.log 2>&1; echo "bf16x6 rc=$?"; tail -n 5 $O/t_bf16x6.log | cut -c1-300
CGAT_BILINEAR_MODE=f32 timeout 2400 python -m pytest tests -q -m gpu > $O/t_f32.log 2>&1; echo "f32 rc=$?"; tail -n 5 $O/t_f32.log | cut -c1-300
CGAT_OVERLAP_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/b_serial.json 2> $O/b_serial.err
python - <<PY
import json
d=json.loads(open("$O/b_serial.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print("serial", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("edge_seg_bwd","edge_ge","edge_gw","edge_gj","rows_dw")})
PY
cp profiles/r02_parity_report.txt $O/ 2>/dev/null
