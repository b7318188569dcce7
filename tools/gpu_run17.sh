cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2ab}; mkdir -p $O
timeout 900 python -m pytest tests/test_chunked.py -q -m gpu -x > $O/t_c.log 2>&1; echo "chunked tests rc=$?"; tail -n 6 $O/t_c.log | cut -c1-400
CGAT_BILINEAR_MODE=bf16x6 timeout 900 python -m pytest tests/test_chunked.py -q -m gpu -x -k "rebuilt" > $O/t_c6.log 2>&1; echo "rebuilt bf16x6 rc=$?"; tail -n 3 $O/t_c6.log | cut -c1-300
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/b$i.json 2> $O/b$i.err
python - <<PY
import json
d=json.loads(open("$O/b$i.json").read().strip().splitlines()[-1])
k=d["kernel_ms_per_step"]
print(round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("edge_seg_bwd","edge_ge","edge_gw","edge_gj","rows_dw")})
PY
done
