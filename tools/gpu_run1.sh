cd $GRAFT_REPO_ROOT
O=gpurun_out/r2b; mkdir -p $O
timeout 1500 python -m pytest tests -x -q -m gpu > $O/t_all.log 2>&1; echo "all rc=$?"
python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_overlap.json 2> $O/bench_overlap.err; echo "bench rc=$?"
CGAT_OVERLAP_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > $O/bench_serial.json 2> $O/bench_serial.err; echo "bench2 rc=$?"
tail -n 5 $O/t_all.log
python - <<'PY'
import json
for f in ("overlap","serial"):
    try:
        d=json.loads(open(f"gpurun_out/r2b/bench_{f}.json").read().strip().splitlines()[-1])
        print(f, d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernel_ms_per_step"].items()})
        print("  wgrad", d["roofline"].get("other_contraction_kernels",{}).get("bilinear_wgrad"), d["roofline"]["kernel"], d["roofline"]["frac"])
    except Exception as e: print(f, "ERR", e)
PY
