cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2g}; mkdir -p $O; rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "tests rc=$?"
tail -n 12 $O/t_all.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_overlap.json 2> $O/bench_overlap.err; echo "bench rc=$?"
CGAT_OVERLAP_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_serial.json 2> $O/bench_serial.err; echo "bench2 rc=$?"
python - <<PY
import json
for f in ("overlap","serial"):
    try:
        d=json.loads(open("$O/bench_%s.json" % f).read().strip().splitlines()[-1])
        print(f, round(d["ms_per_step"],3), {k:v["ms_per_step"] for k,v in d["kernel_ms_per_step"].items()})
    except Exception as e: print(f, "ERR", e)
PY
