cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2r}; mkdir -p $O
for i in 1 2; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench$i.json 2> $O/bench$i.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench$i.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
PY
done
CGAT_OVERLAP_WGRAD=0 python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/bench_serial.json 2> $O/bench_serial.err
python - <<PY
import json
d=json.loads(open("$O/bench_serial.json").read().strip().splitlines()[-1])
print("serial", round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
PY
