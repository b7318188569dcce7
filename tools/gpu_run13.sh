cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2v}; mkdir -p $O
timeout 2700 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "tests rc=$?"; tail -n 8 $O/t_all.log | cut -c1-400
python bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"; tail -c 1500 $O/bench.json
for ce in 4194304 8388608 16777216; do
CGAT_MAX_EDGES_PER_PASS=$ce python bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass > $O/stress_$ce.json 2> $O/stress_$ce.err
python - <<PY
import json
d=json.loads(open("$O/stress_$ce.json").read().strip().splitlines()[-1])
print("stress chunk $ce", round(d["ms_per_step"],1), round(d["value"]), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["ms_per_step"])
PY
done
CGAT_MAX_EDGES_PER_PASS=8388608 python bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass --edge-storage bf16 > $O/stress_bf16.json 2> $O/stress_bf16.err
python - <<PY
import json
d=json.loads(open("$O/stress_bf16.json").read().strip().splitlines()[-1])
print("stress bf16 8M", round(d["ms_per_step"],1), round(d["value"]))
PY
