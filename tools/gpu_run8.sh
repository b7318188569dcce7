cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2n}; mkdir -p $O
timeout 900 python -m pytest tests/test_chunked.py -q -m gpu -x -k "bf16" > $O/t_bf16.log 2>&1; echo "bf16 tests rc=$?"; tail -n 12 $O/t_bf16.log | cut -c1-400
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "tests rc=$?"; tail -n 6 $O/t_all.log | cut -c1-300
for st in f32 bf16; do
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs --edge-storage $st > $O/bench_$st.json 2> $O/bench_$st.err; echo "bench $st rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_$st.json").read().strip().splitlines()[-1])
print("$st", round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
print({k:(v["frac"], v["avg_launch_ms"]) for k,v in d["hbm_bound_kernels"].items()})
PY
done
python bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass --edge-storage bf16 > $O/bench_stress_bf16.json 2> $O/bench_stress_bf16.err; echo "stress rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_stress_bf16.json").read().strip().splitlines()[-1])
print("stress bf16", round(d["ms_per_step"],1), round(d["value"]), d["roofline"]["kernel"], d["roofline"]["frac"], d["roofline"]["ms_per_step"])
PY
