cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2i}; mkdir -p $O
timeout 600 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "chain" > $O/t_chain.log 2>&1; echo "chain rc=$?"; tail -n 15 $O/t_chain.log | cut -c1-300
timeout 2400 python -m pytest tests -q -m gpu -k "chain or golden or stack or hnet or training or simple" > $O/t_all.log 2>&1; echo "tests rc=$?"
tail -n 8 $O/t_all.log | cut -c1-300
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench_overlap.json 2> $O/bench_overlap.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_overlap.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
PY
python bench.py --workload stack --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass > $O/bench_stack.json 2> $O/bench_stack.err; echo "stack rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench_stack.json").read().strip().splitlines()[-1])
print("stack", round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
PY
