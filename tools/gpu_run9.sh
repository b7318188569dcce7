cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2q}; mkdir -p $O
timeout 900 python -m pytest tests/test_hip_kernels.py -q -m gpu -x -k "dense_wgrad or hnet or linear" > $O/t_k.log 2>&1; echo "kernel tests rc=$?"; tail -n 12 $O/t_k.log | cut -c1-400
timeout 1500 python -m pytest tests/test_hip_golden.py tests/test_collate_gpu.py -q -m gpu -x > $O/t_g.log 2>&1; echo "golden tests rc=$?"; tail -n 12 $O/t_g.log | cut -c1-400
python bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-extra-legs > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1])
print(round(d["ms_per_step"],3), {k:(v["launches_per_step"], v["ms_per_step"]) for k,v in d["kernel_ms_per_step"].items()})
PY
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
