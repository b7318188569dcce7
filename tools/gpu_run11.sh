cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2s}; mkdir -p $O
timeout 1500 python -m pytest tests/test_hip_golden.py -q -m gpu -x > $O/t_g.log 2>&1; echo "golden tests rc=$?"; tail -n 15 $O/t_g.log | cut -c1-600
timeout 1500 python -m pytest tests/test_chunked.py tests/test_hip_kernels.py -q -m gpu -x > $O/t_c.log 2>&1; echo "chunked+kernel tests rc=$?"; tail -n 15 $O/t_c.log | cut -c1-600
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
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stats -- python3 $GRAFT_REPO_ROOT/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $GRAFT_REPO_ROOT/$O/stats.log 2>&1
