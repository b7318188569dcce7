#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3d}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1
tail -15 $O/tests.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
python bench.py --workload stack --graphs 64 --steps 20 --warmup 5 $Q > $O/stack64.json 2> $O/stack64.err
python bench.py --workload layer --graphs 64 --steps 20 --warmup 5 $Q > $O/layer64.json 2> $O/layer64.err
python bench.py --workload lightning --steps 3 --warmup 1 > $O/lightning.json 2> $O/lightning.err
python bench.py --steps 10 --warmup 3 $Q > $O/bench_plain.json 2> $O/bench_plain.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stack64_stats -- python3 $R/bench.py --workload stack --graphs 64 --steps 10 --warmup 2 $Q > $O/stack64_stats.log 2>&1
cd $R
for f in stack64 layer64 lightning bench_plain; do echo == $f; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launch_bound"), d.get("roofline",{}).get("kernel_tag") if "$f"=="lightning" else "", {k:v["ms_per_step"] for k,v in d.get("kernel_ms_per_step",{}).items()} if "$f"=="lightning" else "")
except Exception as ex: print("ERR", ex); print(open("$O/$f.err").read()[-2500:])
PY
done
cat gpurun_out/r03_hub_timing.json
