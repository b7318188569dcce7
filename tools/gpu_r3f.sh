#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3f}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider -x > $O/tests_k.log 2>&1
tail -6 $O/tests_k.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
python bench.py --steps 10 --warmup 3 $Q > $O/bench.json 2> $O/bench.err
python - <<PY
import json
d=json.loads(open("$O/bench.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], {k:v["ms_per_step"] for k,v in d["kernel_ms_per_step"].items()})
PY
python -m pytest tests/test_hip_golden.py tests/test_api_holes.py -m gpu -q -p no:cacheprovider -x > $O/tests_g.log 2>&1
tail -6 $O/tests_g.log
cat gpurun_out/r03_hub_timing.json
