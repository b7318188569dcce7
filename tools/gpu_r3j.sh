#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3j}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_golden.py tests/test_api_holes.py tests/test_hip_kernels.py -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1
tail -12 $O/tests.log
python bench.py --workload lightning --steps 3 --warmup 1 > $O/lightning.json 2> $O/lightning.err
python bench.py --workload edge_hyper --steps 3 --warmup 1 > $O/edge_hyper.json 2> $O/edge_hyper.err
for f in lightning edge_hyper; do python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["ms_per_step"],2), {k:v["ms_per_step"] for k,v in d.get("kernel_ms_per_step",{}).items()})
except Exception as ex: print("ERR", ex); print(open("$O/$f.err").read()[-2500:])
PY
done
