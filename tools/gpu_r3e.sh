#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3e}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_kernels.py tests/test_capture.py tests/test_api_holes.py -m gpu -q -p no:cacheprovider > $O/tests_k.log 2>&1
tail -12 $O/tests_k.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
python bench.py --steps 10 --warmup 3 $Q > $O/bench_split.json 2> $O/bench_split.err
CGAT_ROWS_DW_F32=1 python bench.py --steps 10 --warmup 3 $Q > $O/bench_f32dw.json 2> $O/bench_f32dw.err
python bench.py --workload lightning --steps 3 --warmup 1 > $O/lightning.json 2> $O/lightning.err
python bench.py --workload stack --graphs 64 --steps 20 --warmup 5 $Q > $O/stack64.json 2> $O/stack64.err
python bench.py --workload stack --steps 5 --warmup 2 $Q > $O/stack.json 2> $O/stack.err
for f in bench_split bench_f32dw lightning stack64 stack; do echo == $f; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("launch_bound",{}).get("hipgraph") if d.get("launch_bound") else "", {k:v["ms_per_step"] for k,v in d.get("kernel_ms_per_step",{}).items() if k in ("rows_dw","gemm_f32","mlp_chain","linear128")})
except Exception as ex: print("ERR", ex); print(open("$O/$f.err").read()[-2500:])
PY
done
python -m pytest tests -m gpu -q -p no:cacheprovider -x --deselect tests/test_hip_kernels.py --deselect tests/test_capture.py --deselect tests/test_api_holes.py > $O/tests_rest.log 2>&1
tail -8 $O/tests_rest.log
