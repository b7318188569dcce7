#!/bin/bash
# round 3: the new tests (API holes, hipGraph capture, hub segments, collate/trainer), the launch-bound benches at the
# reference's shipped batch size, the Lightning-default network, the train bench again
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3c}
mkdir -p $O
cd $R
python -m pytest tests/test_api_holes.py tests/test_capture.py tests/test_collate_gpu.py tests/test_rccl_one_rank.py tests/test_chunked.py -m gpu -q -p no:cacheprovider > $O/tests_new.log 2>&1
tail -30 $O/tests_new.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
python bench.py --workload stack --graphs 64 --steps 20 --warmup 5 $Q > $O/stack64.json 2> $O/stack64.err
python bench.py --workload layer --graphs 64 --steps 20 --warmup 5 $Q > $O/layer64.json 2> $O/layer64.err
python bench.py --workload train --graphs 64 --steps 20 --warmup 5 > $O/train64.json 2> $O/train64.err
python bench.py --workload lightning --steps 3 --warmup 1 > $O/lightning.json 2> $O/lightning.err
python bench.py --workload train --steps 8 --warmup 3 > $O/train_plain.json 2> $O/train_plain.err
python bench.py --workload stack --steps 5 --warmup 2 $Q --hipgraph > $O/stack.json 2> $O/stack.err
for f in stack64 layer64 train64 lightning train_plain stack; do echo == $f; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("step_ms_gpu_events"), d.get("launch_bound"), d.get("roofline") if "$f"=="lightning" else "", d.get("kernel_ms_per_step") if "$f"=="lightning" else "", d.get("peak_memory_GB"))
except Exception as ex: print("ERR", ex); print(open("$O/$f.err").read()[-2500:])
PY
done
cat gpurun_out/r03_hub_timing.json
