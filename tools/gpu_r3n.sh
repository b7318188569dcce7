#!/bin/bash
# A/B: dT operand preparation per layer on the side stream (early) against all at the end
mkdir -p gpurun_out
python -m pytest tests/test_hip_golden.py tests/test_hip_kernels.py tests/test_capture.py -m gpu -q -x > gpurun_out/r3n_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3n_tests.log
tail -3 gpurun_out/r3n_tests.log
for rep in 1 2 3; do
  for v in 0 1; do
    CGAT_SIDE_EARLY_PREP=$v python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-extra-legs > gpurun_out/r3n_bench_$v.log 2>&1
    python - <<PY
import json
l=open("gpurun_out/r3n_bench_$v.log").read().strip().splitlines()[-1]
try:
    d=json.loads(l); print("early=$v rep=$rep ms=%.3f" % d["ms_per_step"])
except Exception as ex:
    print("early=$v parse fail", l[-300:])
PY
  done
done
