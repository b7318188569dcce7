#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_golden.py tests/test_capture.py tests/test_api_holes.py -m gpu -q -x -k "vector or lightning or capture or dropout or widths" > gpurun_out/r3af_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3af_tests.log; tail -3 gpurun_out/r3af_tests.log
for k in 1 4 8; do
CGAT_HEAD_STREAMS=$k python bench.py --workload lightning --graphs 64 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('streams=$k lightning64 eager', round(d['ms_per_step'],3), 'graph', d['launch_bound']['hipgraph'])"
done
