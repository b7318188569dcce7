#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "heads_linear or linear_backward_dact" > gpurun_out/r3ag_k.log 2>&1; echo "rc=$?" >> gpurun_out/r3ag_k.log; tail -15 gpurun_out/r3ag_k.log
python -m pytest tests/test_hip_golden.py tests/test_capture.py tests/test_api_holes.py -m gpu -q -x -k "vector or lightning or capture or dropout" > gpurun_out/r3ag_tests.log 2>&1; echo "rc=$?" >> gpurun_out/r3ag_tests.log; tail -3 gpurun_out/r3ag_tests.log
python bench.py --workload lightning --graphs 64 --steps 20 --warmup 5 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lightning64 eager', round(d['ms_per_step'],3), 'graph', d['launch_bound']['hipgraph']['ms_per_step'], d['launch_bound']['library_kernel_launches_per_step'])"
python bench.py --workload lightning --steps 3 --warmup 1 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lightning 1M', round(d['ms_per_step'],3))"
