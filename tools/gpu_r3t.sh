#!/bin/bash
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "bilinear or gemm or linear" > gpurun_out/r3t_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3t_tests.log
tail -3 gpurun_out/r3t_tests.log
timeout 600 python tools/width_sweep.py 128 64 96 256 2>&1 | grep -v amdgpu.ids
python bench.py --workload stack --graphs 64 --steps 30 --warmup 5 --hipgraph --no-cpu-baseline > gpurun_out/r3t_stack64.log 2>&1; tail -1 gpurun_out/r3t_stack64.log | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('stack64', d['ms_per_step'], d['launch_bound']['hipgraph']['ms_per_step'])"
