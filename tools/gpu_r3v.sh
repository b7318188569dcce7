#!/bin/bash
# the split (six-pass bf16) form of the generic engine: tests, then A/B on the workloads that use it
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -m gpu -q -x > gpurun_out/r3v_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3v_tests.log
tail -3 gpurun_out/r3v_tests.log
python -m pytest tests/test_hip_golden.py tests/test_api_holes.py tests/test_chunked.py -m gpu -q -x > gpurun_out/r3v_golden.log 2>&1; echo "golden rc=$?" >> gpurun_out/r3v_golden.log
tail -3 gpurun_out/r3v_golden.log
for v in 0 1; do
  echo "== CGAT_GEMM_SPLIT=$v"
  CGAT_GEMM_SPLIT=$v timeout 600 python tools/width_sweep.py 64 96 256 2>&1 | grep -v amdgpu.ids
  CGAT_GEMM_SPLIT=$v python bench.py --workload stack --steps 5 --warmup 2 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('stack', d['ms_per_step'])"
  CGAT_GEMM_SPLIT=$v python bench.py --workload stack --graphs 64 --steps 30 --warmup 5 --hipgraph --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('stack64', d['ms_per_step'], d['launch_bound']['hipgraph']['ms_per_step'])"
  CGAT_GEMM_SPLIT=$v python bench.py --workload lightning --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('lightning', d['ms_per_step'])"
done
