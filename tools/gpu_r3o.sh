#!/bin/bash
# skinny split-K of the generic engine + the embedding gradient with loads in flight: tests, then the small-batch and stack legs
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py tests/test_hip_golden.py tests/test_capture.py -m gpu -q -x > gpurun_out/r3o_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3o_tests.log
tail -3 gpurun_out/r3o_tests.log
python bench.py --workload stack --graphs 64 --steps 30 --warmup 5 --hipgraph --no-cpu-baseline > gpurun_out/r3o_stack64.log 2>&1; tail -1 gpurun_out/r3o_stack64.log | cut -c1-900
python bench.py --workload stack --steps 5 --warmup 2 --no-cpu-baseline > gpurun_out/r3o_stack.log 2>&1; tail -1 gpurun_out/r3o_stack.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3o_prof64 -- python3 $GRAFT_REPO_ROOT/bench.py --workload stack --graphs 64 --steps 20 --warmup 5 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/r3o_prof64.log 2>&1
cd $GRAFT_REPO_ROOT
f=$(find gpurun_out/r3o_prof64 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r3o_stack64_kernel_stats.csv; head -30 gpurun_out/r3o_stack64_kernel_stats.csv | cut -c1-60,100-
rm -rf gpurun_out/r3o_prof64
