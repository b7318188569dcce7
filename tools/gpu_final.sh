#!/bin/bash
# end of round: the golden / kernel suites in the four modes, the width sweep, the whole collection
mkdir -p gpurun_out
for m in f16x3c bf16x6 f16x3 f32; do
  CGAT_BILINEAR_MODE=$m python -m pytest tests/test_hip_golden.py tests/test_hip_kernels.py tests/test_capture.py -m gpu -q -x -p no:cacheprovider > gpurun_out/final_tests_$m.log 2>&1; echo "$m rc=$?" >> gpurun_out/final_tests_$m.log; tail -2 gpurun_out/final_tests_$m.log
done
timeout 600 python tools/width_sweep.py 128 64 96 256 2>&1 | grep -v amdgpu
bash tools/collect_profiles.sh r04final | tail -c 200
