#!/bin/bash
# generic-width bilinear on the fp32 engine: kernel tests, golden parity (small widths), the width sweep
mkdir -p gpurun_out
python -m pytest tests/test_hip_kernels.py -m gpu -q -x -k "bilinear or gemm" > gpurun_out/r3s_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3s_tests.log
tail -4 gpurun_out/r3s_tests.log
python -m pytest tests/test_hip_golden.py -m gpu -q -x > gpurun_out/r3s_golden.log 2>&1; echo "golden rc=$?" >> gpurun_out/r3s_golden.log
tail -3 gpurun_out/r3s_golden.log
timeout 600 python tools/width_sweep.py 64 96 256 100 2>&1 | grep -v amdgpu.ids
