#!/bin/bash
for v in "CGAT_GEMM_SPLIT=0" "CGAT_GEMM_SPLIT_PASSES=6" "CGAT_GEMM_SPLIT_PASSES=8"; do
  echo "== $v"
  env $v python -m pytest tests/test_hip_golden.py -m gpu -q -x -k "golden_tiny or golden_base" 2>&1 | tail -3 | cut -c1-250
  env $v python tools/gemm_engine_probe.py 2>&1 | grep "akm=0 bkm=0\|signed"
done
