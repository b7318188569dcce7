#!/bin/bash
# HEAD validation: full gpu suite (default mode) + bench line
mkdir -p gpurun_out
python -m pytest tests -m gpu -q -x > gpurun_out/r3m_tests.log 2>&1; echo "tests rc=$?" >> gpurun_out/r3m_tests.log
tail -5 gpurun_out/r3m_tests.log
python bench.py --steps 30 --warmup 5 > gpurun_out/r3m_bench.log 2>&1; tail -1 gpurun_out/r3m_bench.log | cut -c1-600
