#!/bin/bash
# kernel statistics of the 64-crystal stack step (after the skinny split-K)
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3p_prof64 -- python3 $R/bench.py --workload stack --graphs 64 --steps 20 --warmup 5 --no-cpu-baseline > $R/gpurun_out/r3p_prof64.log 2>&1
cd $R
f=$(find gpurun_out/r3p_prof64 -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r3p_stack64_kernel_stats.csv
rm -rf gpurun_out/r3p_prof64
head -40 gpurun_out/r3p_stack64_kernel_stats.csv | cut -c1-70,100-
