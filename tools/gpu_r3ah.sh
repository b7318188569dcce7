#!/bin/bash
# kernel statistics of the harness-default network at 64 crystals
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3ah_prof -- python3 $R/bench.py --workload lightning --graphs 64 --steps 20 --warmup 5 > $R/gpurun_out/r3ah_prof.log 2>&1
cd $R
f=$(find gpurun_out/r3ah_prof -name '*kernel_stats.csv' | head -1); cp "$f" gpurun_out/r3ah_lightning64_kernel_stats.csv
rm -rf gpurun_out/r3ah_prof
