#!/bin/bash
# which kernels carry the layer step at widths 64 / 96 / 256
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for C in 64 96; do
rocprofv3 --kernel-trace --stats --output-format csv -d $R/gpurun_out/r3r_w$C -- python3 $R/tools/width_sweep.py $C > $R/gpurun_out/r3r_w$C.log 2>&1
f=$(find $R/gpurun_out/r3r_w$C -name '*kernel_stats.csv' | head -1); cp "$f" $R/gpurun_out/r3r_w${C}_kernel_stats.csv
rm -rf $R/gpurun_out/r3r_w$C
echo "== C=$C"; head -14 $R/gpurun_out/r3r_w${C}_kernel_stats.csv | cut -c1-90
done
