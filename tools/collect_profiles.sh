#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel statistics of the same command and the two PMC
# passes for HBM traffic.  Outputs under gpurun_out/final/; tools/pmc_summary.py + a copy into profiles/ happen on
# the authoring side.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/final
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass > $O/stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-exclusive-pass > $O/pmc_$c.log 2>&1
done
rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE --output-format csv -d $O/pmc_GRBM_GUI_ACTIVE -- python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-pass > $O/pmc_clk.log 2>&1
tail -c 600 $O/bench.json
