#!/bin/bash
# round 4, first measurement: the layer step in the 24-bit-operand mode (bf16x6) -- bench line + rocprofv3 kernel stats
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r4a}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
export CGAT_BILINEAR_MODE=${MODE:-bf16x6}
Q="--no-cpu-baseline --no-extra-legs"
python3 $R/bench.py $Q > $O/bench.json 2> $O/bench.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 5 --warmup 2 $Q --no-exclusive-pass > $O/stats.log 2>&1
CGAT_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -- python3 $R/bench.py --steps 5 --warmup 2 $Q --no-exclusive-pass > $O/stats_serial.log 2>&1
tail -c 600 $O/bench.json
