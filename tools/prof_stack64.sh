#!/bin/bash
# Dev tool (GPU box): rocprofv3 kernel statistics of the 64-crystal 4-layer step (eager, 10 + 2 steps) as a per-step table.
# usage: tools/prof_stack64.sh TAG [workload] [extra bench flags]
TAG=${1:-x}; WL=${2:-stack}; [ $# -ge 1 ] && shift; [ $# -ge 1 ] && shift
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_${WL}64_$TAG
rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --workload $WL --graphs 64 --steps 10 --warmup 2 --no-cpu-baseline "$@" > $O/log.txt 2>&1
f=$(ls $O/*/*_kernel_stats.csv | head -1)
cp $f gpurun_out/r06_${WL}64_${TAG}_kernel_stats.csv
python3 - "$f" <<'P'
import csv,re,sys
rows=list(csv.DictReader(open(sys.argv[1])))
# steps profiled: 12 eager + graph capture (1) + replays: normalise by the calls of a once-per-step kernel
per=None
for r in rows:
    if 'bilinear_wgrad128_f16c_kernel' in r['Name']: per=int(r['Calls'])/(5 if 'lightning' in sys.argv[1] else 4)
if not per: per=12
tot=0;n=0
print(f"steps in profile ~ {per}")
for r in rows:
    name=re.sub(r'\(.*','',r['Name'])[:58]
    c=int(r['Calls'])/per; t=int(r['TotalDurationNs'])/per/1000
    tot+=t;n+=c
    if t>25: print(f"{name:60s} {c:6.1f} {t:8.1f} us  avg {float(r['AverageNs'])/1000:6.1f}")
print("total us/step",round(tot),"launches/step",round(n))
P
