#!/bin/bash
# Dev tool (GPU box): per-kernel table (grid, LDS, registers, calls, median) of the 64-crystal workloads' kernel traces
cd /tmp; export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-/root/repo}
for w in lightning stack; do
O=$R/gpurun_out/prof64_$w; rm -rf $O; mkdir -p $O
rocprofv3 --kernel-trace --output-format csv -d $O -- python3 $R/bench.py --workload $w --graphs 64 --steps 10 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $O/log.txt 2>&1
python3 - $O $w <<'P'
import csv,glob,collections,sys
f=glob.glob(sys.argv[1]+'/*/*_kernel_trace.csv')[0]
agg=collections.defaultdict(list)
for r in csv.DictReader(open(f)):
    n=r['Kernel_Name'].split('(')[0][:46]
    key=(n,int(r['Grid_Size_X'])//int(r['Workgroup_Size_X']),r['Grid_Size_Y'],r['Workgroup_Size_X'],r['LDS_Block_Size'],r['VGPR_Count'])
    agg[key].append((int(r['End_Timestamp'])-int(r['Start_Timestamp']))/1e3)
tot=sum(sum(v) for v in agg.values())
print(sys.argv[2],'total kernel ms',round(tot/1e3,1))
for k,v in sorted(agg.items(), key=lambda kv:-sum(kv[1]))[:34]:
    v.sort()
    print(' ',k, len(v), 'tot',round(sum(v)/1e3,1),'med',round(v[len(v)//2],1))
P
done
