cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
O=gpurun_out/prof_rccl1; rm -rf $O; mkdir -p $O
export CGAT_DIST_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=29561
rocprofv3 --kernel-trace --stats --output-format csv -d $O -- python3 bench.py --gpus 1 --steps 10 --warmup 3 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $O/log.txt 2>&1
f=$(ls $O/*/*_kernel_stats.csv | head -1)
python3 - "$f" <<'P'
import csv,sys,re
rows=list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n=r['Name']
    if any(k in n.lower() for k in ('nccl','rccl','fill','elementwise','copy','reduce_kernel','add','mul')):
        print(re.sub(r'\(.*','',n)[:70], r['Calls'], round(int(r['TotalDurationNs'])/13/1000,1),'us/step', round(float(r['AverageNs'])/1000,1))
P
tail -2 $O/log.txt | cut -c1-300
