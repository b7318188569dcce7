cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/linres; mkdir -p gpurun_out/linres
rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/linres -- python3 bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > gpurun_out/linres/log.txt 2>&1
python3 - <<'P'
import csv,glob
f=glob.glob('gpurun_out/linres/**/*kernel_stats.csv',recursive=True)[0]
for r in csv.DictReader(open(f)):
    if 'linear128_res' in r['Name'] or 'edge_z_kernel<6, false' in r['Name'] or 'mlp_chain' in r['Name']:
        print(r['Calls'], r['AverageNs'], r['MinNs'], r['MaxNs'], r['Name'][:50])
P
