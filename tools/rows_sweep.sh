for w in lightning stack; do
for c in 2048 1024; do for p in 2048 1024; do
CGAT_ROWPROG_MAX_ROWS=$c CGAT_ROWPROG_PY_MAX_ROWS=$p python bench.py --workload $w --graphs 64 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]);h=d['launch_bound']['hipgraph']
print('$w c=$c py=$p eager',round(d['ms_per_step'],2),'graph',h['ms_per_step'],h['library_kernel_launches_in_graph'])"
done; done; done
