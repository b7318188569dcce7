for v in 1 0 1 0; do
CGAT_OVERLAP_WGRAD=$v python bench.py --no-cpu-baseline --no-extra-legs --no-exclusive-pass --steps 20 --warmup 5 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('overlap=$v', round(d['ms_per_step'],3))
"
done
