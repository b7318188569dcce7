#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3l}
mkdir -p $O
cd $R
python -m pytest tests/test_hip_golden.py tests/test_chunked.py -m gpu -q -p no:cacheprovider -x > $O/tests.log 2>&1
tail -5 $O/tests.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
run() { name=$1; shift
  env "$@" python bench.py --steps 10 --warmup 3 $Q > $O/bench_$name.json 2> $O/bench_$name.err
  python - <<PY
import json
d=json.loads(open("$O/bench_$name.json").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print("$name:", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("edge_z","edge_seg_bwd","bilinear_wgrad","edge_ge","edge_gw")})
PY
}
run zx2 CGAT_EDGE_ZX2=1
run zx1 CGAT_EDGE_ZX2=0
run zx2b CGAT_EDGE_ZX2=1
CGAT_EDGE_ZX2=1 python bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass > $O/stress_zx2.json 2> $O/stress_zx2.err
CGAT_EDGE_ZX2=0 python bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass > $O/stress_zx1.json 2> $O/stress_zx1.err
for f in stress_zx2 stress_zx1; do python - <<PY
import json
d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print("$f", round(d["ms_per_step"],1), d["kernel_ms_per_step"].get("edge_z"))
PY
done
