#!/bin/bash
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3g}
mkdir -p $O
cd $R
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
for sp in 0 8 4 6 2; do
  CGAT_ASPLIT=$sp python bench.py --steps 10 --warmup 3 $Q > $O/bench_sp$sp.json 2> $O/bench_sp$sp.err
  python - <<PY
import json
d=json.loads(open("$O/bench_sp$sp.json").read().strip().splitlines()[-1]); k=d["kernel_ms_per_step"]; print("asplit $sp:", round(d["ms_per_step"],3), {t:k[t]["ms_per_step"] for t in ("bilinear_rows","bilinear_dual","mlp_chain","linear128")})
PY
done
