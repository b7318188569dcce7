#!/bin/bash
# round 3, first GPU call: the whole -m gpu suite (all failures listed), then the layer / train benches plain and with
# the gradient all-reduce forced on over a one-rank RCCL communicator (CGAT_DIST_FORCE=1)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r3a}
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests.log 2>&1
tail -40 $O/tests.log
Q="--no-cpu-baseline --no-extra-legs --no-exclusive-pass"
python bench.py --steps 10 --warmup 3 $Q > $O/bench_plain.json 2> $O/bench_plain.err
CGAT_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 10 --warmup 3 $Q > $O/bench_rccl1.json 2> $O/bench_rccl1.err
python bench.py --workload train --steps 5 --warmup 2 > $O/train_plain.json 2> $O/train_plain.err
CGAT_DIST_FORCE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --workload train --steps 5 --warmup 2 > $O/train_rccl1.json 2> $O/train_rccl1.err
for f in bench_plain bench_rccl1 train_plain train_rccl1; do echo $f; python - <<PY
import json
try:
    d=json.loads(open("$O/$f.json").read().strip().splitlines()[-1]); print(d["ms_per_step"], d.get("allreduce"))
except Exception as ex: print("ERR", ex); print(open("$O/$f.err").read()[-1500:])
PY
done
