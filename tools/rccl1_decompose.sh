#!/bin/bash
# Dev tool (GPU box): what the one-rank RCCL leg of tools/collect_profiles.sh adds to the layer step, taken apart on ONE box:
#   a  plain `python bench.py`                                  (no process group, no averager)
#   b  the same under torch.distributed.run                     (launcher + its environment: OMP_NUM_THREADS=1 ...)
#   c  b + CGAT_DIST_FORCE=1                                    (process group over RCCL + GradientAverager: the collection's leg)
#   d  a + the launcher's environment variables, CGAT_DIST_FORCE=1, no launcher process
# two rounds, to see the run-to-run spread
cd ${GRAFT_REPO_ROOT:-$(pwd)}
Q="--steps 20 --warmup 5 --no-cpu-baseline --no-exclusive-pass --no-extra-legs"
ms() { python3 -c "import json,sys; d=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$1', round(d['ms_per_step'],3), (d.get('allreduce') or {}).get('launched_from_hooks_per_step'))"; }
for r in 1 2; do
python3 bench.py $Q 2>/dev/null | ms a
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2955$r bench.py --gpus 1 $Q 2>/dev/null | ms b
CGAT_DIST_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 2956$r bench.py --gpus 1 $Q 2>/dev/null | ms c
CGAT_DIST_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2957$r python3 bench.py --gpus 1 $Q 2>/dev/null | ms d
CGAT_DIST_FORCE=1 RANK=0 WORLD_SIZE=1 LOCAL_RANK=0 MASTER_ADDR=127.0.0.1 MASTER_PORT=2958$r OMP_NUM_THREADS=1 python3 bench.py --gpus 1 $Q 2>/dev/null | ms d_omp1
done
