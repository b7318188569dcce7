#!/bin/bash
# Runs on the GPU box (gpurun): the bench line, the rocprofv3 kernel statistics of the same command, the PMC passes for
# HBM traffic (FETCH_SIZE / WRITE_SIZE in separate passes), the matrix-core counters (SQ_VALU_MFMA_BUSY_CYCLES with
# GRBM_GUI_ACTIVE for the clock) and the other workloads.  Outputs under gpurun_out/<dir>/; tools/pmc_summary.py,
# tools/counter_summary.py and a copy into profiles/ happen on the authoring side (tools/publish_profiles.sh).
# Every rocprofv3 command has the program directly after `--`.
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-r06final}
mkdir -p $O
python3 $R/tools/tree_id.py > $O/tree_id.txt
cd /tmp; export TMPDIR=/tmp
python3 $R/bench.py > $O/bench.json 2> $O/bench.err
Q="--no-cpu-baseline --no-exclusive-pass --no-extra-legs"
# (the bench command's own step counts: with 5 + 2 steps the two cold warm-up steps were 2 of the 7 in every average, and the
# averages sat 3-5 % above the timed region's HIP-event durations; the line printed by THIS run is kept beside its statistics)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -- python3 $R/bench.py --steps 20 --warmup 5 $Q > $O/stats.log 2>&1
grep '^{' $O/stats.log | tail -1 > $O/stats_bench.json
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python3 $R/bench.py --steps 1 --warmup 1 $Q > $O/pmc_$c.log 2>&1
done
CGAT_OVERLAP_WGRAD=0 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/pmc_mfma -- python3 $R/bench.py --steps 2 --warmup 1 $Q > $O/pmc_mfma.log 2>&1
python3 $R/bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass > $O/bench_stress.json 2> $O/bench_stress.err
python3 $R/bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass --edge-storage bf16 > $O/bench_stress_bf16.json 2> $O/bench_stress_bf16.err
python3 $R/bench.py --workload stress --steps 2 --warmup 1 --no-exclusive-pass --edge-storage bf16-mma > $O/bench_stress_bf16mma.json 2> $O/bench_stress_bf16mma.err
# BASELINE configs[4]'s "rocprof roofline run": kernel statistics and HBM counters of one stress step (fp32 and bf16 storage)
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress_stats -- python3 $R/bench.py --workload stress --steps 1 --warmup 1 --no-exclusive-pass > $O/stress_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress_bf16_stats -- python3 $R/bench.py --workload stress --steps 1 --warmup 1 --no-exclusive-pass --edge-storage bf16 > $O/stress_bf16_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stress_bf16mma_stats -- python3 $R/bench.py --workload stress --steps 1 --warmup 1 --no-exclusive-pass --edge-storage bf16-mma > $O/stress_bf16mma_stats.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/stress_pmc_$c -- python3 $R/bench.py --workload stress --steps 1 --warmup 0 --no-exclusive-pass > $O/stress_pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/stress_bf16_pmc_$c -- python3 $R/bench.py --workload stress --steps 1 --warmup 0 --no-exclusive-pass --edge-storage bf16 > $O/stress_bf16_pmc_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/stress_bf16mma_pmc_$c -- python3 $R/bench.py --workload stress --steps 1 --warmup 0 --no-exclusive-pass --edge-storage bf16-mma > $O/stress_bf16mma_pmc_$c.log 2>&1
done
python3 $R/bench.py --workload train --steps 8 --warmup 3 > $O/bench_train.json 2> $O/bench_train.err
python3 $R/bench.py --workload stack --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass --hipgraph > $O/bench_stack.json 2> $O/bench_stack.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stack_stats -- python3 $R/bench.py --workload stack --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive-pass > $O/stack_stats.log 2>&1
python3 $R/bench.py --workload edge_hyper --steps 3 --warmup 1 > $O/bench_edge_hyper.json 2> $O/bench_edge_hyper.err
# the harness' shipped default network, and its shipped batch size (64 crystals per GPU): launch-bound regime, hipGraph replay
python3 $R/bench.py --workload lightning --steps 3 --warmup 1 > $O/bench_lightning.json 2> $O/bench_lightning.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lightning_stats -- python3 $R/bench.py --workload lightning --steps 2 --warmup 1 > $O/lightning_stats.log 2>&1
python3 $R/bench.py --workload lightning --graphs 64 --steps 20 --warmup 5 > $O/bench_lightning64.json 2> $O/bench_lightning64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/lightning64_stats -- python3 $R/bench.py --workload lightning --graphs 64 --steps 10 --warmup 2 > $O/lightning64_stats.log 2>&1
python3 $R/bench.py --workload stack --graphs 64 --steps 20 --warmup 5 $Q > $O/bench_stack64.json 2> $O/bench_stack64.err
python3 $R/bench.py --workload layer --graphs 64 --steps 20 --warmup 5 $Q > $O/bench_layer64.json 2> $O/bench_layer64.err
python3 $R/bench.py --workload train --graphs 64 --steps 20 --warmup 5 > $O/bench_train64.json 2> $O/bench_train64.err
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stack64_stats -- python3 $R/bench.py --workload stack --graphs 64 --steps 10 --warmup 2 $Q > $O/stack64_stats.log 2>&1
# the gradient all-reduce path over a ONE-RANK RCCL communicator (CGAT_DIST_FORCE=1): what every rank of an N > 1 run does
cd $R
CGAT_DIST_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29541 bench.py --gpus 1 --steps 20 --warmup 5 $Q > $O/bench_rccl1.json 2> $O/bench_rccl1.err
python3 bench.py --steps 20 --warmup 5 $Q > $O/bench_norccl.json 2> $O/bench_norccl.err
CGAT_DIST_FORCE=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29542 bench.py --gpus 1 --workload train --steps 8 --warmup 3 > $O/bench_train_rccl1.json 2> $O/bench_train_rccl1.err
# the N > 1 code path on the one GPU this box has (both ranks on cuda:0, gloo instead of RCCL: functional evidence only)
CGAT_DIST_BACKEND=gloo CGAT_DIST_SHARE_GPU=1 python3 -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 3 --warmup 1 --no-cpu-baseline --no-extra-legs --no-exclusive-pass > $O/bench_2ranks_one_gpu.json 2> $O/bench_2ranks_one_gpu.err
python3 tools/abi_call_census.py 64 > $O/census64.txt 2>&1
tail -c 400 $O/bench.json
