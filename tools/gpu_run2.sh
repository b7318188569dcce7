cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2e}; mkdir -p $O; rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests -q -m gpu > $O/t_all.log 2>&1; echo "tests rc=$?"
tail -n 25 $O/t_all.log | cut -c1-400
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
python bench.py --workload train --steps 5 --warmup 2 > $O/bench_train.json 2> $O/bench_train.err; echo "bench train rc=$?"
tail -c 600 $O/bench_train.err; tail -c 900 $O/bench_train.json
python tools/vector_attention_time.py > $O/vector_time.txt 2>&1; cat $O/vector_time.txt | tail -3
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/vec_stats -- python3 $GRAFT_REPO_ROOT/tools/vector_attention_time.py > $GRAFT_REPO_ROOT/$O/vec_stats.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/stack_stats -- python3 $GRAFT_REPO_ROOT/bench.py --workload stack --steps 3 --warmup 1 --no-cpu-baseline --no-exclusive-pass > $GRAFT_REPO_ROOT/$O/stack_stats.log 2>&1
ls $GRAFT_REPO_ROOT/$O
