# profile collection + counter passes without the test suites (gpurun -- 'bash tools/gpu_collect_only.sh <outdir>')
cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_hip_golden.py -q -m gpu -x -k "determinism or side_stream or nodes_layer" > gpurun_out/${1:-final}_t.log 2>&1; echo "tests rc=$?"; tail -n 3 gpurun_out/${1:-final}_t.log | cut -c1-200
bash tools/collect_profiles.sh ${1:-final}
bash tools/gpu_prof_counters.sh ${1:-final}/counters > /dev/null 2>&1
