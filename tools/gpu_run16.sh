cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2aa}; mkdir -p $O
timeout 900 python -m pytest tests/test_chunked.py -q -m gpu -x > $O/t_c.log 2>&1; echo "chunked tests rc=$?"; tail -n 6 $O/t_c.log | cut -c1-400
CGAT_BILINEAR_MODE=bf16x6 timeout 900 python -m pytest tests/test_chunked.py -q -m gpu -x -k "rebuilt" > $O/t_c6.log 2>&1; echo "rebuilt bf16x6 rc=$?"; tail -n 3 $O/t_c6.log | cut -c1-300
bash tools/collect_profiles.sh ${1:-r2aa}
bash tools/gpu_prof_counters.sh ${1:-r2aa}/counters > /dev/null 2>&1
