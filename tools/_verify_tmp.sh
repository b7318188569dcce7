cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r2au
timeout 600 python tools/vector_attention_time.py > gpurun_out/r2au/vec.log 2>&1; echo "rc=$?"; tail -6 gpurun_out/r2au/vec.log | cut -c1-300
