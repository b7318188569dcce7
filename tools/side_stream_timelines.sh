cd /tmp; export TMPDIR=/tmp
O=$GRAFT_REPO_ROOT/gpurun_out/${1:-r2z}; mkdir -p $O
CGAT_SIDE_WGRAD_WGS=256 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats256 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $O/stats256.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats128 -- python3 $GRAFT_REPO_ROOT/bench.py --steps 4 --warmup 2 --no-cpu-baseline --no-exclusive-pass --no-extra-legs > $O/stats128.log 2>&1
ls $O
