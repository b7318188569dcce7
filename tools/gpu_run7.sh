cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2l}; mkdir -p $O
for m in bf16x6 f32; do
  CGAT_BILINEAR_MODE=$m timeout 2400 python -m pytest tests -q -m gpu > $O/t_$m.log 2>&1; echo "tests $m rc=$?"
  tail -n 8 $O/t_$m.log | cut -c1-300
done
