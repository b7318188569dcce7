cd $GRAFT_REPO_ROOT
O=gpurun_out/${1:-r2f}; mkdir -p $O; rm -f gpurun_out/parity_report.txt
timeout 2400 python -m pytest tests -q -m gpu -k "lightning or dynamic_range or training_step or attention_pool or chunked or plan_hub" > $O/t_sel.log 2>&1; echo "tests rc=$?"
tail -n 30 $O/t_sel.log | cut -c1-300
cp gpurun_out/parity_report.txt $O/ 2>/dev/null
