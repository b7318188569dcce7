#!/bin/bash
# the -m gpu suite in the four arithmetic modes (per-mode parity reports land in gpurun_out/parity_report_<mode>.txt)
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-modes}
mkdir -p $O
cd $R
rm -f gpurun_out/parity_report_*.txt
python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_f16x3c.log 2>&1; tail -4 $O/tests_f16x3c.log
for m in bf16x6 f16x3 f32; do
  CGAT_BILINEAR_MODE=$m python -m pytest tests -m gpu -q -p no:cacheprovider > $O/tests_$m.log 2>&1; tail -4 $O/tests_$m.log
done
