#!/bin/bash
# Per-kernel SQ / GRBM counters of the benchmark step (rocprofv3 --pmc passes; program directly after `--`).
# Usage on the GPU box: bash tools/gpu_prof_counters.sh <outdir-under-gpurun_out>
R=${GRAFT_REPO_ROOT:-$(pwd)}
O=$R/gpurun_out/${1:-counters}
mkdir -p $O
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L > $O/counter_list.txt 2>&1
B="python3 $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline --no-exclusive-pass --no-extra-legs"
export CGAT_OVERLAP_WGRAD=${CGAT_OVERLAP_WGRAD:-0}
rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES --output-format csv -d $O/mfma -- $B > $O/mfma.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_WAIT_INST_LDS SQ_INSTS_VALU SQ_INSTS_LDS --output-format csv -d $O/waits -- $B > $O/waits.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_VMEM SQ_INSTS_VALU_MFMA_MOPS_F16 --output-format csv -d $O/lds -- $B > $O/lds.log 2>&1
ls -R $O | head -40
