#!/bin/bash
# Dev tool: timing-only ablations of the per-edge forward kernel (CGAT_EZX_ABL bits: 1 no Z stores, 2 no Pi loads,
# 4 no logits; results are wrong by construction).  Prints the kernel's time per launch from bench.py's HIP events.
for g in 0 1 2 3 7; do
  CGAT_EZX_ABL=$g python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-exclusive-pass 2>&1 | tail -1 > /tmp/ezx_$g.json
  python -c "import json; d=json.load(open('/tmp/ezx_$g.json')); print('ABL', $g, d['kernel_ms_per_step']['edge_z']['ms_per_step'], 'ms')"
done
