#!/bin/bash
# Authoring side: turn one gpurun_out/<dir> written by tools/collect_profiles.sh into the files under profiles/ that
# DESIGN.md §6 and bench.py cite.   bash tools/publish_profiles.sh gpurun_out/r3final r03
set -e
D=${1:?gpurun_out/<dir>}
T=${2:-r06}
R=$(cd "$(dirname "$0")/.." && pwd)
P=$R/profiles
cp $D/bench.json $P/${T}_final_bench.json
cp $(ls $D/stats/*/*_kernel_stats.csv | head -1) $P/${T}_final_kernel_stats.csv
[ -s $D/stats_bench.json ] && cp $D/stats_bench.json $P/${T}_final_kernel_stats_bench.json
for c in FETCH_SIZE WRITE_SIZE; do cp $(ls $D/pmc_$c/*/*_counter_collection.csv | head -1) $P/${T}_final_pmc_${c}_counter_collection.csv; done
python3 $R/tools/pmc_summary.py $D/pmc_FETCH_SIZE $D/pmc_WRITE_SIZE $T
COUNTER_JSON=$P/mfma_counters.json python3 $R/tools/counter_summary.py $D/pmc_mfma > $P/${T}_final_counters_mfma.txt
# the bench line read profiles/mfma_counters.json as it was on the box (the previous collection); attach this one's
python3 - <<PY
import json
b = "$P/${T}_final_bench.json"
d = json.loads(open(b).read().strip().splitlines()[-1])
if d.get("roofline"):
    d["roofline"]["mfma_utilisation_from_counters"] = json.load(open("$P/mfma_counters.json"))
open(b, "w").write(json.dumps(d) + "\n")
PY
for w in stress train stack edge_hyper lightning lightning64 stack64 layer64 train64 rccl1 norccl train_rccl1; do [ -s $D/bench_$w.json ] && cp $D/bench_$w.json $P/${T}_bench_$w.json; done
[ -d $D/lightning_stats ] && cp $(ls $D/lightning_stats/*/*_kernel_stats.csv | head -1) $P/${T}_lightning_kernel_stats.csv
[ -d $D/lightning64_stats ] && cp $(ls $D/lightning64_stats/*/*_kernel_stats.csv | head -1) $P/${T}_lightning64_kernel_stats.csv
[ -d $D/stack64_stats ] && cp $(ls $D/stack64_stats/*/*_kernel_stats.csv | head -1) $P/${T}_stack64_kernel_stats.csv
for w in 2ranks_one_gpu train_2ranks_one_gpu; do [ -s $D/bench_$w.json ] && cp $D/bench_$w.json $P/${T}_bench_$w.json; done
[ -s $D/bench_stress_bf16.json ] && cp $D/bench_stress_bf16.json $P/${T}_bench_stress_bf16.json
[ -s $D/bench_stress_bf16mma.json ] && cp $D/bench_stress_bf16mma.json $P/${T}_bench_stress_bf16mma.json
[ -s $D/census64.txt ] && grep -v amdgpu.ids $D/census64.txt > $P/${T}_census64.txt
[ -d $D/stress_bf16mma_stats ] && cp $(ls $D/stress_bf16mma_stats/*/*_kernel_stats.csv | head -1) $P/${T}_stress_bf16mma_kernel_stats.csv
[ -d $D/stress_bf16mma_pmc_FETCH_SIZE ] && python3 $R/tools/pmc_summary.py $D/stress_bf16mma_pmc_FETCH_SIZE $D/stress_bf16mma_pmc_WRITE_SIZE ${T}_stress_bf16mma csv-only
[ -d $D/stress_stats ] && cp $(ls $D/stress_stats/*/*_kernel_stats.csv | head -1) $P/${T}_stress_kernel_stats.csv
[ -d $D/stress_bf16_stats ] && cp $(ls $D/stress_bf16_stats/*/*_kernel_stats.csv | head -1) $P/${T}_stress_bf16_kernel_stats.csv
[ -d $D/stress_pmc_FETCH_SIZE ] && python3 $R/tools/pmc_summary.py $D/stress_pmc_FETCH_SIZE $D/stress_pmc_WRITE_SIZE ${T}_stress csv-only
[ -d $D/stress_bf16_pmc_FETCH_SIZE ] && python3 $R/tools/pmc_summary.py $D/stress_bf16_pmc_FETCH_SIZE $D/stress_bf16_pmc_WRITE_SIZE ${T}_stress_bf16 csv-only
[ -d $D/stack_stats ] && cp $(ls $D/stack_stats/*/*_kernel_stats.csv | head -1) $P/${T}_stack_kernel_stats.csv
# published bench files hold exactly the JSON line (launcher / gloo chatter stripped)
python3 - <<PY
import glob, json
for f in glob.glob("$P/${T}_bench_*.json") + glob.glob("$P/${T}_final_bench.json"):
    lines = [ln for ln in open(f).read().splitlines() if ln.startswith("{")]
    if lines:
        json.loads(lines[-1])
        open(f, "w").write(lines[-1] + "\n")
PY
ls -la $P | grep ${T}_ | wc -l
