"""Turns the rocprofv3 PMC passes of `bench.py` (separate FETCH_SIZE / WRITE_SIZE runs, --kernel-trace only) into
profiles/pmc_contraction_kernels.json (HBM bytes per launch of the three hypernetwork contraction kernels, read by
bench.py for roofline.traffic) and a per-kernel CSV.

    python tools/pmc_summary.py gpurun_out/pmcF_FETCH_SIZE gpurun_out/pmcF_WRITE_SIZE r01

gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE counts wide coalesced reads at half their bytes, so the read
side is doubled; WRITE_SIZE is exact for 16-byte streaming stores.  Both counters are in KiB."""
import collections, csv, glob, json, os, sys

fetch_dir, write_dir, tag = sys.argv[1], sys.argv[2], sys.argv[3]
CSV_ONLY = len(sys.argv) > 4 and sys.argv[4] == "csv-only"   # (other workloads: the per-kernel CSV, not the headline's JSON)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def load(d, counter):
    f = max(glob.glob(os.path.join(d, "**", "*_counter_collection.csv"), recursive=True), key=os.path.getmtime)
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "").split("<")[0]].append(float(r["Counter_Value"]))
    return agg, f


fetch, ff = load(fetch_dir, "FETCH_SIZE")
write, wf = load(write_dir, "WRITE_SIZE")
rows = []
for k in sorted(set(fetch) | set(write), key=lambda k: -(sum(fetch.get(k, [0])) * 2 + sum(write.get(k, [0])))):
    nf, nw = len(fetch.get(k, [])), len(write.get(k, []))
    f_kb = sum(fetch.get(k, [0])) / max(nf, 1)
    w_kb = sum(write.get(k, [0])) / max(nw, 1)
    rows.append((k, max(nf, nw), f_kb, w_kb, int((2 * f_kb + w_kb) * 1024)))
with open(os.path.join(ROOT, "profiles", f"{tag}_pmc_hbm_per_kernel.csv"), "w") as f:
    f.write("kernel,dispatches,FETCH_SIZE_KB_per_launch,WRITE_SIZE_KB_per_launch,hbm_bytes_per_launch(2*FETCH+WRITE)\n")
    for r in rows:
        f.write("%s,%d,%.1f,%.1f,%d\n" % r)
HBM_KERNELS = ("edge_z_kernel", "edge_z6w_kernel", "edge_zx_kernel", "seg_bwd_msg_kernel", "seg_bwd_soft_kernel", "seg_bwd_att_kernel",
               "edge_seg_bwd_kernel", "seg_wsum_vec_kernel", "seg_softmax_fwd_kernel", "edge_gj_kernel", "edge_ge_kernel",
               "edge_gw_kernel", "mlp_chain128_x6_kernel", "rows_dw128_split_batch_kernel")
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from tree_id import stamp
commit = stamp(os.path.dirname(os.path.abspath(fetch_dir.rstrip("/"))))
MODE = os.environ.get("CGAT_BILINEAR_MODE", "f16x3c")
if CSV_ONLY:
    # other workloads (the 64 M-edge stress step, fp32 / bf16 edge storage): the largest launch of every HBM-bound kernel
    # = one closed chunk's per-edge launch, for bench.py's frac_from_counters of that workload
    sf = os.path.join(ROOT, "profiles", "pmc_stress_kernels.json")
    allw = json.load(open(sf)) if os.path.exists(sf) else {}
    allw[tag.split("_", 1)[1] if "_" in tag else tag] = {
        "collected_at_commit": commit, "mode": MODE,
        "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024",
        "hbm_bytes_largest_launch": {k: int((2 * max(fetch.get(k, [0.0])) + max(write.get(k, [0.0]))) * 1024)
                                     for k in HBM_KERNELS if k in fetch or k in write}}
    json.dump(allw, open(sf, "w"), indent=1)
    sys.exit(0)
N, C = 83340, 128
# f16x3: the prepared T / r operands are two fp16 planes = 4 bytes per element; f16x3c (the default since round 4): the
# prepared T is 25 KB per 32 output columns and `a` = 6.25 bytes per element; the bf16 forms: three 2-byte planes
alg = {"bilinear_rows128_ring16_kernel": 4 * N * C * 4 + C ** 3 * 4,           # p, q, init, out + the two planes of T
       "bilinear_rows128_dual_kernel": 7 * N * C * 4 + C ** 3 * 4,             # p, q, zz, init1, out1, init2, out2 + T
       "bilinear_rows128_ring16c_kernel": 4 * N * C * 4 + C * 4 * 25600,       # p, q, init, out + the f16x3c image of T
       "bilinear_rows128_dualc_kernel": 7 * N * C * 4 + C * 4 * 25600,
       "bilinear_wgrad128_bf16_kernel": 2 * N * C * 4 + N * C * 6 + C ** 3 * 4,  # pT, qT, three bf16 planes of r + out
       # the batched f16x3 launch covers the four predicted layers: 4 x (pT, qT, two fp16 planes of r, out)
       "bilinear_wgrad128_f16p_kernel": 4 * (2 * N * C * 4 + N * C * 4 + C ** 3 * 4),
       # the batched f16x3c launch (round 5), four predicted layers: pT, qF (fp32), the r stream of 50 KB per 64 rows
       # (6.25 bytes per element), out
       "bilinear_wgrad128_f16c_kernel": 4 * (2 * N * C * 4 + (N // 64 + 1) * 51200 + C ** 3 * 4)}
out = {"collected_at_commit": commit, "mode": MODE,
       "command": "rocprofv3 --kernel-trace --pmc {FETCH_SIZE|WRITE_SIZE} --output-format csv -- python3 bench.py "
                  "--steps 1 --warmup 1 --no-cpu-baseline (two separate passes)",
       "correction": "hbm_bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 (gfx950: FETCH_SIZE halves wide coalesced reads)"}
for k, n, f_kb, w_kb, b in rows:
    if k in alg:
        out[k] = {"dispatches": n, "FETCH_SIZE_KB_per_launch": round(f_kb, 1), "WRITE_SIZE_KB_per_launch": round(w_kb, 1),
                  "hbm_bytes_per_launch": b, "algorithmic_bytes_per_launch": alg[k]}
# per-edge kernels: the per-EDGE launches only (the same template also runs the small per-node products): the launch with the
# most bytes of each name
def _largest(agg, k):
    return max(agg.get(k, [0.0]))
hb = {}
for k in HBM_KERNELS:
    if k in fetch or k in write:
        f_kb, w_kb = _largest(fetch, k), _largest(write, k)
        hb[k] = {"FETCH_SIZE_KB_largest_launch": round(f_kb, 1), "WRITE_SIZE_KB_largest_launch": round(w_kb, 1),
                 "hbm_bytes_largest_launch": int((2 * f_kb + w_kb) * 1024)}
out["hbm_bound_kernels"] = hb
json.dump(out, open(os.path.join(ROOT, "profiles", "pmc_contraction_kernels.json"), "w"), indent=1)
print(json.dumps(out, indent=1))
