"""Clock the chip holds in each kernel: GRBM_GUI_ACTIVE (summed over the 8 XCDs) / 8 / duration, from one
`rocprofv3 --kernel-trace --pmc GRBM_GUI_ACTIVE` pass of bench.py (tools/collect_profiles.sh).

    python tools/clock_summary.py gpurun_out/final/pmc_GRBM_GUI_ACTIVE profiles/r01_final_clock_per_kernel.csv"""
import collections, csv, glob, os, sys

d, out = sys.argv[1], sys.argv[2]
cc = max(glob.glob(os.path.join(d, "*", "*_counter_collection.csv")), key=os.path.getmtime)
kt = max(glob.glob(os.path.join(d, "*", "*_kernel_trace.csv")), key=os.path.getmtime)
dur = {r["Dispatch_Id"]: int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))}
agg = collections.defaultdict(lambda: [0, 0.0, 0.0])
for r in csv.DictReader(open(cc)):
    if r["Counter_Name"] != "GRBM_GUI_ACTIVE" or r["Dispatch_Id"] not in dur:
        continue
    k = r["Kernel_Name"].split("(")[0].replace("void ", "")
    a = agg[k]
    a[0] += 1
    a[1] += float(r["Counter_Value"])
    a[2] += dur[r["Dispatch_Id"]]
rows = sorted(agg.items(), key=lambda kv: -kv[1][2])
with open(out, "w") as f:
    f.write("kernel,dispatches,avg_duration_us,clock_GHz(GRBM_GUI_ACTIVE/8/duration)\n")
    for k, (n, cyc, ns) in rows:
        if ns > 0:
            f.write("%s,%d,%.1f,%.3f\n" % (k, n, ns / n / 1e3, cyc / 8.0 / ns))
print(open(out).read()[:1800])
