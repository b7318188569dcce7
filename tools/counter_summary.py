#!/usr/bin/env python3
"""Per-kernel sums of rocprofv3 --pmc counter_collection CSVs: python tools/counter_summary.py <dir> [<dir> ...]
Prints kernel, launches, and the mean per launch of every counter found; with SQ_VALU_MFMA_BUSY_CYCLES and
GRBM_GUI_ACTIVE in the same pass also the matrix-core utilisation busy / (GUI_ACTIVE / 8 XCDs x 1024 SIMDs)."""
import csv, glob, json, os, sys
from collections import defaultdict


def load(d):
    acc = defaultdict(lambda: defaultdict(float))
    cnt = defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"].split("(")[0].replace("void ", "")
            acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
            cnt[k].add(r["Dispatch_Id"])
    return acc, cnt


def main():
    out = {}
    for d in sys.argv[1:]:
        acc, cnt = load(d)
        for k in sorted(acc, key=lambda k: -acc[k].get("GRBM_GUI_ACTIVE", acc[k].get("SQ_WAVE_CYCLES", 0))):
            n = len(cnt[k])
            row = {c: v / n for c, v in acc[k].items()}
            if "SQ_VALU_MFMA_BUSY_CYCLES" in row and "GRBM_GUI_ACTIVE" in row and row["GRBM_GUI_ACTIVE"] > 0:
                row["mfma_util"] = row["SQ_VALU_MFMA_BUSY_CYCLES"] / (row["GRBM_GUI_ACTIVE"] / 8 * 1024)
            out.setdefault(k, {"launches": n}).update(row)
    for k, row in out.items():
        print(k[:60].ljust(60), " ".join(f"{c}={v:.4g}" for c, v in row.items()))
    if os.environ.get("COUNTER_JSON"):
        # compact form for the bench line: the kernels that use the matrix cores, utilisation to 3 digits
        sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
        from tree_id import stamp
        # the collection directory (parent of the pass directory) holds the tree id the GPU box computed
        commit = stamp(os.path.dirname(os.path.abspath(sys.argv[1].rstrip("/"))))
        brief = {"collected_at_commit": commit,
                 "what": "SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs x 1024 SIMDs) per kernel, mean over its "
                         "launches; rocprofv3 --pmc serialises kernels and the pass runs with CGAT_OVERLAP_WGRAD=0, "
                         "so every kernel has the whole chip",
                 "kernels": {k: {"launches": row["launches"], "mfma_util": round(row["mfma_util"], 3)}
                             for k, row in out.items() if row.get("mfma_util", 0) >= 0.01}}
        json.dump(brief, open(os.environ["COUNTER_JSON"], "w"), indent=1)


if __name__ == "__main__":
    main()
