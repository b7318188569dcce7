#!/usr/bin/env python3
"""Merges the per-mode parity reports written by `pytest -m gpu` (gpurun_out/parity_report_<mode>.txt, one run per
arithmetic mode) into profiles/<tag>_parity_report_<mode>.txt (copies) and profiles/<tag>_parity_attribution.txt: every
tensor of the BASELINE-shaped fixtures that is more than 1e-4 of |ref| away from the oracle's fp32 run in the default
mode (f16x3c), with its error in the four modes, the oracle's own fp32-vs-fp64 deviation nf, and what the excess is
attributed to.   python tools/parity_attribution.py gpurun_out r04"""
import collections, os, re, shutil, sys

src, tag = sys.argv[1], sys.argv[2]
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MODES = ("f16x3c", "bf16x6", "f16x3", "f32")   # the default mode first
ROW = re.compile(r"\s+(\S+)\s+([\d.e+-]+)\s+([\d.]+)\s+([\d.]+)\s+([\d.e+-]+)\s+(.*)")


def load(mode):
    rows, cur = collections.OrderedDict(), None
    path = os.path.join(src, f"parity_report_{mode}.txt")
    for line in open(path):
        if line.startswith("["):
            cur = line.split("]")[0][1:]
            continue
        m = ROW.match(line)
        if m:
            rows[(cur, m.group(1))] = dict(rel=float(m.group(2)), err_nf=float(m.group(3)), err64_nf=float(m.group(4)),
                                           nf_rel=float(m.group(5)), verdict=m.group(6).strip())
    shutil.copy(path, os.path.join(ROOT, "profiles", f"{tag}_parity_report_{mode}.txt"))
    return rows


rep = {m: load(m) for m in MODES}
out = ["Tensors of the BASELINE-shaped fixtures (closed-form sin-pattern parameters, derivative patterns forced) whose error vs the",
       "oracle's fp32 run exceeds 1e-4 |ref| in the default mode (f16x3c, 24-bit operands), in all four arithmetic modes.",
       "  err = ||hip - oracle32||_inf / ||oracle64||_inf;  nf = ||oracle32 - oracle64||_inf / ||oracle64||_inf (the reference's own fp32 noise)",
       "  x64 = ||hip - oracle64||_inf / nf: how many times the reference's own fp32 deviation the HIP result is from the fp64 truth",
       "", f"{'case / tensor':86s} {'nf':>9s} | " + " | ".join(f"{m:>8s} err   x64" for m in MODES) + " | attribution"]
n_listed = 0
stats = collections.Counter()
for key, r in rep[MODES[0]].items():
    case, name = key
    if not (case in ("net_mean", "nodes_first0", "nodes_first1") or case.startswith("net_")):
        continue
    if r["rel"] <= 1e-4 or r["nf_rel"] == 0 or r["rel"] > 1e3:      # (rel > 1e3: |ref| ~ 0, admitted by the zero floor)
        continue
    cols = []
    for m in MODES:
        q = rep[m].get(key)
        cols.append(f"{q['rel']:.2e} {q['err64_nf']:5.2f}" if q else f"{'-':>8s} {'-':>5s}")
    others = [rep[m][key]["rel"] for m in ("bf16x6", "f32") if key in rep[m]]
    if "case_scale" in r["verdict"]:
        why = "numerically zero gradient (|ref64| <= 1e-7 of the case's largest gradient): admitted by the zero floor"
    elif r["nf_rel"] >= 2.5e-5 and r["err64_nf"] <= 4:
        why = "oracle fp32 itself >= 2.5e-5 from fp64; hip within 4 nf of fp64: summation-order noise of a cancelling sum"
    elif others and min(others) <= 1e-4:
        why = "arithmetic of the default mode (bf16x6 or f32 is within 1e-4)"
    else:
        why = "summation order (all modes alike)"
    stats[why] += 1
    n_listed += 1
    out.append(f"{(case + ' ' + name)[:86]:86s} {r['nf_rel']:.2e} | " + " | ".join(cols) + f" | {why}")
out += ["", f"{n_listed} tensors listed; attribution counts: " + "; ".join(f"{v} x {k}" for k, v in stats.items())]
for m in MODES:
    rows = [r for k, r in rep[m].items() if r["rel"] < 1e3 and r["nf_rel"] > 0]
    xs = sorted(r["err64_nf"] for r in rows)
    out.append(f"mode {m}: {len(rows)} compared tensors; (hip - oracle64) / nf: median {xs[len(xs) // 2]:.2f}, 90th percentile "
               f"{xs[int(0.9 * len(xs))]:.2f}, max {xs[-1]:.2f}; tensors above 1e-4 |ref|: {sum(1 for r in rows if r['rel'] > 1e-4)}")
open(os.path.join(ROOT, "profiles", f"{tag}_parity_attribution.txt"), "w").write("\n".join(out) + "\n")
print("\n".join(out[-8:]))
