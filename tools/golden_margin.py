"""Dev tool: per-tensor margin (error / allowed) of one golden fixture case on the GPU."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden"))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as G  # noqa: E402
import recipe  # noqa: E402
import cgat_amd as P  # noqa: E402

fname, cname = sys.argv[1], sys.argv[2]
cases = recipe.tiny_cases(P) if fname.startswith("tiny") else recipe.base_cases(P)
case = cases[cname]
ref = G.case_arrays(fname, cname)
y, grads, _ = recipe.run_case(case, torch.float32, device="cuda:0")
case_scale = max([float(v[1]) for k, v in ref.items() if k.startswith("nf.")] + [0.0])
rows = []
for name, g in grads.items():
    if name + ".none" in ref or g is None or name not in ref:
        continue
    nf_abs, ref_max = ref["nf." + name]
    err = np.abs(g.detach().cpu().numpy().astype(np.float64) - ref[name]).max()
    allowed = max(1e-4 * ref_max, G.NOISE_MULT * nf_abs, 1e-6 * case_scale)
    rows.append((err / allowed, name, err, ref_max, nf_abs))
rows.sort(reverse=True)
print("engine:", os.environ.get("CGAT_GEMM_SPLIT", "split"), "case scale", case_scale)
for r in rows[:12]:
    print("  margin %.2f  %-60s err %.3e |ref| %.3e nf %.3e" % r)
