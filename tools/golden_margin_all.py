"""Dev tool: the worst per-tensor margin (error / allowed) of every tiny golden case on the GPU."""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT]
import numpy as np  # noqa: E402
import torch  # noqa: E402

import golden_util as G  # noqa: E402
import recipe  # noqa: E402
import cgat_amd as P  # noqa: E402

print("engine:", os.environ.get("CGAT_GEMM_SPLIT", "split"))
import types  # noqa: E402
NS = types.SimpleNamespace(MultiHeadNetwork=P.MultiHeadNetwork, GATConvNodes=P.GATConvNodes, GATConvEdges=P.GATConvEdges,
                           MHAttention=P.MHAttention, CGAtNet=P.CGAtNet, H_Net_0=P.H_Net_0, H_Net=P.H_Net,
                           SimpleNetwork=P.SimpleNetwork, ResidualNetwork=P.ResidualNetwork,
                           WeightedAttention=P.WeightedAttention, MessageLayer=P.MessageLayer, Roost=P.Roost,
                           RoostSimpleNetwork=P.SimpleNetwork)
ONLY = sys.argv[1] if len(sys.argv) > 1 else ""
for cname, case in recipe.tiny_cases(NS).items():
    if ONLY not in cname:
        continue
    ref = G.case_arrays("tiny.npz", cname)
    y, grads, _ = recipe.run_case(case, torch.float32, device="cuda:0")
    out_err = G.maxnorm_rel(y.detach().cpu().numpy(), ref["out"])
    nf_out = np.abs(ref["out"].astype(np.float64) - ref["out_f64"]).max() / max(np.abs(ref["out_f64"]).max(), 1e-300)
    case_scale = max([float(v[1]) for k, v in ref.items() if k.startswith("nf.")] + [0.0])
    worst = (0.0, "")
    for name, g in grads.items():
        if name + ".none" in ref or g is None or name not in ref:
            continue
        nf_abs, ref_max = ref["nf." + name]
        err = np.abs(g.detach().cpu().numpy().astype(np.float64) - ref[name]).max()
        allowed = max(1e-4 * ref_max, G.NOISE_MULT * nf_abs, 1e-6 * case_scale)
        if err / allowed > worst[0]:
            worst = (err / allowed, name)
    print(f"  {cname:22s} out err {out_err:.2e} (ref noise {nf_out:.1e})  worst grad margin {worst[0]:.2f} {worst[1][:60]}", flush=True)
