"""Dev tool: tiny golden case `roost` -- where does the error of graphs.0.pooling.0.gate_nn.fc_out.weight come from?
(a) per-segment sums of the gate-logit gradient (zero in exact arithmetic), (b) the weight-gradient reduction redone in
fp64 from the HIP run's own operands."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT]
import torch
import recipe
import cgat_amd as P
from oracle import cgat_oracle as O


def ns(M):
    return types.SimpleNamespace(MultiHeadNetwork=M.MultiHeadNetwork, GATConvNodes=M.GATConvNodes, GATConvEdges=M.GATConvEdges,
                                 MHAttention=M.MHAttention, CGAtNet=M.CGAtNet, H_Net_0=M.H_Net_0, H_Net=M.H_Net,
                                 SimpleNetwork=M.SimpleNetwork, ResidualNetwork=M.ResidualNetwork,
                                 WeightedAttention=M.WeightedAttention, MessageLayer=M.MessageLayer, Roost=M.Roost,
                                 RoostSimpleNetwork=M.SimpleNetwork)


def run(case, dtype, device):
    torch.manual_seed(1)
    mod = recipe.fill_params(case.mk()).to(dtype).to(device)
    rec = {}
    gate = mod.graphs[0].pooling[0].gate_nn

    def hook(m, inp, out):
        rec["gate_out"] = out.detach().double().cpu()
        out.register_hook(lambda g: rec.__setitem__("g_gate", g.detach().double().cpu()))
    gate.register_forward_hook(hook)

    def hook_fc(m, inp, out):
        rec["hidden"] = inp[0].detach().double().cpu()
    gate.fc_out.register_forward_hook(hook_fc)
    inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.inputs(dtype).items()}
    rec["index"] = inputs["self_fea_idx"].cpu() if "self_fea_idx" in inputs else None
    y = case.call(mod, inputs)
    w = gate.fc_out.weight
    (gw,) = torch.autograd.grad((y * recipe.cotangent(y).to(device)).sum(), [w])
    rec["gw"] = gw.detach().double().cpu()
    rec["inputs"] = {k: v for k, v in inputs.items()}
    return rec


a = run(recipe.tiny_cases(ns(P))["roost"], torch.float32, "cuda:0")
b32 = run(recipe.tiny_cases(ns(O))["roost"], torch.float32, "cpu")
b64 = run(recipe.tiny_cases(ns(O))["roost"], torch.float64, "cpu")
print("inputs:", {k: (tuple(v.shape) if torch.is_tensor(v) else v) for k, v in a["inputs"].items()})
idx = None
for k, v in b64["inputs"].items():
    if torch.is_tensor(v) and v.dtype == torch.int64 and v.dim() == 1 and v.numel() == b64["g_gate"].shape[0]:
        idx = v.cpu(); print("segment index from input", k); break
for name, r in (("hip", a), ("oracle32", b32), ("oracle64", b64)):
    g = r["g_gate"].reshape(-1)
    line = f"{name:9s} max|g_gate| {float(g.abs().max()):.3e}  err vs 64 {float((g - b64['g_gate'].reshape(-1)).abs().max()):.3e}"
    if idx is not None:
        S = int(idx.max()) + 1
        ssum = torch.zeros(S, dtype=torch.float64).index_add(0, idx, g)
        line += f"  max |segment sum| {float(ssum.abs().max()):.3e}"
    gw_re = (r["g_gate"].reshape(-1, 1) * b64["hidden"]).sum(0)   # (the fp64 run's hidden rows: their own error is 1e-7)
    line += (f"  |dW| {float(r['gw'].abs().max()):.3e} err(dW) {float((r['gw'].reshape(-1) - b64['gw'].reshape(-1)).abs().max()):.3e}"
             f"  err(fp64 reduction of OWN g_gate, hidden) {float((gw_re - b64['gw'].reshape(-1)).abs().max()):.3e}")
    print(line)
    print("          g_gate:", " ".join(f"{float(x):+.6e}" for x in g))
h = b64["hidden"]
print("hidden rows: max |h| %.3e, max deviation from its segment mean %.3e" % (
    float(h.abs().max()), float((h - (torch.zeros(int(idx.max()) + 1, h.shape[1], dtype=torch.float64).index_add(0, idx, h) /
                                      torch.bincount(idx).double()[:, None])[idx]).abs().max()) if idx is not None else -1))
