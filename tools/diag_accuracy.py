"""Dev tool: where does the HIP path's rounding differ from the oracle's?  Runs one golden case
on (product, cuda), (oracle fp32), (oracle fp64) with forward hooks on every submodule and prints
max-norm relative deviations from the fp64 run, module by module, then the same for gradients."""
import os, sys, types
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden")); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import recipe
import cgat_amd as P
from oracle import cgat_oracle as O


def ns(M):
    return types.SimpleNamespace(
        MultiHeadNetwork=M.MultiHeadNetwork, GATConvNodes=M.GATConvNodes, GATConvEdges=M.GATConvEdges,
        MHAttention=M.MHAttention, CGAtNet=M.CGAtNet, H_Net_0=M.H_Net_0, H_Net=M.H_Net, SimpleNetwork=M.SimpleNetwork,
        ResidualNetwork=M.ResidualNetwork, WeightedAttention=M.WeightedAttention, MessageLayer=M.MessageLayer,
        Roost=M.Roost, RoostSimpleNetwork=M.SimpleNetwork)


cname = sys.argv[1] if len(sys.argv) > 1 else "net_mean"
table = sys.argv[2] if len(sys.argv) > 2 else "tiny"
get = recipe.tiny_cases if table == "tiny" else recipe.base_cases


def run(M, dtype, device):
    case = get(ns(M))[cname]
    torch.manual_seed(1)
    mod = recipe.fill_params(case.mk()).to(dtype).to(device)
    outs = {}

    def hook(name):
        def f(m, i, o):
            if torch.is_tensor(o):
                outs[name] = o.detach().double().cpu()
        return f
    for name, m in mod.named_modules():
        if name:
            m.register_forward_hook(hook(name))
    inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.inputs(dtype).items()}
    leaves = {k: v for k, v in inputs.items() if torch.is_tensor(v) and v.is_floating_point()}
    for v in leaves.values():
        v.requires_grad_(True)
    y = case.call(mod, inputs)
    params = dict(mod.named_parameters())
    gr = torch.autograd.grad((y * recipe.cotangent(y).to(device)).sum(), list(leaves.values()) + list(params.values()),
                             allow_unused=True)
    grads = {n: (None if g is None else g.detach().double().cpu())
             for n, g in zip(["in." + k for k in leaves] + list(params), gr)}
    return y.detach().double().cpu(), outs, grads


y64, o64, g64 = run(O, torch.float64, "cpu")
y32, o32, g32 = run(O, torch.float32, "cpu")
yp, op, gp = run(P, torch.float32, "cuda:0")


def rel(a, b):
    d = b.abs().max().item()
    return (a - b).abs().max().item() / (d if d > 0 else 1.0)


print(f"{'module output':70s} {'hip vs f64':>11s} {'orac32 vs f64':>13s} ratio")
for k in o64:
    if k in op and k in o32 and op[k].shape == o64[k].shape:
        a, b = rel(op[k], o64[k]), rel(o32[k], o64[k])
        if a > 4 * b and a > 1e-6:
            print(f"{k:70s} {a:11.2e} {b:13.2e} {a / max(b, 1e-30):6.1f}")
print(f"{'OUT':70s} {rel(yp, y64):11.2e} {rel(y32, y64):13.2e}")
print("gradients with hip error > 4x oracle fp32 error:")
for k in g64:
    if g64[k] is None or gp.get(k) is None:
        continue
    a, b = rel(gp[k], g64[k]), rel(g32[k], g64[k])
    if a > 4 * b and a > 1e-6:
        print(f"{k:70s} {a:11.2e} {b:13.2e} {a / max(b, 1e-30):6.1f}")
