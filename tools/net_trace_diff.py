"""Dev tool: where a tiny golden network case departs from the oracle -- every submodule's output and the gradient
arriving at it, HIP (fp32, GPU) against the oracle in fp64 with the same parameters and inputs."""
import os
import sys
import types

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden"), ROOT]
import torch  # noqa: E402

import recipe  # noqa: E402
import cgat_amd as P  # noqa: E402
from oracle import cgat_oracle as O  # noqa: E402


def ns(M):
    return types.SimpleNamespace(MultiHeadNetwork=M.MultiHeadNetwork, GATConvNodes=M.GATConvNodes, GATConvEdges=M.GATConvEdges,
                                 MHAttention=M.MHAttention, CGAtNet=M.CGAtNet, H_Net_0=M.H_Net_0, H_Net=M.H_Net,
                                 SimpleNetwork=M.SimpleNetwork, ResidualNetwork=M.ResidualNetwork,
                                 WeightedAttention=M.WeightedAttention, MessageLayer=M.MessageLayer, Roost=M.Roost,
                                 RoostSimpleNetwork=M.SimpleNetwork)


def trace(case, dtype, device):
    torch.manual_seed(1)
    mod = recipe.fill_params(case.mk()).to(dtype).to(device)
    rec = {}

    def fwd_hook(name):
        def h(m, inp, out):
            o = out[0] if isinstance(out, (tuple, list)) else out
            if torch.is_tensor(o) and o.is_floating_point():
                rec["out:" + name] = o.detach().double().cpu()
                if o.requires_grad:
                    o.register_hook(lambda g, n=name: rec.__setitem__("gout:" + n, g.detach().double().cpu()))
        return h
    for name, m in mod.named_modules():
        if name:
            m.register_forward_hook(fwd_hook(name))
    inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.inputs(dtype).items()}
    for v in inputs.values():
        if torch.is_tensor(v) and v.is_floating_point():
            v.requires_grad_(True)
    y = case.call(mod, inputs)
    params = dict(mod.named_parameters())
    g = torch.autograd.grad((y * recipe.cotangent(y).to(device)).sum(), list(params.values()), allow_unused=True)
    for (n, _), gg in zip(params.items(), g):
        if gg is not None:
            rec["gp:" + n] = gg.detach().double().cpu()
    return rec


cname = sys.argv[1]
a = trace(recipe.tiny_cases(ns(P))[cname], torch.float32, "cuda:0")
b = trace(recipe.tiny_cases(ns(O))[cname], torch.float64, "cpu")
print("engine:", os.environ.get("CGAT_GEMM_SPLIT", "split"))
scale_g = max(float(v.abs().max()) for k, v in b.items() if k.startswith("gp:"))
rows = []
for k in b:
    if k in a and a[k].shape == b[k].shape:
        den = float(b[k].abs().max())
        err = float((a[k] - b[k]).abs().max())
        rows.append((k, err / max(den, 1e-300), err / scale_g if k.startswith("gp:") else float("nan")))
for kind in ("out:", "gout:", "gp:"):
    sel = sorted([r for r in rows if r[0].startswith(kind)], key=lambda r: -r[1])[:14]
    print(kind)
    for k, rel, rs in sel:
        print(f"   {k[:90]:90s} err/|ref| {rel:.2e}   err/scale {rs:.2e}")
print("activation patterns (output of rezero k != 0 <=> ReLU active):")
for k in sorted(b):
    if k.startswith("out:output_nn.rezeros."):
        pa, pb = a[k] != 0, b[k] != 0
        d = (pa != pb)
        if d.any():
            idx = d.nonzero()
            vals = [(float(a[k][tuple(i)]), float(b[k][tuple(i)])) for i in idx[:4]]
            print(f"   {k}: {int(d.sum())} of {d.numel()} differ, e.g. (hip, oracle64) {vals}   max|out| {float(b[k].abs().max()):.2e}")
        else:
            print(f"   {k}: same pattern")
print("output head, layer by layer (err / |ref|max):")
for k in sorted(b):
    if "output_nn" in k and k in a and a[k].shape == b[k].shape:
        den = float(b[k].abs().max())
        print(f"   {k:40s} {float((a[k] - b[k]).abs().max()) / max(den, 1e-300):.2e}   |ref| {den:.2e}")
