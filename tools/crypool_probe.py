"""Dev tool (round 4): why is `cry_pool.MH_A.fc_in.weight` of the BASELINE-shaped fixture `net_mean`
hundreds of times further from the fp64 truth than the oracle's own fp32 run, in every arithmetic mode?  Traces the
gradient arriving at every module of the crystal pooling (HIP fp32 on the GPU, oracle fp32 and fp64 on the CPU with the
HIP run's derivative patterns forced) and the conditioning of the weight-gradient reduction itself."""
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


def trace(case, dtype, device, ctx):
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
            i0 = inp[0] if isinstance(inp, (tuple, list)) and inp else None
            if torch.is_tensor(i0) and i0.is_floating_point():
                rec["in:" + name] = i0.detach().double().cpu()
        return h
    for name, m in mod.named_modules():
        if name.startswith("cry_pool") or name in ("output_nn",):
            m.register_forward_hook(fwd_hook(name))
    inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.inputs(dtype).items()}
    with ctx(mod):
        y = case.call(mod, inputs)
    params = dict(mod.named_parameters())
    sel = {n: p for n, p in params.items() if n.startswith("cry_pool")}
    g = torch.autograd.grad((y * recipe.cotangent(y).to(device)).sum(), list(sel.values()), allow_unused=True)
    for (n, _), gg in zip(sel.items(), g):
        if gg is not None:
            rec["gp:" + n] = gg.detach().double().cpu()
    return rec, mod


cname = sys.argv[1] if len(sys.argv) > 1 else "net_mean"
held = {}


def rec_ctx(mod):
    cm = P.debug.record_masks(mod)
    held["masks"] = cm.masks
    return cm


a, _ = trace(recipe.base_cases(ns(P))[cname], torch.float32, "cuda:0", rec_ctx)
forced = lambda mod: O.forced_masks(mod, held["masks"])
b32, _ = trace(recipe.base_cases(ns(O))[cname], torch.float32, "cpu", forced)
b64, m64 = trace(recipe.base_cases(ns(O))[cname], torch.float64, "cpu", forced)
print(f"case {cname}, mode {P.get_bilinear_mode()}")
print(f"{'tensor':58s} {'|ref64|':>10s} {'hip-ref64':>10s} {'o32-ref64':>10s} {'hip/|ref|':>10s} {'o32/|ref|':>10s}")
for k in sorted(b64):
    if k in a and k in b32 and a[k].shape == b64[k].shape:
        den = float(b64[k].abs().max())
        eh = float((a[k] - b64[k]).abs().max())
        eo = float((b32[k] - b64[k]).abs().max())
        print(f"{k[:58]:58s} {den:10.3e} {eh:10.3e} {eo:10.3e} {eh / max(den, 1e-300):10.2e} {eo / max(den, 1e-300):10.2e}")
# conditioning of dW = g_hid^T pair for MH_A.fc_in: sum |terms| / |sum| per output, and the two column halves apart
gh = b64.get("gout:cry_pool.MH_A.fc_in")
pin = b64.get("in:cry_pool.MH_A")
w = "gp:cry_pool.MH_A.fc_in.weight"
if w in b64:
    G = b64[w].reshape(b64[w].shape[0], -1)
    C2 = G.shape[1]
    for nm, sl in (("x half (node features)", slice(0, C2 // 2)), ("cry_fea half (constant per crystal)", slice(C2 // 2, C2))):
        den = float(G[:, sl].abs().max())
        eh = float((a[w].reshape(G.shape)[:, sl] - G[:, sl]).abs().max())
        eo = float((b32[w].reshape(G.shape)[:, sl] - G[:, sl]).abs().max())
        print(f"  {w} {nm}: |ref| {den:.3e}  hip err {eh:.3e} ({eh / den:.2e})  oracle32 err {eo:.3e} ({eo / den:.2e})")
if gh is not None and pin is not None:
    gh2 = gh.reshape(gh.shape[0], -1)
    x = pin.reshape(pin.shape[0], -1)
    print("  reduction dW[j,i] = sum_n g[n,j] x[n,i]: rows", gh2.shape[0], " g", tuple(gh2.shape), " x", tuple(x.shape))
    if gh2.shape[0] == x.shape[0]:
        S = gh2.t() @ x
        A = gh2.abs().t() @ x.abs()
        cond = (A / S.abs().clamp_min(1e-300))
        print(f"  conditioning sum|g x| / |sum g x|: median {float(cond.median()):.1f}, at the largest |dW| element "
              f"{float(cond.flatten()[S.abs().argmax()]):.1f}; max |sum|g x|| = {float(A.max()):.3e} vs max |dW| = {float(S.abs().max()):.3e}")
