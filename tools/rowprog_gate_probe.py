"""Dev tool (GPU box): intermediates of Roost's crystal-pooling gate network (net_embed fixture) with the modules'
small-row programs on and off, and an fp64 re-evaluation of the fc_out weight gradient from each run's own operands."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import torch, recipe
    from test_hip_golden import product_ns
    case = recipe.tiny_cases(product_ns())["net_embed"]
    cap = {}
    def ctx(mod):
        net = mod.roost.cry_pool[0].gate_nn
        def hook(m, inp, out):
            cap["x"] = inp[0].detach().clone(); cap["gate"] = out.detach().clone()
            out.register_hook(lambda g: cap.__setitem__("g_gate", g.detach().clone()))
        net.register_forward_hook(hook)
        import contextlib
        return contextlib.nullcontext()
    y, grads, mod = recipe.run_case(case, torch.float32, device="cuda:0", ctx=ctx)
    net = mod.roost.cry_pool[0].gate_nn
    W0, b0, W1 = net.fcs[0].weight.detach().double(), net.fcs[0].bias.detach().double(), net.fc_out.weight.detach().double()
    x = cap["x"].double()
    pre = x @ W0.t() + b0
    h = torch.where(pre > 0, pre, 0.01 * pre)
    gW1_from_own = (cap["g_gate"].double().t() @ h)            # fp64 from this run's g_gate and exact h
    cap.update(gW1=grads["gp.roost.cry_pool.0.gate_nn.fc_out.weight"].detach().clone(), gW1_own64=gW1_from_own.float(),
               h64=h.float())
    torch.save({k: v.cpu() for k, v in cap.items()}, sys.argv[2])
    sys.exit(0)
import torch
outs = {}
for tag, pyrows in (("on", "2048"), ("off", "0")):
    env = dict(os.environ, CGAT_ROWPROG_PY_MAX_ROWS=pyrows)
    f = f"/tmp/gate_probe_{tag}.pt"
    subprocess.check_call([sys.executable, __file__, "--child", f], env=env)
    outs[tag] = torch.load(f)
a, b = outs["on"], outs["off"]
for k in a:
    d = (a[k].double() - b[k].double()).abs().max().item()
    print(f"{k:10s} shape {tuple(a[k].shape)} max|on| {a[k].abs().max().item():.4e} max|on-off| {d:.3e}")
for tag in ("on", "off"):
    o = outs[tag]
    print(tag, "kernel gW1 vs fp64-from-own-operands:", (o["gW1"].double() - o["gW1_own64"].double()).abs().max().item(),
          " sum g_gate:", o["g_gate"].double().sum().item(), " |g_gate| max", o["g_gate"].abs().max().item())
print("g_gate on :", a["g_gate"].flatten()[:12].tolist())
print("g_gate off:", b["g_gate"].flatten()[:12].tolist())
print("gate on :", a["gate"].flatten()[:12].tolist())
print("gate off:", b["gate"].flatten()[:12].tolist())
