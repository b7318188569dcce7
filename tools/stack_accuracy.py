"""Dev tool: full-stack (random init, 40 crystals) gradient errors of the HIP path vs the fp64 oracle,
per arithmetic mode, next to the fp32 oracle's own deviation from fp64."""
import copy, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import torch
import recipe
import cgat_amd as P
from oracle import cgat_oracle as O

b, roost = P.synthetic_batch(40, 20, 12, seed=8)
torch.manual_seed(1)
om = O.CGAtNet(200, 128, 4, msg_heads=3, neighbor_number=12, update_edges=True)
om64 = copy.deepcopy(om).double()


def run(m, dev, dt):
    bb = recipe.GraphBatch(b.x.to(dt).to(dev), b.edge_index.to(dev), b.edge_attr.to(dev), b.batch.to(dev))
    r = tuple(t.to(dev).to(dt) if t.is_floating_point() else t.to(dev) for t in roost)
    y = m(bb, r)
    cot = torch.randn(y.shape, generator=torch.Generator().manual_seed(9)).to(dt).to(dev)
    ps = dict(m.named_parameters())
    g = torch.autograd.grad((y * cot).sum(), list(ps.values()), allow_unused=True)
    return y.detach().double().cpu(), {k: (None if v is None else v.detach().double().cpu()) for k, v in zip(ps, g)}


y64, g64 = run(om64, "cpu", torch.float64)
y32, g32 = run(om, "cpu", torch.float32)
scale = max(float(v.abs().max()) for v in g64.values() if v is not None)
res = {}
for mode in ("f32", "bf16x6", "bf16x3", "f16x3"):
    P.set_bilinear_mode(mode)
    pm = P.CGAtNet(200, 128, 4, msg_heads=3, neighbor_number=12, update_edges=True)
    pm.load_state_dict(om.state_dict())
    pm = pm.to("cuda:0")
    yp, gp = run(pm, "cuda:0", torch.float32)
    res[mode] = (yp, gp)
    print(f"mode {mode}: out max-norm rel err {float((yp - y64).abs().max() / y64.abs().max()):.2e}  (oracle fp32: {float((y32 - y64).abs().max() / y64.abs().max()):.2e})")
print(f"largest gradient of the case: {scale:.3e}")
print(f"{'tensor':74s} {'|ref|':>9s} {'orac32':>9s} {'f32':>9s} {'bf16x6':>9s} {'bf16x3':>9s} {'f16x3':>9s}   (abs err vs fp64 / largest gradient)")
rows = []
for k, v in g64.items():
    if v is None:
        continue
    e = lambda g: float((g[k] - v).abs().max()) / scale
    rows.append((max(e(res["bf16x6"][1]), e(res["f32"][1]), e(res["f16x3"][1])), k, float(v.abs().max()) / scale, e(g32), e(res["f32"][1]), e(res["bf16x6"][1]),
                 e(res["bf16x3"][1]), e(res["f16x3"][1])))
for r in sorted(rows, reverse=True)[:14]:
    print(f"{r[1]:74s} {r[2]:9.2e} {r[3]:9.2e} {r[4]:9.2e} {r[5]:9.2e} {r[6]:9.2e} {r[7]:9.2e}")
rel = lambda g, k: float((g[k] - g64[k]).abs().max() / g64[k].abs().max())
worst = {m: max((rel(res[m][1], k), k) for k in g64 if g64[k] is not None and float(g64[k].abs().max()) > 1e-4 * scale) for m in res}
for m, (e, k) in worst.items():
    print(f"mode {m}: worst per-tensor max-norm relative error among gradients >= 1e-4 of the largest: {e:.2e} ({k})")
print("oracle fp32 worst:", max((rel(g32, k), k) for k in g64 if g64[k] is not None and float(g64[k].abs().max()) > 1e-4 * scale))
