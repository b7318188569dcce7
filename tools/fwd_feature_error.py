"""Dev tool: error structure of the node features entering the global pooling, per arithmetic mode."""
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
cap = {}
def hook(tag):
    def f(m, i):
        cap[tag] = i[0].detach().double().cpu()
    return f
def run(m, dev, dt, tag):
    h = m.cry_pool.register_forward_pre_hook(hook(tag))
    bb = recipe.GraphBatch(b.x.to(dt).to(dev), b.edge_index.to(dev), b.edge_attr.to(dev), b.batch.to(dev))
    r = tuple(t.to(dev).to(dt) if t.is_floating_point() else t.to(dev) for t in roost)
    with torch.no_grad():
        m(bb, r)
    h.remove()
run(om64, "cpu", torch.float64, "f64"); run(om, "cpu", torch.float32, "o32")
for mode in ("f32", "bf16x6"):
    P.set_bilinear_mode(mode)
    pm = P.CGAtNet(200, 128, 4, msg_heads=3, neighbor_number=12, update_edges=True)
    pm.load_state_dict(om.state_dict()); pm = pm.to("cuda:0")
    run(pm, "cuda:0", torch.float32, mode)
ref = cap["f64"]
print("fea", tuple(ref.shape), "max|fea|", float(ref.abs().max()))
for tag in ("o32", "f32", "bf16x6"):
    d = cap[tag] - ref
    rows = d.abs().max(dim=1).values
    print(f"{tag:7s} max|err| {float(d.abs().max()):.3e}  mean err {float(d.mean()):+.3e}  rms {float(d.pow(2).mean().sqrt()):.3e}  "
          f"worst rows {rows.topk(5).indices.tolist()} ({[f'{v:.1e}' for v in rows.topk(5).values.tolist()]})  "
          f"rows with err>1e-5: {int((rows > 1e-5).sum())}")
