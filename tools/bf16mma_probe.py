"""Dev tool (GPU box): per-tensor error of the K = 64 layer vs the oracle's fp64 run in the edge-storage modes
(tests/test_chunked.py::test_config5_bf16_edge_storage_vs_oracle's case)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from oracle import cgat_oracle as O
dev = "cuda:0"
b, _ = P.synthetic_batch(6, 20, 64, seed=47)
g = torch.Generator().manual_seed(48)
N, E = b.num_nodes, b.edge_index.shape[1]
x, e, x0, cot = (torch.randn(n, 128, generator=g) for n in (N, E, N, N))
torch.manual_seed(1)
om = O.GATConvNodes(128, 128, 128, 3, concat=True).double()
pm = P.GATConvNodes(128, 128, 128, 3, concat=True)
pm.load_state_dict({k: v.float() for k, v in om.state_dict().items()})
pm = pm.to(dev)
xo, eo = x.double().requires_grad_(True), e.double().requires_grad_(True)
yo = om(xo, b.edge_index, eo, x0.double())
names = ["x", "e"] + [n for n, _ in om.named_parameters()]
go = torch.autograd.grad((yo * cot.double()).sum(), [xo, eo] + list(om.parameters()))
scale = max(float(r.abs().max()) for r in go)
def run():
    xp, ep = x.to(dev).requires_grad_(True), e.to(dev).requires_grad_(True)
    y = pm(xp, b.edge_index.to(dev), ep, x0.to(dev))
    return y.detach(), torch.autograd.grad((y * cot.to(dev)).sum(), [xp, ep] + list(pm.parameters()))
for st in sys.argv[1:] or ["f32", "bf16", "bf16-mma"]:
    P.set_edge_storage(st)
    y, gs = run()
    P.set_edge_storage("f32")
    print(f"== {st}: out {float((y.double().cpu() - yo.detach()).abs().max() / yo.abs().max()):.2e}")
    rows = []
    for n, a, r in zip(names, gs, go):
        err = float((a.double().cpu() - r).abs().max())
        rows.append((err / max(float(r.abs().max()), 1e-3 * scale), err / float(r.abs().max()), float(r.abs().max()) / scale, n))
    for t in sorted(rows, reverse=True)[:8]:
        print(f"   {t[3]:60s} err/max(|r|,1e-3 scale) {t[0]:.2e}  err/|r| {t[1]:.2e}  |r|/scale {t[2]:.1e}")
