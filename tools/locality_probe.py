"""Dev tool (GPU box): test_beyond_int32_element_counts' comparison (last 50 crystals of a 7000-crystal batch vs alone),
per quantity, in the current arithmetic mode; CGAT_ROWPROG_MAX_ROWS=0 puts the small batch on the big batch's kernels."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
dev = "cuda:0"
G, A, K = 7000, 20, 12
b, _ = P.synthetic_batch(G, A, K, seed=5)
N, E = b.num_nodes, b.edge_index.shape[1]
torch.manual_seed(1)
m = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)
g = torch.Generator().manual_seed(8)
x, e, x0 = (torch.randn(s, 128, generator=g).to(dev) for s in (N, E, N))
ei = b.edge_index.to(dev)
n0, e0 = (G - 50) * A, (G - 50) * A * K
cot = torch.randn(50 * A, 128, generator=g).to(dev)
def run(xs, eis, es, x0s, sl):
    xs, es = xs.clone().requires_grad_(True), es.clone().requires_grad_(True)
    with P.debug.record_masks(m) as masks:
        y = m(xs, eis, es, x0s)
    gx, ge = torch.autograd.grad((y[sl] * cot).sum(), [xs, es])
    return y[sl].detach(), gx[sl].detach(), ge, masks
yb, gxb, geb, mb = run(x, ei, e, x0, slice(n0, N))
ys, gxs, ges, ms = run(x[n0:].contiguous(), (ei[:, e0:] - n0).contiguous(), e[e0:].contiguous(), x0[n0:].contiguous(), slice(0, 50 * A))
rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
print("mode", P.get_bilinear_mode(), "rowprog rows", os.environ.get("CGAT_ROWPROG_MAX_ROWS", "2048"))
print(" y", rel(yb, ys), " gx", rel(gxb, gxs), " ge", rel(geb[e0:], ges))
flips = 0
for k in mb:
    for a_, b_ in zip(mb[k], ms[k]):
        a2 = a_[-b_.shape[0]:] if a_.shape[0] != b_.shape[0] else a_
        flips += int((a2 != b_).sum())
print(" derivative-pattern entries that differ between the two evaluations:", flips)
