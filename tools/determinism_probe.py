"""Dev tool: run the attention op twice and report which saved tensors / gradients differ bitwise; also compare
the fused edge-Z path against the f32 GEMM path at the same size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import ops

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 200
b, _ = P.synthetic_batch(graphs, 20, 12, seed=2)
g = torch.Generator().manual_seed(6)
N, E = b.num_nodes, b.edge_index.shape[1]
dev = "cuda:0"
x = torch.randn(N, 128, generator=g).to(dev).requires_grad_(True)
e = torch.randn(E, 128, generator=g).to(dev).requires_grad_(True)
torch.manual_seed(1)
layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)
plan = ops.get_plan(b.edge_index.to(dev), N)
W = [layer.MH_A.fc_in.weight, layer.MH_A.fc_in.bias, layer.MH_A.fc_out.weight, layer.MH_A.fc_out.bias,
     layer.MH_M.fc_in.weight, layer.MH_M.fc_in.bias, layer.MH_M.fc_out.weight, layer.MH_M.fc_out.bias]
W = [w.reshape(w.shape[0], -1) if w.dim() == 3 else w for w in W]
cot = torch.randn(N, 128, generator=g).to(dev)
names = ["x", "e", "A_in_w", "A_in_b", "A_out_w", "A_out_b", "M_in_w", "M_in_b", "M_out_w", "M_out_b"]

def run():
    y = ops.NodesAttentionFn.apply(x, e, plan, 3, *W)
    saved = y.grad_fn.saved_tensors[2]
    grads = torch.autograd.grad((y * cot).sum(), [x, e] + W)
    torch.cuda.synchronize()
    return y.detach().clone(), saved.detach().clone(), [t.detach().clone() for t in grads]

def parts(saved):
    Z = saved[:E * 1536].view(E, 1536)
    al = saved[E * 1536:E * 1536 + E * 3].view(E, 3)
    return Z, al

for mode in (("bf16x6",) if os.environ.get("PROBE_FAST") else ("bf16x6", "f32")):
    P.set_bilinear_mode(mode)
    r = [run() for _ in range(3)]
    for k in (1, 2):
        Z0, a0 = parts(r[0][1]); Zk, ak = parts(r[k][1])
        dz = (Z0 != Zk)
        print(f"[{mode}] run0 vs run{k}: Z differs at {int(dz.sum())} elements (rows {torch.nonzero(dz.any(1)).flatten()[:8].tolist()}, "
              f"cols {torch.nonzero(dz.any(0)).flatten()[:8].tolist()}), alpha differs at {int((a0 != ak).sum())}, out differs at {int((r[0][0] != r[k][0]).sum())}")
        for n, u, v in zip(names, r[0][2], r[k][2]):
            nd = int((u != v).sum())
            if nd:
                print(f"    grad {n}: {nd} elements differ, max {float((u - v).abs().max()):.3e}")
    res = r[0]
    if mode == "bf16x6":
        keep = res
Zf, af = parts(res[1]); Zb, ab = parts(keep[1])
print("fused vs f32 path: Z max diff %.3e, alpha max diff %.3e, out max diff %.3e" % (
    float((Zf - Zb).abs().max()), float((af - ab).abs().max()), float((res[0] - keep[0]).abs().max())))

# which logits are wrong?  recompute alpha from the saved Z with torch and compare row by row
P.set_bilinear_mode(P.ops.DEFAULT_MODE)
y, saved, _ = run()
Z, al = parts(saved)
wA = W[2].reshape(3, 256); bA = W[3].reshape(3)
zl = torch.nn.functional.leaky_relu(Z[:, :768].reshape(E, 3, 256), 0.01)
a_ref = (zl * wA[None]).sum(-1) + bA[None]
# softmax over destination segments with the plan's CSR (in-degrees vary in the synthetic batch)
rp = plan.dst_rowptr.long()
seg = torch.repeat_interleave(torch.arange(N, device=dev), rp[1:] - rp[:-1])
mx = torch.full((N, 3), -1e30, device=dev).scatter_reduce(0, seg[:, None].expand(E, 3), a_ref, "amax")
ex = (a_ref - mx[seg]).exp()
sm = torch.zeros(N, 3, device=dev).index_add_(0, seg, ex)
al_ref = ex / (sm[seg] + 1e-16)
bad = ((al - al_ref).abs() > 1e-5).any(1)
idx = torch.nonzero(bad).flatten()
print("rows with wrong alpha:", int(bad.sum()), "of", E)
print("first bad rows:", idx[:40].tolist())
print("bad rows mod 128 histogram (top):", torch.bincount(idx % 128, minlength=128).tolist())
print("bad per head:", ((al - al_ref).abs() > 1e-5).sum(0).tolist())
