"""Dev tool: the fused edge-Z kernel (split-bf16) against the generic GEMM + row-dot path (f32 mode): saved Z,
alpha, S and the layer gradients, same inputs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import ops

graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 60
b, _ = P.synthetic_batch(graphs, 20, 12, seed=3)
g = torch.Generator().manual_seed(4)
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
res = {}
for mode in ("f32", "bf16x6"):
    P.set_bilinear_mode(mode)
    captured = {}
    orig = ops.NodesAttentionFn.forward
    y = ops.NodesAttentionFn.apply(x, e, plan, 3, *W)
    saved = y.grad_fn.saved_tensors[2] if hasattr(y.grad_fn, "saved_tensors") else None
    grads = torch.autograd.grad((y * cot).sum(), [x, e] + [w for w in W])
    torch.cuda.synchronize()
    res[mode] = (y.detach().clone(), saved.detach().clone(), [t.detach().clone() for t in grads])
HHd = 768
Z0, Z1 = res["f32"][1][:E * 1536].view(E, 1536), res["bf16x6"][1][:E * 1536].view(E, 1536)
d = (Z0 - Z1).abs()
print("Z  max |diff| %.3e  max|Z| %.3e   argmax row %d col %d" % (float(d.max()), float(Z0.abs().max()),
      int(d.argmax()) // 1536, int(d.argmax()) % 1536))
print("rows with diff > 1e-4:", torch.nonzero(d.max(dim=1).values > 1e-4).flatten()[:20].tolist())
print("cols with diff > 1e-4:", torch.nonzero(d.max(dim=0).values > 1e-4).flatten()[:20].tolist())
al0 = res["f32"][1][E * 1536:E * 1536 + E * 3]; al1 = res["bf16x6"][1][E * 1536:E * 1536 + E * 3]
print("alpha max diff %.3e" % float((al0 - al1).abs().max()))
print("out max diff %.3e (max %.3e)" % (float((res["f32"][0] - res["bf16x6"][0]).abs().max()), float(res["f32"][0].abs().max())))
names = ["x", "e", "A_in_w", "A_in_b", "A_out_w", "A_out_b", "M_in_w", "M_in_b", "M_out_w", "M_out_b"]
for n, a, c in zip(names, res["f32"][2], res["bf16x6"][2]):
    print("grad %-8s max diff %.3e  (max %.3e)" % (n, float((a - c).abs().max()), float(a.abs().max())))
sign_flips = int(((Z0 > 0) != (Z1 > 0)).sum())
print("sign flips between the two Z:", sign_flips)
