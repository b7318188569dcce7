"""Child process of tests/test_hip_kernels.py::test_per_edge_forward_wide_tile_is_bit_identical: one GATConvNodes attention
forward + backward on a seeded batch; prints SHA-256 digests of the saved tensors (Z, coefficients, weighted
sums), the output and the gradients.  The per-edge forward's form is chosen by the environment (CGAT_EDGE_Z6W), which the
library reads once per process."""
import hashlib, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import ops


def digest(t):
    return hashlib.sha256(t.detach().contiguous().cpu().numpy().tobytes()).hexdigest()


def main():
    mode, storage, graphs = sys.argv[1], sys.argv[2], int(sys.argv[3])
    P.set_bilinear_mode(mode)
    ops.set_edge_storage(storage)
    b, _ = P.synthetic_batch(graphs, 20, 12, seed=3)     # 61 crystals: E = 14 640, a ragged last tile for either form
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
    y = ops.NodesAttentionFn.apply(x, e, plan, 3, *W)
    saved = [t for t in y.grad_fn.saved_tensors if t is not None and t.dtype == torch.float32 and t.numel() >= E * 3]
    grads = torch.autograd.grad((y * cot).sum(), [x, e] + W)
    torch.cuda.synchronize()
    out = {"E": E, "N": N, "y": digest(y), "saved": [digest(t) for t in saved], "grads": [digest(t) for t in grads]}
    if storage == "f32":
        # the same kernel as the first layer of the vector-attention variants: LeakyReLU'd rows stored, their maximum returned
        w_in = (0.1 * torch.randn(512, 384, generator=g)).to(dev).requires_grad_(True)
        b_in = (0.1 * torch.randn(512, generator=g)).to(dev).requires_grad_(True)
        hidden, hmax = ops.EdgeHiddenFn.apply(x, e, plan, w_in, b_in)
        gh = torch.autograd.grad((hidden * hidden).sum(), [x, e, w_in, b_in])
        torch.cuda.synchronize()
        out["hidden"] = [digest(hidden), digest(hmax)] + [digest(t) for t in gh]
    print("DIGEST " + json.dumps(out))


if __name__ == "__main__":
    main()
