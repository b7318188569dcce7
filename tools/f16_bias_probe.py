"""Dev tool: signed-error statistics (bias?) of the f16x3 kernels on cancelling sums, in units of sum |x||w|:
linear128 (K = 128 -> N), the K > 128 -> 128 route (edge_ge form), the chain kernel, the width-128 contraction."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib
P = lambda t: None if t is None else t.data_ptr()


def stats(name, got, ref, scale):
    e = got.double() - ref
    n = e.numel()
    s = float(scale.mean())
    print(f"{name:34s} mean {float(e.mean()) / s:+.2e}  rms {float(e.pow(2).mean().sqrt()) / s:.2e}  (rms/sqrt(n) {float(e.pow(2).mean().sqrt()) / s / n ** 0.5:.1e})",
          flush=True)


print("mode", ops.get_bilinear_mode())
g = torch.Generator().manual_seed(3)
M = 8192
for K, N in [(128, 128), (128, 1536), (256, 128), (1536, 128)]:
    x = torch.randn(M, K, generator=g).to(dev)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    y = torch.empty(M, N, device=dev)
    xmax = x.abs().max().reshape(1)
    ws = torch.empty(lib.cgat_linear_forward_workspace_bytes(M, K, N), dtype=torch.uint8, device=dev)
    _lib.check(lib.cgat_linear_forward(P(x), K, P(w), K, None, P(y), N, M, K, N, _lib.ACT_NONE, P(xmax), P(ws), ws.numel(), None), "fwd")
    torch.cuda.synchronize()
    stats(f"linear {K} -> {N}", y, x.double() @ w.double().t(), x.double().abs() @ w.double().abs().t())
W = 128
rows = 4096
p, q = torch.randn(rows, W, generator=g).to(dev), torch.randn(rows, W, generator=g).to(dev)
T = (torch.randn(W, W, W, generator=g) / W).to(dev)
out = torch.empty(rows, W, device=dev)
ws = torch.empty(lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
_lib.check(lib.cgat_bilinear_rows(P(p), W, P(q), W, P(T), None, W, P(out), W, rows, W, W, W, P(ws), ws.numel(), None), "rows")
torch.cuda.synchronize()
stats("contraction (rows form, W = 128)", out, torch.einsum("na,nb,abc->nc", p.double(), q.double(), T.double()),
      torch.einsum("na,nb,abc->nc", p.double().abs(), q.double().abs(), T.double().abs()))
