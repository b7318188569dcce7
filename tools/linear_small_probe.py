"""Dev tool: ops.linear (cgat_linear_forward / _backward) on few-row shapes against fp64: y, g_x, g_w, g_b errors."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib, ops  # noqa: E402

dev = torch.device("cuda:0")
print("CGAT_GEMM_SPLIT =", os.environ.get("CGAT_GEMM_SPLIT", "(default: on)"))
for (M, K, N) in [(4, 1024, 1024), (4, 1024, 512), (4, 512, 512), (4, 384, 1024), (4, 256, 128), (64, 1024, 1024), (4, 1024, 2)]:
    g = torch.Generator().manual_seed(M + K + N)
    x = torch.randn(M, K, generator=g).to(dev).requires_grad_(True)
    w = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev).requires_grad_(True)
    b = torch.randn(N, generator=g).to(dev).requires_grad_(True)
    cot = torch.randn(M, N, generator=g).to(dev)
    y = ops.linear(x, w, b, _lib.ACT_RELU)
    gx, gw, gb = torch.autograd.grad((y * cot).sum(), [x, w, b])
    xd, wd, bd = (t.detach().double().requires_grad_(True) for t in (x, w, b))
    yd = torch.relu(xd @ wd.t() + bd)
    rx, rw, rb = torch.autograd.grad((yd * cot.double()).sum(), [xd, wd, bd])
    rel = lambda a, r: float((a.double() - r).abs().max() / r.abs().max())
    print(f"M={M:3d} K={K:5d} N={N:5d}  y {rel(y, yd.detach()):.2e}  g_x {rel(gx, rx):.2e}  g_w {rel(gw, rw):.2e}  g_b {rel(gb, rb):.2e}", flush=True)
