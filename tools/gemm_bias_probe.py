"""Dev tool: is the engine's error on CANCELLING sums (random-sign operands) biased?  mean and rms of the signed error
over all outputs, in units of sum_k |a||b|; |mean| >> rms / sqrt(outputs) means a coherent offset."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)
print("CGAT_GEMM_SPLIT =", os.environ.get("CGAT_GEMM_SPLIT", "(default: on)"), " passes", os.environ.get("CGAT_GEMM_SPLIT_PASSES", "8"))
for (M, N, K, bkm) in [(1024, 1024, 1024, 0), (1024, 1024, 1024, 1), (1024, 1024, 64, 0), (512, 512, 8192, 0)]:
    g = torch.Generator().manual_seed(K)
    A = torch.randn(M, K, generator=g).to(dev)
    B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
    Cm = torch.empty(M, N, device=dev)
    d = _lib.GemmDesc()
    d.alpha, d.beta, d.splits = 1.0, 0.0, 1
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.B, d.ldb, d.b_kmajor, d.C, d.ldc = A.data_ptr(), K, B.data_ptr(), B.shape[1], bkm, Cm.data_ptr(), N
    _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), "gemm")
    torch.cuda.synchronize()
    Bd = B.double() if bkm else B.double().t()
    ref = A.double() @ Bd
    scale = float((A.double().abs() @ Bd.abs()).mean())
    e = Cm.double() - ref
    n = e.numel()
    print(f"{M}x{N}x{K} bkm={bkm}: mean {float(e.mean()) / scale:+.2e}  rms {float(e.pow(2).mean().sqrt()) / scale:.2e}  "
          f"(rms/sqrt(n) = {float(e.pow(2).mean().sqrt()) / scale / n ** 0.5:.1e})", flush=True)
