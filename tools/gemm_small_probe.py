"""Dev tool: max-norm and rms error of cgat_gemm against fp64 on small / ragged shapes, every operand layout."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
ws = torch.empty(1 << 26, dtype=torch.uint8, device=dev)
print("CGAT_GEMM_SPLIT =", os.environ.get("CGAT_GEMM_SPLIT", "(default: on)"))
for (M, N, K) in [(4, 1024, 1024), (4, 1024, 48), (80, 64, 48), (80, 32, 96), (1280, 128, 200), (80, 16, 16), (80, 96, 32),
                  (960, 32, 48), (80, 1, 32), (4, 2, 128), (300, 48, 80)]:
    for akm in (False, True):
        for bkm in (False, True):
            g = torch.Generator().manual_seed(M + N + K)
            A = torch.randn((K, M) if akm else (M, K), generator=g).to(dev)
            B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
            Cm = torch.full((M, N), float("nan"), device=dev)
            d = _lib.GemmDesc()
            d.alpha, d.beta, d.splits = 1.0, 0.0, 1
            d.M, d.N, d.K = M, N, K
            d.A, d.lda, d.a_kmajor = A.data_ptr(), A.shape[1], int(akm)
            d.B, d.ldb, d.b_kmajor = B.data_ptr(), B.shape[1], int(bkm)
            d.C, d.ldc = Cm.data_ptr(), N
            _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), "gemm")
            torch.cuda.synchronize()
            ref = (A.t() if akm else A).double() @ (B if bkm else B.t()).double()
            e = (Cm.double() - ref)
            scale = (A.double().abs() if not akm else A.double().abs().t()) @ (B.double().abs() if bkm else B.double().abs().t())
            print(f"{M:5d}x{N:5d}x{K:5d} akm={int(akm)} bkm={int(bkm)}  max err/|ref|max {float(e.abs().max() / ref.abs().max()):.2e}"
                  f"  max err/sum|a||b| {float((e.abs() / scale).max()):.2e}", flush=True)
