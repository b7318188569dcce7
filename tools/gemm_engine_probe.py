"""Dev tool: throughput of the generic engine (cgat_gemm) on large plain products, per operand layout.
Run with CGAT_GEMM_SPLIT=0 for the f32-input engine, default for the six-pass bf16 engine."""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
ws = torch.empty(1 << 28, dtype=torch.uint8, device=dev)


def run(name, M, N, K, akm, bkm, reps=5):
    g = torch.Generator().manual_seed(0)
    A = torch.randn((K, M) if akm else (M, K), generator=g).to(dev)
    B = torch.randn((K, N) if bkm else (N, K), generator=g).to(dev)
    Cm = torch.empty(M, N, device=dev)
    d = _lib.GemmDesc()
    d.alpha, d.beta, d.splits = 1.0, 0.0, 1
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.a_kmajor = A.data_ptr(), A.shape[1], int(akm)
    d.B, d.ldb, d.b_kmajor = B.data_ptr(), B.shape[1], int(bkm)
    d.C, d.ldc = Cm.data_ptr(), N
    for _ in range(2):
        _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), name)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(reps):
        _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), name)
    t1.record()
    torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / reps
    ref = (A.t() if akm else A)[:64].double() @ (B if bkm else B.t()).double()
    err = float((Cm[:64].double() - ref).abs().max() / ref.abs().max())
    print(f"{name:40s} {ms:8.3f} ms {2.0 * M * N * K / ms / 1e9:7.1f} TFLOP/s  err {err:.1e}", flush=True)


def bias(M, N, K):
    """Mean SIGNED relative error of a product of positive operands (every running sum grows with one sign): rounding
    noise averages out over the M x N outputs, a biased accumulator does not."""
    g = torch.Generator().manual_seed(1)
    A = torch.rand(M, K, generator=g).to(dev) + 0.5
    B = torch.rand(N, K, generator=g).to(dev) + 0.5
    Cm = torch.empty(M, N, device=dev)
    d = _lib.GemmDesc()
    d.alpha, d.beta, d.splits = 1.0, 0.0, 1
    d.M, d.N, d.K = M, N, K
    d.A, d.lda, d.B, d.ldb, d.C, d.ldc = A.data_ptr(), K, B.data_ptr(), K, Cm.data_ptr(), N
    _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), "bias")
    torch.cuda.synchronize()
    ref = A.double() @ B.double().t()
    rel = (Cm.double() - ref) / ref
    print(f"signed error of {M}x{N}x{K}, positive operands: mean {float(rel.mean()):+.2e}  rms {float(rel.pow(2).mean().sqrt()):.2e}", flush=True)


print("CGAT_GEMM_SPLIT =", os.environ.get("CGAT_GEMM_SPLIT", "(default: on)"))
for akm in (False, True):
    for bkm in (False, True):
        run(f"8192x4096x4096 akm={int(akm)} bkm={int(bkm)}", 8192, 4096, 4096, akm, bkm)
run("1000080x1536x128 (per-edge fwd)", 1000080, 1536, 128, False, False, reps=3)
run("83340x128x4096 (outer-like K)", 83340, 128, 4096, False, True, reps=3)
bias(2048, 2048, 16384)
bias(512, 512, 83340)
