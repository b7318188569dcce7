"""Dev tool: time cgat_gemm variants of the edge-phase shapes in isolation (HIP events)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib, ops
import cgat_amd as P

dev = torch.device("cuda:0")
b, _ = P.synthetic_batch(4167, 20, 12, seed=0)
ei = b.edge_index.to(dev)
N, E = b.num_nodes, ei.shape[1]
plan = ops.EdgePlan(ei, N)
W2, Ce, D = 1536, 128, 384
g = torch.Generator().manual_seed(0)
e = torch.randn(E, Ce, generator=g).to(dev)
Wcat = torch.randn(W2, D, generator=g).to(dev)
Pi = torch.randn(N, W2, generator=g).to(dev)
Pj = torch.randn(N, W2, generator=g).to(dev)
Z = torch.empty(E, W2, device=dev)
gE = torch.empty(E, Ce, device=dev)
gW = torch.empty(W2, D, device=dev)
ws = torch.empty(1 << 30, dtype=torch.uint8, device=dev)


def run(name, flops, **kw):
    d = _lib.GemmDesc()
    d.alpha, d.beta, d.splits = 1.0, 0.0, 1
    for k, v in kw.items():
        setattr(d, k, v.data_ptr() if torch.is_tensor(v) else v)
    for _ in range(2):
        _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), name)
    torch.cuda.synchronize()
    t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0.record()
    for _ in range(3):
        _lib.check(_lib.lib.cgat_gemm(C.byref(d), ws.data_ptr(), ws.numel(), None), name)
    t1.record(); torch.cuda.synchronize()
    ms = t0.elapsed_time(t1) / 3
    print(f"{name:58s} {ms:8.3f} ms  {flops / ms / 1e9:7.1f} TFLOP/s", flush=True)


fz = 2.0 * E * W2 * Ce
base = dict(M=E, N=W2, K=Ce, A=e, lda=Ce, B=Wcat[:, 128:], ldb=D, C=Z, ldc=W2)
run("Z plain (no gather, no adds)", fz, **base)
run("Z + row gather", fz, **base, a_rgather=plan.dst_perm)
run("Z + adds (no gather)", fz, **base, add1=Pi, add1_idx=plan.dst_sorted, add2=Pj, add2_idx=plan.src_sorted, ld_add=W2)
run("Z full (gather + adds)", fz, **base, a_rgather=plan.dst_perm, add1=Pi, add1_idx=plan.dst_sorted, add2=Pj,
    add2_idx=plan.src_sorted, ld_add=W2)
run("Z narrow N=128 (1 n-tile), full", fz / 12, **dict(base, N=128), a_rgather=plan.dst_perm, add1=Pi,
    add1_idx=plan.dst_sorted, add2=Pj, add2_idx=plan.src_sorted, ld_add=W2)
run("g_e = gZ @ W_e (K=1536), scatter", fz, M=E, N=Ce, K=W2, A=Z, lda=W2, B=Wcat[:, 128:], ldb=D, b_kmajor=1, C=gE, ldc=Ce,
    c_scatter=plan.dst_perm)
run("g_e without scatter", fz, M=E, N=Ce, K=W2, A=Z, lda=W2, B=Wcat[:, 128:], ldb=D, b_kmajor=1, C=gE, ldc=Ce)
for sp in (0, 16, 64, 128):
    run(f"gW_e = gZ^T @ e[perm] (K=E) splits={sp}", fz, M=W2, N=Ce, K=E, A=Z, lda=W2, a_kmajor=1, B=e, ldb=Ce, b_kmajor=1,
        b_kgather=plan.dst_perm, C=gW[:, 128:], ldc=D, splits=sp)
Zb = Z.reshape(E, 12, 128).permute(1, 0, 2).contiguous()
for sp in (0, 128):
    run(f"gW_e blocked gZ layout splits={sp}", fz, M=W2, N=Ce, K=E, A=Zb, lda=128, a_block=E * 128, a_kmajor=1, B=e, ldb=Ce,
        b_kmajor=1, b_kgather=plan.dst_perm, C=gW[:, 128:], ldc=D, splits=sp)
run("g_e blocked gZ layout", fz, M=E, N=Ce, K=W2, A=Zb, lda=128, a_block=E * 128, B=Wcat[:, 128:], ldb=D, b_kmajor=1, C=gE,
    ldc=Ce, c_scatter=plan.dst_perm)
run("gW_e without k-gather splits=0", fz, M=W2, N=Ce, K=E, A=Z, lda=W2, a_kmajor=1, B=e, ldb=Ce, b_kmajor=1, C=gW[:, 128:],
    ldc=D, splits=0)
x = torch.randn(N, 128, generator=g).to(dev)
y = torch.empty(N, 128, device=dev)
run("trunk layer [N,128]x[128,128] tanh", 2.0 * N * 128 * 128, M=N, N=128, K=128, A=x, lda=128, B=Wcat[:128, :128], ldb=D,
    C=y, ldc=128, act=1)
run("node proj [N,128]x[128,1536]", 2.0 * N * 128 * W2, M=N, N=W2, K=128, A=x, lda=128, B=Wcat, ldb=D, C=Pi, ldc=W2)
A4 = torch.randn(8192, 4096, generator=g).to(dev); B4 = torch.randn(4096, 4096, generator=g).to(dev); C4 = torch.empty(8192, 4096, device=dev)
run("square-ish 8192x4096x4096 NT", 2.0 * 8192 * 4096 * 4096, M=8192, N=4096, K=4096, A=A4, lda=4096, B=B4, ldb=4096, C=C4, ldc=4096)

import os
for abl in ("0", "1", "2", "4", "6", "7"):
    os.environ["CGAT_GEMM_ABL"] = abl
    run(f"square NT, ablation {abl} (1 no barrier, 2 no gloads, 4 no LDS stores)", 2.0 * 8192 * 4096 * 4096, M=8192, N=4096, K=4096, A=A4, lda=4096, B=B4, ldb=4096, C=C4, ldc=4096)
os.environ["CGAT_GEMM_ABL"] = "0"
