"""Dev tool (GPU box): single products of the small-row programs at the shapes a 64-crystal step of the harness-default
network issues (long K beside few output tiles), hot and after a 512-MB flush, each checked against fp64."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import rowprog as rp
dev = torch.device("cuda:0")
def timeit(fn, reps=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
flush = torch.empty(512 << 20, dtype=torch.uint8, device=dev)
for (M, N, K, tr) in ((1280, 128, 2560, 0), (1280, 128, 1536, 0), (1280, 256, 128, 0), (1280, 2560, 128, 0), (1280, 128, 128, 0),
                      (128, 128, 1280, 1), (128, 2560, 1280, 1), (64, 1024, 1024, 0), (436, 256, 256, 0), (1280, 128, 768, 0)):
    if tr:      # weight-gradient layout: both operands k-major (A[k, m], B[k, n])
        At, Bt = rnd(K, M), rnd(K, N)
        A, B = At.t(), Bt.t()
    else:
        A, B = rnd(M, K), rnd(N, K)
    C = torch.empty(M, N, device=dev)
    o = [rp.op(0, M, N, K, A, B, C)]
    t_hot = timeit(lambda: rp.run(o, dev))
    def cold():
        flush.zero_(); rp.run(o, dev)
    t_fl = timeit(lambda: flush.zero_())
    t_cold = timeit(cold) - t_fl
    ref = (A.double() @ B.double().t()).float()
    err = float((C - ref).abs().max() / ref.abs().max())
    print(f"{M:5d} x {N:5d} x {K:5d} {'kmajor' if tr else 'rowmaj'}: hot {t_hot:6.1f} us, after a 512-MB flush {t_cold:6.1f} us, err {err:.1e}")
