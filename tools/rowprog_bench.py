"""Dev tool (GPU box): what a grid barrier, a phase and a launch of the small-row programs cost (HIP events, 100 reps)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import _lib, rowprog as rp
dev = torch.device("cuda:0")
def timeit(fn, reps=100):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / reps * 1e3
g = torch.Generator().manual_seed(0)
rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
# (1) n trivial phases
x, w = rnd(16, 16), rnd(16, 16)
outs = [torch.empty(16, 16, device=dev) for _ in range(9)]
for n in (1, 2, 4, 8):
    ops = [rp.op(p, 16, 16, 16, x if p == 0 else outs[p - 1], w, outs[p]) for p in range(n)]
    print(f"trivial program, {n} phases: {timeit(lambda: rp.run(ops, dev)):.1f} us")
# (1b) n trivial phases that keep 128 workgroups busy
xb, wb = rnd(2048, 16), rnd(16, 16)
outb = [torch.empty(2048, 16, device=dev) for _ in range(9)]
for n in (1, 2, 8):
    ops = [rp.op(p, 2048, 16, 16, xb if p == 0 else outb[p - 1], wb, outb[p]) for p in range(n)]
    print(f"2048-row x 16 program (128 tiles/phase), {n} phases: {timeit(lambda: rp.run(ops, dev)):.1f} us")
# (2) single products
for (M, N, K) in ((1280, 128, 128), (1280, 384, 128), (1280, 128, 1536), (64, 1024, 1024), (64, 1024, 128), (436, 256, 256), (128, 128, 1280)):
    A, B, C = rnd(M, K), rnd(N, K), torch.empty(M, N, device=dev)
    o = [rp.op(0, M, N, K, A, B, C)]
    print(f"single op {M} x {N} x {K}: {timeit(lambda: rp.run(o, dev)):.1f} us")
# (3) the output head
torch.manual_seed(0)
net = P.ResidualNetwork(128, 2, [1024, 1024, 512, 512, 256, 256, 128]).to(dev)
xh = rnd(64, 128).requires_grad_(True)
with torch.no_grad():
    print(f"head forward (8 phases): {timeit(lambda: net(xh)):.1f} us")
def fb():
    y = net(xh); y.sum().backward()
print(f"head forward + backward: {timeit(fb, 50):.1f} us (incl. autograd host time)")
