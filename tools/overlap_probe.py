"""Dev tool: does an MFMA-bound contraction kernel overlap with an HBM-bound stream on a second HIP stream?
Times (a) 4 weight-gradient launches, (b) an elementwise pass over 6 GB repeated 4 times, (c) both back to back on one
stream, (d) both concurrently on two streams."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib
dev = "cuda:0"
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q, r = (torch.randn(rows, W, generator=g).to(dev) for _ in range(3))
out = torch.empty(W, W, W, device=dev)
ws = torch.empty(_lib.lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
big = torch.randn(1500000000 // 4 * 4, device=dev)            # 6 GB
big2 = torch.empty_like(big)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()

def mfma(stream):
    for _ in range(4):
        _lib.check(_lib.lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, out.data_ptr(), rows, W, W, W,
                                                ws.data_ptr(), ws.numel(), stream.cuda_stream), "wgrad")
def hbm(stream):
    with torch.cuda.stream(stream):
        for _ in range(4):
            torch.mul(big, 2.0, out=big2)

def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize(); e0.record()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps

cur = torch.cuda.current_stream()
a = timed(lambda: mfma(cur))
b = timed(lambda: hbm(cur))
c = timed(lambda: (mfma(cur), hbm(cur)))
def both():
    s1.wait_stream(cur); s2.wait_stream(cur)
    mfma(s1); hbm(s2)
    cur.wait_stream(s1); cur.wait_stream(s2)
d = timed(both)
print(f"4 wgrad: {a:.2f} ms | 4 x 12 GB elementwise: {b:.2f} ms | sequential: {c:.2f} ms | two streams: {d:.2f} ms")
