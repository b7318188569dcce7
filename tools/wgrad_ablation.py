"""Dev tool: timing-only ablations of the weight-gradient kernel (CGAT_WGRAD_ABL: 1 no global loads, 2 no split,
4 no barrier; results are wrong by construction)."""
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
def call():
    _lib.check(_lib.lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, out.data_ptr(), rows, W, W, W,
                                            ws.data_ptr(), ws.numel(), None), "wgrad")
for _ in range(3): call()
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(10): call()
e1.record(); torch.cuda.synchronize()
print("CGAT_WGRAD_ABL=%s: %.3f ms per call (incl. transposes, plane split, slab sum)" % (os.environ.get("CGAT_WGRAD_ABL", "0"), e0.elapsed_time(e1) / 10))
