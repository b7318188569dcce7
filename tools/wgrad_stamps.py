"""Dev tool (GPU box; needs a library built with CGAT_HIPCC_FLAGS=-DWGC_STAMPS): phase boundaries of the f16x3c weight
gradient kernel in shader cycles, workgroup 0, iterations 100..103.  Per wave and iteration six s_memtime stamps:
grp 0 (waves 0-3): 0 top, 1 q landed, 2 split done, 3 matrix phase done, 4 DMA landed, 5 barrier passed;
grp 1 (waves 4-7): 0 top, 1 matrix phase done, 2 q landed, 3 split done, 4 vmcnt(0), 5 barrier passed."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib
dev = torch.device("cuda:0")
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q, r = (torch.randn(rows, W, generator=g).to(dev) for _ in range(3))
out = torch.zeros(W, W, W, device=dev)
ws = torch.empty(_lib.lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
_lib.lib.cgat_set_bilinear_mode(4)
for _ in range(3):
    _lib.check(_lib.lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, out.data_ptr(), rows, W, W, W,
                                            ws.data_ptr(), ws.numel(), None), "wgrad")
torch.cuda.synchronize()
st = out.view(-1)[:512].view(torch.int64).cpu().view(8, 4, 8)[:, :, :8]   # [6]: split arithmetic done, [7]: first of the three image conversions issued
t0 = int(st[:, 0, 0].min())
for it in range(4):
    print(f"iteration {100 + it}")
    for w in range(8):
        print(f"  wave {w} (grp {w >> 2}):", " ".join(f"{int(x) - t0:7d}" for x in st[w, it]))
