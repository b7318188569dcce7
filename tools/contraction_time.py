"""Dev tool: kernel-only timing of the three hypernetwork contraction entry points per arithmetic mode (run under
rocprofv3 --kernel-trace --stats to separate the operand preparation from the contraction kernels)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib
dev = torch.device("cuda:0")
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q, z = (torch.randn(rows, W, generator=g).to(dev) for _ in range(3))
T = (torch.randn(W, W, W, generator=g) / W).to(dev)
out, o1, o2 = (torch.empty(rows, W, device=dev) for _ in range(3))
ws = torch.empty(max(_lib.lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), _lib.lib.cgat_bilinear_dual_workspace_bytes(rows)), dtype=torch.uint8, device=dev)
modes = [int(m) for m in (sys.argv[1].split(",") if len(sys.argv) > 1 else "2,4,6".split(","))]
for mode in modes:
    _lib.lib.cgat_set_bilinear_mode(mode)
    for _ in range(6):
        _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), None, W, out.data_ptr(), W, rows, W, W, W, ws.data_ptr(), ws.numel(), None), "rows")
        _lib.check(_lib.lib.cgat_bilinear_dual(p.data_ptr(), W, q.data_ptr(), W, z.data_ptr(), W, T.data_ptr(), None, W, o1.data_ptr(), W, None, W, o2.data_ptr(), W, rows, ws.data_ptr(), ws.numel(), None), "dual")
    torch.cuda.synchronize()
print("done")
