"""Dev tool: timing-only ablations of the forward contraction kernel (CGAT_RING_ABL bits: 1 no barrier, 2 no global
loads, 4 no LDS fragment reads, 8 no flush; results are wrong by construction) -- what the matrix-core skeleton costs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib, ops
dev = torch.device("cuda:0")
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q = torch.randn(rows, W, generator=g).to(dev), torch.randn(rows, W, generator=g).to(dev)
T = (torch.randn(W, W, W, generator=g) / W).to(dev)
out = torch.empty(rows, W, device=dev)
ws = torch.empty(_lib.lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
def call():
    _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), None, W, out.data_ptr(), W, rows, W, W, W,
                                           ws.data_ptr(), ws.numel(), None), "rows")
for abl in ("", "0", "1", "2", "3"):
    os.environ.pop("CGAT_RING_ABL", None)
    if abl: os.environ["CGAT_RING_ABL"] = abl
    ops.prof_reset(); ops.prof_enable(True)
    for _ in range(12): call()
    torch.cuda.synchronize(); ops.prof_enable(False)
    n, ms = ops.prof_get("bilinear_rows")
    print(f"ABL {abl or '-':>3s}: {ms / n * 1e3:7.1f} us per launch (kernel only)")
