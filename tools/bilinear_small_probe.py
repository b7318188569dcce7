"""Dev tool: error of the generic-width bilinear contractions (outer-product operands of the engine) against fp64."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from cgat_amd import _lib  # noqa: E402

dev = torch.device("cuda:0")
lib = _lib.lib
print("CGAT_GEMM_SPLIT =", os.environ.get("CGAT_GEMM_SPLIT", "(default: on)"))
for W, rows in [(16, 80), (16, 960), (22, 130), (64, 1000), (96, 333)]:
    g = torch.Generator().manual_seed(W + rows)
    p, q, r = (torch.randn(rows, W, generator=g).to(dev) for _ in range(3))
    T = (torch.randn(W, W, W, generator=g) / W).to(dev)
    out = torch.full((rows, W), float("nan"), device=dev)
    ws = torch.empty(max(lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), 256), dtype=torch.uint8, device=dev)
    _lib.check(lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), None, W, out.data_ptr(), W, rows, W, W, W,
                                      ws.data_ptr(), ws.numel(), None), "rows")
    torch.cuda.synchronize()
    ref = torch.einsum("na,nb,abc->nc", p.double(), q.double(), T.double())
    e1 = float((out.double() - ref).abs().max() / ref.abs().max())
    o2 = torch.full((W, W, W), float("nan"), device=dev)
    ws = torch.empty(max(lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), 256), dtype=torch.uint8, device=dev)
    _lib.check(lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, o2.data_ptr(), rows, W, W, W,
                                       ws.data_ptr(), ws.numel(), None), "wgrad")
    torch.cuda.synchronize()
    ref2 = torch.einsum("na,nb,nc->abc", p.double(), q.double(), r.double())
    e2 = float((o2.double() - ref2).abs().max() / ref2.abs().max())
    print(f"W={W:3d} rows={rows:5d}  rows-form err {e1:.2e}   wgrad err {e2:.2e}", flush=True)
