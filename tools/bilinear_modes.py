"""Dev tool: accuracy (vs fp64 on a row sample) and interleaved timing of the bilinear_rows arithmetic modes."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib
dev = torch.device("cuda:0")
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q = torch.randn(rows, W, generator=g).to(dev), torch.randn(rows, W, generator=g).to(dev)
T = (torch.randn(W, W, W, generator=g) / W).to(dev)
init = torch.randn(rows, W, generator=g).to(dev)
out = torch.empty(rows, W, device=dev)
ws = torch.empty(_lib.lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
sel = torch.cat([torch.arange(160), torch.arange(rows - 160, rows), torch.randint(0, rows, (320,), generator=g)])
ref = torch.einsum("na,nb,abc->nc", p[sel].double().cpu(), q[sel].double().cpu(), T.double().cpu()) + init[sel].double().cpu()
def call():
    _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), init.data_ptr(), W, out.data_ptr(),
                                           W, rows, W, W, W, ws.data_ptr(), ws.numel(), None), "bilinear_rows")
res = {}
for rnd in range(5):
    for mode in (0, 6, 3):
        _lib.lib.cgat_set_bilinear_mode(mode)
        call(); torch.cuda.synchronize()
        if rnd == 0:
            err = float((out[sel].double().cpu() - ref).abs().max() / ref.abs().max())
            print(f"mode {mode}: max-norm rel err vs fp64 = {err:.3e}  finite={bool(torch.isfinite(out).all())}")
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): call()
        e1.record(); torch.cuda.synchronize()
        res.setdefault(mode, []).append(e0.elapsed_time(e1) / 4)
fl = 2.0 * rows * W ** 3
for mode, t in res.items():
    t = sorted(t); med = t[len(t) // 2]
    print(f"mode {mode}: median {med:.3f} ms incl. T preparation + slab sum -> {fl / med / 1e9:.1f} TFLOP/s (fp32-equivalent)")
