"""Dev tool: interleaved A/B timing of bilinear_rows kernel variants in ONE process (guide rule 24)."""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib
dev = torch.device("cuda:0")
rows, W = 83340, 128
g = torch.Generator().manual_seed(0)
p, q = torch.randn(rows, W, generator=g).to(dev), torch.randn(rows, W, generator=g).to(dev)
T = (torch.randn(W, W, W, generator=g) / W).to(dev)
out = torch.empty(rows, W, device=dev)
nb = _lib.lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W)
ws = torch.empty(nb, dtype=torch.uint8, device=dev)
variants = [int(v) for v in (sys.argv[1:] or ["161", "162", "164", "321", "322", "324"])]
def call():
    _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), None, W, out.data_ptr(), W, rows,
                                           W, W, W, ws.data_ptr(), ws.numel(), None), "bilinear_rows")
res = {v: [] for v in variants}
ref = None
for rnd in range(6):
    for v in variants:
        os.environ["CGAT_BIL_VARIANT_LIVE"] = str(v)
        call(); torch.cuda.synchronize()
        if rnd == 0:
            o = out.clone()
            if ref is None: ref = o
            else: print(v, "max rel diff vs first variant", float((o - ref).abs().max() / ref.abs().max()))
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): call()
        e1.record(); torch.cuda.synchronize()
        res[v].append(e0.elapsed_time(e1) / 4)
fl = 2.0 * rows * W ** 3
for v in variants:
    t = sorted(res[v]); med = t[len(t) // 2]
    print(f"variant {v}: median {med:.3f} ms (min {t[0]:.3f})  incl. T re-layout+slab sum;  {fl / med / 1e9:.1f} TFLOP/s")
