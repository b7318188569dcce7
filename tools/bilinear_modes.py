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
    for mode in (0, 6, 3, 2, 4):
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

# the fused pair of gradients (bilinear_dual) in the same modes, plus a row-scale stress for the fp16 form
z = torch.randn(rows, W, generator=g).to(dev)
o1, o2 = torch.empty(rows, W, device=dev), torch.empty(rows, W, device=dev)
ws2 = torch.empty(_lib.lib.cgat_bilinear_dual_workspace_bytes(rows), dtype=torch.uint8, device=dev)
M = torch.einsum("nb,abc->nac", q[sel].double().cpu(), T.double().cpu())
r1 = torch.einsum("na,nac->nc", p[sel].double().cpu(), M); r2 = torch.einsum("nc,nac->na", z[sel].double().cpu(), M)
def dual():
    _lib.check(_lib.lib.cgat_bilinear_dual(p.data_ptr(), W, q.data_ptr(), W, z.data_ptr(), W, T.data_ptr(), None, W, o1.data_ptr(), W,
                                           None, W, o2.data_ptr(), W, rows, ws2.data_ptr(), ws2.numel(), None), "dual")
for mode in (6, 3, 2, 4):
    _lib.lib.cgat_set_bilinear_mode(mode)
    dual(); torch.cuda.synchronize()
    e1_ = float((o1[sel].double().cpu() - r1).abs().max() / r1.abs().max()); e2_ = float((o2[sel].double().cpu() - r2).abs().max() / r2.abs().max())
    ts = []
    for _ in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(4): dual()
        e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
    print(f"dual mode {mode}: err {e1_:.3e} / {e2_:.3e}   median {sorted(ts)[2]:.3f} ms")
# rows whose magnitudes span 1e-6 .. 1e4 and a T of magnitude 1e-3: relative error PER ROW
scale = torch.logspace(-6, 4, rows).to(dev)[:, None]
qs = q * scale
Ts = T * 1e-3
refs = torch.einsum("na,nb,abc->nc", p[sel].double().cpu(), qs[sel].double().cpu(), Ts.double().cpu())
for mode in (6, 2, 4):
    _lib.lib.cgat_set_bilinear_mode(mode)
    _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, qs.data_ptr(), W, Ts.data_ptr(), None, W, out.data_ptr(),
                                           W, rows, W, W, W, ws.data_ptr(), ws.numel(), None), "bilinear_rows")
    torch.cuda.synchronize()
    d = (out[sel].double().cpu() - refs).abs().amax(1) / refs.abs().amax(1)
    print(f"row-scaled q, mode {mode}: worst per-row max-norm rel err {float(d.max()):.3e}")
# weight gradient in the same modes (+ operands of very different magnitudes)
wout = torch.empty(W, W, W, device=dev)
ws3 = torch.empty(_lib.lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), dtype=torch.uint8, device=dev)
for tag, (pp, qq, rr) in {"unit": (p, q, z), "scaled": (p * 3e-4, q * 2e3, z * 1e-5)}.items():
    refw = torch.einsum("na,nb,nc->abc", pp[:20000].double(), qq[:20000].double(), rr[:20000].double()).cpu()
    for mode in (6, 3, 2, 4):
        _lib.lib.cgat_set_bilinear_mode(mode)
        def wg(n=rows):
            _lib.check(_lib.lib.cgat_bilinear_wgrad(pp.data_ptr(), W, qq.data_ptr(), W, rr.data_ptr(), W, wout.data_ptr(), n, W, W, W,
                                                    ws3.data_ptr(), ws3.numel(), None), "wgrad")
        wg(20000); torch.cuda.synchronize()
        err = float((wout.double().cpu() - refw).abs().max() / refw.abs().max())
        ts = []
        for _ in range(5):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(4): wg()
            e1.record(); torch.cuda.synchronize(); ts.append(e0.elapsed_time(e1) / 4)
        print(f"wgrad {tag} mode {mode}: err(20000 rows) {err:.3e}   median {sorted(ts)[2]:.3f} ms (83340 rows, incl. pre-passes)")
_lib.lib.cgat_set_bilinear_mode(6)
