"""Dev tool: the hypernetwork weight gradient out[a,b,c] = sum_n p[n,a] q[n,b] r[n,c] in every arithmetic mode:
error vs fp64 over ALL rows (rms of err / sum|terms| and max-norm relative), kernel time from the library's launch
timers (tag bilinear_wgrad) and wall time incl. operand preparation.  `python tools/wgrad_probe.py [rows ...]`"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from cgat_amd import _lib, ops

dev = torch.device("cuda:0")
W = 128
MODES = {"bf16x6": 6, "f16x3": 2, "f16x3c": 4}


def ref64(p, q, r):
    out = torch.zeros(W, W, W, dtype=torch.float64, device=dev)
    den = torch.zeros(W, W, W, dtype=torch.float64, device=dev)
    for i in range(0, p.shape[0], 4096):
        pp, qq, rr = (t[i:i + 4096].double() for t in (p, q, r))
        out += torch.einsum("na,nb,nc->abc", pp, qq, rr)
        den += torch.einsum("na,nb,nc->abc", pp.abs(), qq.abs(), rr.abs())
    return out, den


def run(rows, cases):
    g = torch.Generator().manual_seed(rows)
    p, q, r = (torch.randn(rows, W, generator=g).to(dev) for _ in range(3))
    if "wide" in cases:     # wide dynamic range inside every operand: most products far below the tensor maximum
        sc = lambda: torch.pow(10.0, -3 * torch.rand(rows, W, generator=g)).to(dev)
        p, q, r = p * sc(), q * sc(), r * sc()
    ref, den = ref64(p, q, r)
    out = torch.empty(W, W, W, device=dev)
    ws = torch.empty(max(_lib.lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), 256), dtype=torch.uint8, device=dev)

    def call():
        _lib.check(_lib.lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, out.data_ptr(), rows, W, W, W,
                                                ws.data_ptr(), ws.numel(), None), "wgrad")
    for name, m in MODES.items():
        _lib.lib.cgat_set_bilinear_mode(m)
        out.fill_(float("nan"))
        call(); torch.cuda.synchronize()
        o = out.double()
        mx = float((o - ref).abs().max() / ref.abs().max())
        rms = float(((o - ref) / den).square().mean().sqrt())
        a = out.clone(); call(); torch.cuda.synchronize()
        same = bool(torch.equal(a, out))
        ops.prof_enable(True); ops.prof_reset()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5): call()
        e1.record(); torch.cuda.synchronize()
        n, ms = ops.prof_get("bilinear_wgrad")
        ops.prof_enable(False)
        print(f"rows {rows:6d} {cases:5s} {name:7s}: max-norm rel {mx:.3e}  rms err/sum|terms| {rms:.3e}  repeatable {same}  "
              f"kernel {ms / max(n, 1):.3f} ms  wall {e0.elapsed_time(e1) / 5:.3f} ms", flush=True)
    _lib.lib.cgat_set_bilinear_mode(4)


if __name__ == "__main__":
    sizes = [int(a) for a in sys.argv[1:]] or [33, 1000, 20001, 83340]
    for rows in sizes:
        run(rows, "unit")
    run(sizes[-1], "wide")
