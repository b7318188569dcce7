"""Operand-width discriminator cases shared by tests/test_hip_kernels.py::test_operand_width_discriminator and
tools/arith_discriminator.py.  Operands whose blocks span 2^12: two entries per row at the block maximum that CANCEL to the
level of the many small entries (2^-8 .. 2^-12 of the maximum), so that what is left of the sum is carried by the small
terms while its error is carried by how many bits the LARGE operands kept.  Error vs fp64 as rms of err / sum|terms|."""
import torch

W = 128
MODES = ("f32", "bf16x6", "f16x3", "f16x3c")

def make_case(rows, seed):
    """q[n]: small entries everywhere, two big ones (+g, -g(1 - d)) at columns b0(n), b1(n); T[a, b1, c] = T[a, b0, c] for the
    pairs used, so that q T cancels to the d-level; p dense gaussian."""
    g = torch.Generator().manual_seed(seed)
    p = torch.randn(rows, W, generator=g)
    small = torch.randn(rows, W, generator=g) * torch.pow(2.0, -8 - 4 * torch.rand(rows, W, generator=g))
    q = small.clone()
    big = 1.0 + torch.rand(rows, generator=g)                      # block maximum of the row, in [1, 2)
    dlt = torch.pow(2.0, -9 - 2 * torch.rand(rows, generator=g))   # the pair cancels to 2^-9 .. 2^-11 of it
    q[:, 0] = big
    q[:, 1] = -big * (1 - dlt)
    T = torch.randn(W, W, W, generator=g) / W
    T[:, 1, :] = T[:, 0, :]
    return p, q, T


def rms_rel(out, ref, den):
    return float(((out.double() - ref) / den).square().mean().sqrt())


def run_rows(_lib, ops, dev, rows=4096, seed=5):
    p, q, T = (t.to(dev) for t in make_case(rows, seed))
    ref = torch.einsum("na,nb,abc->nc", p.double(), q.double(), T.double())
    den = torch.einsum("na,nb,abc->nc", p.double().abs(), q.double().abs(), T.double().abs())
    out = torch.empty(rows, W, device=dev)
    res = {}
    for mode in MODES:
        ops.set_bilinear_mode(mode)
        ws = torch.empty(max(_lib.lib.cgat_bilinear_rows_workspace_bytes(rows, W, W, W), 256), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.cgat_bilinear_rows(p.data_ptr(), W, q.data_ptr(), W, T.data_ptr(), None, W, out.data_ptr(), W, rows,
                                               W, W, W, ws.data_ptr(), ws.numel(), None), "bilinear_rows")
        torch.cuda.synchronize()
        res[mode] = rms_rel(out, ref, den)
    ops.set_bilinear_mode(ops.DEFAULT_MODE)
    return res


def run_dual(_lib, ops, dev, rows=4096, seed=6):
    p, q, T = (t.to(dev) for t in make_case(rows, seed))
    g = torch.Generator().manual_seed(seed + 1)
    z = torch.randn(rows, W, generator=g).to(dev)
    M = torch.einsum("nb,abc->nac", q.double(), T.double())
    Ma = torch.einsum("nb,abc->nac", q.double().abs(), T.double().abs())
    r1, d1 = torch.einsum("na,nac->nc", p.double(), M), torch.einsum("na,nac->nc", p.double().abs(), Ma)
    r2, d2 = torch.einsum("nc,nac->na", z.double(), M), torch.einsum("nc,nac->na", z.double().abs(), Ma)
    o1, o2 = torch.empty(rows, W, device=dev), torch.empty(rows, W, device=dev)
    res = {}
    for mode in MODES:
        ops.set_bilinear_mode(mode)
        ws = torch.empty(max(_lib.lib.cgat_bilinear_dual_workspace_bytes(rows), 256), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.cgat_bilinear_dual(p.data_ptr(), W, q.data_ptr(), W, z.data_ptr(), W, T.data_ptr(), None, W,
                                               o1.data_ptr(), W, None, W, o2.data_ptr(), W, rows, ws.data_ptr(), ws.numel(),
                                               None), "dual")
        torch.cuda.synchronize()
        res[mode] = (rms_rel(o1, r1, d1), rms_rel(o2, r2, d2))
    ops.set_bilinear_mode(ops.DEFAULT_MODE)
    return res


def make_wgrad_case(rows, seed):
    """r[n]: small entries, p q products: q has the cancelling pair over ROWS: rows come in pairs (2m, 2m + 1) with
    q[2m + 1] = -q[2m] (1 - d) on the big column and r[2m + 1] = r[2m], p[2m + 1] = p[2m]: the sum over n cancels pairwise."""
    g = torch.Generator().manual_seed(seed)
    half = rows // 2
    p = torch.randn(half, W, generator=g).repeat_interleave(2, 0)
    r = torch.randn(half, W, generator=g).repeat_interleave(2, 0)
    q = torch.randn(rows, W, generator=g) * torch.pow(2.0, -8 - 4 * torch.rand(rows, W, generator=g))
    big = (1.0 + torch.rand(half, generator=g))
    dlt = torch.pow(2.0, -9 - 2 * torch.rand(half, generator=g))
    q[0::2, 0] = big
    q[1::2, 0] = -big * (1 - dlt)
    return p, q, r


def run_wgrad(_lib, ops, dev, rows=4096, seed=7):
    p, q, r = (t.to(dev) for t in make_wgrad_case(rows, seed))
    ref = torch.einsum("na,nb,nc->abc", p.double(), q.double(), r.double())
    den = torch.einsum("na,nb,nc->abc", p.double().abs(), q.double().abs(), r.double().abs())
    out = torch.empty(W, W, W, device=dev)
    res = {}
    for mode in MODES:
        ops.set_bilinear_mode(mode)
        ws = torch.empty(max(_lib.lib.cgat_bilinear_wgrad_workspace_bytes(rows, W, W, W), 256), dtype=torch.uint8, device=dev)
        _lib.check(_lib.lib.cgat_bilinear_wgrad(p.data_ptr(), W, q.data_ptr(), W, r.data_ptr(), W, out.data_ptr(), rows, W, W, W,
                                                ws.data_ptr(), ws.numel(), None), "wgrad")
        torch.cuda.synchronize()
        # the big column b = 0 is where the cancellation happens
        res[mode] = rms_rel(out[:, 0, :], ref[:, 0, :], den[:, 0, :])
    ops.set_bilinear_mode(ops.DEFAULT_MODE)
    return res


