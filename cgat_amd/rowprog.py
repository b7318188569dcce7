"""Small-row dense networks as ONE launch per direction (csrc/rowprog.hip, C ABI cgat_rowprog_run).

At the batch size the reference harness ships (--batch-size 64, lightning_module.py:468-473) the output head
(ResidualNetwork, message_changed.py:81-138), Roost's gate / message networks (roost_message.py:137-153, 324-355) and
the per-crystal networks run over 64 ... 2 048 rows: a few hundred kFLOP per layer, bound by kernel boundaries.  A network
(or several networks that read the same rows) becomes a *program* of products that one persistent kernel walks phase by
phase -- forward: one phase per layer (the layer's residual product rides in the same op); backward: one phase per layer
holding its input gradient, weight gradient(s) and bias gradient (the activation derivative is applied to the operand as
it is loaded, no pre-activation gradient is ever stored).

Exact fp32 products (f32-input matrix instructions), fixed summation order: bitwise reproducible.
"""
import ctypes as C
import os

import torch

from . import _lib, debug
from ._lib import lib, check

MAX_ROWS = int(os.environ.get("CGAT_ROWPROG_PY_MAX_ROWS", os.environ.get("CGAT_ROWPROG_MAX_ROWS", "2048")))    # 0: never (every dense layer on the generic engine)

_sync = {}


def _sync_words(device):
    """The launch's barrier counters: zero-filled once per device, then owned by the library (include/cgat_hip.h)."""
    key = device.index if device.index is not None else torch.cuda.current_device()
    t = _sync.get(key)
    if t is None:
        if torch.cuda.is_current_stream_capturing():
            raise RuntimeError("cgat_amd.rowprog: the first small-row program of a process must run outside a stream "
                               "capture (GraphedStep's warm-up steps do)")
        t = torch.zeros(_lib.ROWPROG_SYNC_WORDS, dtype=torch.int32, device=device)
        torch.cuda.synchronize(device)           # the fill is complete before any stream uses the counters
        _sync[key] = t
    return t


def barrier_timeouts(device=None):
    """1 if a grid barrier of a small-row program ever gave up on this device (its results are void), else 0.
    Synchronises; test instrumentation."""
    dev = torch.device("cuda", torch.cuda.current_device()) if device is None else torch.device(device)
    t = _sync.get(dev.index if dev.index is not None else torch.cuda.current_device())
    if t is None:
        return 0
    torch.cuda.synchronize(dev)
    return int(t[_lib.ROWPROG_SYNC_WORDS - 16])


def _st(t, transposed=False):
    """(row stride, k stride) of a 2-D operand X(r, k); transposed: X(r, k) = t[k, r]."""
    return (t.stride(1), t.stride(0)) if transposed else (t.stride(0), t.stride(1))


def op(phase, M, N, K, A, B0, out, *, a_t=False, b0_t=False, dact=None, dact_type=0, B1=None, b1_t=False, bias=None,
       act=0, resid=None, accumulate=False, h_out=None, rowsum=None, alpha=1.0):
    """One product of a program (cgat_rowprog_op); every operand a 2-D fp32 tensor view, `x_t`: use it transposed."""
    o = _lib.RowProgOp()
    o.phase, o.M, o.N, o.K = phase, M, N, K
    o.A = A.data_ptr(); o.a_rs, o.a_ks = _st(A, a_t)
    if dact is not None:
        o.dact = dact.data_ptr(); o.d_rs, o.d_ks = _st(dact, a_t); o.dact_type = dact_type
    o.B0 = B0.data_ptr(); o.b0_rs, o.b0_ks = _st(B0, b0_t)
    if B1 is not None:
        o.B1 = B1.data_ptr(); o.b1_rs, o.b1_ks = _st(B1, b1_t)
    if bias is not None:
        o.bias = bias.data_ptr()
    o.act = act
    if resid is not None:
        o.resid = resid.data_ptr(); o.ld_resid = resid.stride(0)
    o.out = out.data_ptr(); o.ldo = out.stride(0); o.alpha, o.beta = alpha, (1.0 if accumulate else 0.0)
    if h_out is not None:
        o.h_out = h_out.data_ptr(); o.ld_h = h_out.stride(0)
    if rowsum is not None:
        o.rowsum = rowsum.data_ptr()
    o._keep = (A, B0, out, dact, B1, bias, resid, h_out, rowsum)
    return o


def run(ops, device):
    """Run the ops (sorted by phase) as one launch -- or several when there are more than 24, cut at phase boundaries."""
    sync = _sync_words(device)
    stream = C.c_void_p(torch.cuda.current_stream(device).cuda_stream)
    i = 0
    with torch.cuda.device(device):
        while i < len(ops):
            j = min(len(ops), i + _lib.ROWPROG_MAX_OPS)
            if j < len(ops):
                while j > i + 1 and ops[j].phase == ops[j - 1].phase:      # never cut inside a phase
                    j -= 1
                if ops[j].phase == ops[j - 1].phase:
                    raise ValueError("cgat_amd.rowprog: a phase with more than 24 ops")
            prog = _lib.RowProg()
            prog.n_ops = j - i
            base = ops[i].phase
            for k in range(i, j):
                prog.op[k - i] = ops[k]
                prog.op[k - i].phase = ops[k].phase - base
            check(lib.cgat_rowprog_run(C.byref(prog), C.c_void_p(sync.data_ptr()), stream), "cgat_rowprog_run")
            i = j


def eligible(x):
    return (MAX_ROWS > 0 and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and 0 < x.shape[0] <= MAX_ROWS and
            x.shape[1] > 0)


def _rows2d(t):
    """A 2-D fp32 view the kernel can address: any strides, as long as they are element strides of one storage."""
    if t.dtype != torch.float32:
        raise TypeError(f"cgat_amd: expected float32, got {t.dtype}")
    return t


class RowNetsFn(torch.autograd.Function):
    """Several dense networks on the SAME input rows, each a stack of layers
         h_l = act_l(x_l W_l^T + b_l),    x_(l+1) = h_l + skip_l(x_l),    skip in {none, identity, x_l R_l^T}
    (SimpleNetwork: LeakyReLU, no skips -- message_changed.py:36-63; ResidualNetwork: ReLU with identity / bias-free
    linear skips, then fc_out -- message_changed.py:86-135), forward in one launch, backward in one launch.

    apply(x, final_resid, spec, *params): spec = tuple over networks of tuples over layers of (act, skip, has_bias);
    params = per network, per layer: W, b (or None), R (or None).  final_resid (or None) is added to the FIRST network's
    output (CGAtNet's `edge_attr + Edge(...)`, CGAT.py:582).  Returns one output per network."""

    @staticmethod
    def forward(ctx, x, final_resid, spec, *params):
        dev = x.device
        x = _rows2d(x)
        M = x.shape[0]
        nets = []
        it = iter(params)
        for net_spec in spec:
            layers = []
            for (act, skip, has_b) in net_spec:
                W, b, R = next(it), next(it), next(it)
                W2 = W.detach().reshape(W.shape[0], -1)
                layers.append(dict(act=act, skip=skip, W=W2, Wp=W, b=None if b is None else b.detach(),
                                   R=None if R is None else R.detach().reshape(R.shape[0], -1)))
            nets.append(layers)
        ops = []
        outs = []
        for ni, layers in enumerate(nets):
            cur = x
            for l, L in enumerate(layers):
                N, K = L["W"].shape
                if cur.shape[1] != K:
                    raise ValueError(f"cgat_amd.rowprog: layer {l} expects {K} inputs, got {cur.shape[1]}")
                out = torch.empty(M, N, dtype=torch.float32, device=dev)
                sep_h = L["skip"] != 0 and L["act"] != _lib.ACT_NONE     # the activation value is not the layer's output
                h = torch.empty(M, N, dtype=torch.float32, device=dev) if sep_h else None
                resid = cur if L["skip"] == 1 else None
                if L["skip"] == 1 and N != K:
                    raise ValueError("cgat_amd.rowprog: identity skip needs equal widths")
                if l == len(layers) - 1 and ni == 0 and final_resid is not None:
                    if resid is not None:
                        raise ValueError("cgat_amd.rowprog: final residual on a layer with an identity skip")
                    resid = final_resid
                ops.append(op(l, M, N, K, cur, L["W"], out, B1=L["R"] if L["skip"] == 2 else None, bias=L["b"],
                              act=L["act"], resid=resid, h_out=h))
                L["x"], L["h"] = cur, (h if sep_h else out)
                cur = out
            outs.append(cur)
        ops.sort(key=lambda o: o.phase)
        run(ops, dev)
        if debug.recording():
            for layers in nets:
                for L in layers:
                    if L["act"] in (_lib.ACT_LEAKY, _lib.ACT_RELU):
                        debug.note(L["Wp"], L["h"] > 0)
        ctx.spec, ctx.has_fr = spec, final_resid is not None
        saved = [x]
        for layers in nets:
            for L in layers:
                saved += [L["W"], L["R"], L["x"] if L["x"] is not x else None, L["h"]]
        ctx.save_for_backward(*saved)
        ctx.pshapes = [None if p is None else p.shape for p in params]
        return tuple(outs)

    @staticmethod
    def backward(ctx, *g_outs):
        x, *rest = ctx.saved_tensors
        dev = x.device
        M = x.shape[0]
        spec = ctx.spec
        it = iter(rest)
        nets = []
        for net_spec in spec:
            layers = []
            for (act, skip, has_b) in net_spec:
                W, R, xin, h = next(it), next(it), next(it), next(it)
                layers.append(dict(act=act, skip=skip, has_b=has_b, W=W, R=R, x=x if xin is None else xin, h=h))
            nets.append(layers)
        need_x = ctx.needs_input_grad[0]
        ops = []
        g_x = torch.empty_like(x) if need_x else None
        grads = []
        depth = max(len(layers) for layers in nets)
        first_written = False
        g_final = None
        for ni, layers in enumerate(nets):
            g = g_outs[ni]
            if g is None:
                g = torch.zeros(M, layers[-1]["W"].shape[0], dtype=torch.float32, device=dev)
            g = g if (g.dtype == torch.float32 and g.dim() == 2) else g.reshape(M, -1).float()
            if ni == 0:
                g_final = g
            net_grads = []
            L_n = len(layers)
            for l in range(L_n - 1, -1, -1):
                L = layers[l]
                N, K = L["W"].shape
                ph = L_n - 1 - l
                dact = L["h"] if L["act"] != _lib.ACT_NONE else None
                gW = torch.empty(N, K, dtype=torch.float32, device=dev)
                gb = torch.empty(N, dtype=torch.float32, device=dev) if L["has_b"] else None
                # dW[n,k] = sum_m gpre[m,n] x[m,k]; db[n] = sum_m gpre[m,n]
                ops.append(op(ph, N, K, M, g, L["x"], gW, a_t=True, b0_t=True, dact=dact, dact_type=L["act"], rowsum=gb))
                gR = None
                if L["skip"] == 2:
                    gR = torch.empty(N, K, dtype=torch.float32, device=dev)
                    ops.append(op(ph, N, K, M, g, L["x"], gR, a_t=True, b0_t=True))
                net_grads.append((gW, gb, gR))
                if l > 0 or need_x:
                    if l > 0:
                        gx = torch.empty(M, K, dtype=torch.float32, device=dev)
                        acc, phx = False, ph
                    else:
                        # the networks share x: the first one writes its gradient, the others add theirs one phase later each
                        gx, acc = g_x, first_written
                        phx = max(ph, depth - 1) + (ni if first_written else 0)
                        first_written = True
                    # dx[m,k] = sum_n gpre[m,n] W[n,k] (+ sum_n g[m,n] R[n,k] | + g[m,k])
                    ops.append(op(phx, M, K, N, g, L["W"], gx, b0_t=True, dact=dact, dact_type=L["act"],
                                  B1=L["R"] if L["skip"] == 2 else None, b1_t=True,
                                  resid=g if L["skip"] == 1 else None, accumulate=acc))
                    g = gx
            grads.append(net_grads[::-1])
        ops.sort(key=lambda o: o.phase)
        # phases must be consecutive from 0
        remap = {p: i for i, p in enumerate(sorted({o.phase for o in ops}))}
        for o in ops:
            o.phase = remap[o.phase]
        run(ops, dev)
        flat = []
        k = 0
        for ni, net_spec in enumerate(spec):
            for l, _ in enumerate(net_spec):
                gW, gb, gR = grads[ni][l]
                shapes = ctx.pshapes[k:k + 3]
                flat += [gW.reshape(shapes[0]), gb, None if gR is None else gR.reshape(shapes[2])]
                k += 3
        return (g_x, g_final if ctx.has_fr else None, None, *flat)


def mlp_spec(n_hidden, act, has_bias=True):
    """SimpleNetwork: n_hidden activated layers, one linear output layer, no skips."""
    return tuple((act, 0, has_bias) for _ in range(n_hidden)) + ((_lib.ACT_NONE, 0, has_bias),)


class RowMultiHeadFn(torch.autograd.Function):
    """MultiHeadNetwork (CGAT.py:65-112: H independent D -> Hd -> O networks with LeakyReLU(0.01), stored as grouped 1x1
    convolutions) on a few hundred rows: hid = leaky(fea W_in^T + b_in) for all heads as one product, then every head's
    second layer on its column block -- two phases of ONE launch; the backward likewise (second layers' weight and input
    gradients, then the first layer's with the activation derivative applied to the operand as it is loaded)."""

    @staticmethod
    def forward(ctx, fea, w_in, b_in, w_out, b_out, H, Hd, O):
        dev = fea.device
        M, D = fea.shape
        Wi = w_in.detach().reshape(H * Hd, -1)
        if Wi.shape[1] != D:
            raise ValueError(f"cgat_amd.rowprog: MultiHeadNetwork expects {Wi.shape[1]} inputs, got {D}")
        Wo = w_out.detach().reshape(H * O, Hd)
        bi = None if b_in is None else b_in.detach()
        bo = None if b_out is None else b_out.detach()
        hid = torch.empty(M, H * Hd, dtype=torch.float32, device=dev)
        out = torch.empty(M, H * O, dtype=torch.float32, device=dev)
        ops = [op(0, M, H * Hd, D, fea, Wi, hid, bias=bi, act=_lib.ACT_LEAKY)]
        for h in range(H):
            ops.append(op(1, M, O, Hd, hid[:, h * Hd:(h + 1) * Hd], Wo[h * O:(h + 1) * O], out[:, h * O:(h + 1) * O],
                          bias=None if bo is None else bo[h * O:(h + 1) * O]))
        run(ops, dev)
        if debug.recording():
            debug.note(w_in, hid > 0)
        ctx.dims = (H, Hd, O)
        ctx.shapes = (w_in.shape, w_out.shape, b_in is not None, b_out is not None)
        ctx.save_for_backward(fea, Wi, Wo, hid)
        return out.reshape(M, H, O)

    @staticmethod
    def backward(ctx, g):
        fea, Wi, Wo, hid = ctx.saved_tensors
        H, Hd, O = ctx.dims
        dev = fea.device
        M, D = fea.shape
        g = g.reshape(M, H * O)
        if g.dtype != torch.float32:
            g = g.float()
        g_hid = torch.empty(M, H * Hd, dtype=torch.float32, device=dev)
        g_wo = torch.empty(H * O, Hd, dtype=torch.float32, device=dev)
        g_bo = torch.empty(H * O, dtype=torch.float32, device=dev) if ctx.shapes[3] else None
        g_wi = torch.empty(H * Hd, D, dtype=torch.float32, device=dev)
        g_bi = torch.empty(H * Hd, dtype=torch.float32, device=dev) if ctx.shapes[2] else None
        need_x = ctx.needs_input_grad[0]
        g_fea = torch.empty(M, D, dtype=torch.float32, device=dev) if need_x else None
        ops = []
        for h in range(H):
            gh, hh = g[:, h * O:(h + 1) * O], hid[:, h * Hd:(h + 1) * Hd]
            ops.append(op(0, O, Hd, M, gh, hh, g_wo[h * O:(h + 1) * O], a_t=True, b0_t=True,
                          rowsum=None if g_bo is None else g_bo[h * O:(h + 1) * O]))
            ops.append(op(0, M, Hd, O, gh, Wo[h * O:(h + 1) * O], g_hid[:, h * Hd:(h + 1) * Hd], b0_t=True))
        ops.append(op(1, H * Hd, D, M, g_hid, fea, g_wi, a_t=True, b0_t=True, dact=hid, dact_type=_lib.ACT_LEAKY, rowsum=g_bi))
        if need_x:
            ops.append(op(1, M, D, H * Hd, g_hid, Wi, g_fea, b0_t=True, dact=hid, dact_type=_lib.ACT_LEAKY))
        run(ops, dev)
        return (g_fea, g_wi.reshape(ctx.shapes[0]), g_bi, g_wo.reshape(ctx.shapes[1]), g_bo, None, None, None)
