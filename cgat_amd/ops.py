"""torch.autograd wrappers over the C ABI.  PyTorch supplies device memory, the current
stream and the autograd tape; all arithmetic of the path runs in libcgat_hip kernels.
"""
import ctypes as C

import os

import torch

from . import _lib, debug, rowprog
from ._lib import lib, check


# ----------------------------------------------------------------------------------------
# plumbing
# ----------------------------------------------------------------------------------------
def _require_gpu(*tensors):
    for t in tensors:
        if t is not None and not t.is_cuda:
            raise RuntimeError("cgat_amd: the HIP path needs tensors on an MI355X device (cuda:N); "
                               "there is no CPU fallback -- use oracle/ only for checking")


def _f32c(t):
    if t.dtype != torch.float32:
        raise TypeError(f"cgat_amd: expected float32, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def _ptr(t):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


_ws = {}


def workspace(nbytes, device):
    """Grow-only per-device scratch buffer.  All launches are ordered on the current stream, so
    one buffer serves every call (no allocation in the steady state)."""
    if torch.cuda.is_current_stream_capturing():
        # inside a hipGraph capture (cgat_amd.GraphedStep) the buffer must belong to THAT graph's memory pool: a cached
        # one would come from an earlier capture on the same capture stream, whose pool dies with its graph.  The pool
        # reuses the block as soon as the tensor is dropped, exactly as the caching allocator does in eager mode.
        return torch.empty(int(nbytes) + 4096, dtype=torch.uint8, device=device)
    key = (device.index, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(int(nbytes * 1.25) + 4096, dtype=torch.uint8, device=device)
        _ws[key] = buf
    return buf


def _scratch(numel, dtype, device):
    """A saved-for-backward / side buffer whose size the library dictates (only its pointer is used): the element count is
    rounded up to 1/16 of its octave.  Ragged batches change E by a few per cent from step to step; with exact sizes
    every new maximum made torch's caching allocator hipMalloc a fresh multi-GB block per layer (two 840-ms steps among
    121-ms ones in the training bench), with rounded sizes the cached blocks fit."""
    n = int(numel)
    if n > (1 << 20):
        step = 1 << (n.bit_length() - 5)
        n = (n + step - 1) // step * step
    return torch.empty(n, dtype=dtype, device=device)


# Destinations for parameter gradients (cgat_amd.dist.GradientAverager registers one): a backward pass that is about to
# allocate the gradient of a parameter asks the sinks first and writes straight into what they hand out -- the
# all-reduce bucket -- instead of into a fresh tensor that autograd then adds to (or the averager copies into) the bucket:
# 38 MB of gradients per layer step made 0.44 ms of fills, adds and copies at world size 1 (rocprofv3, round 6).
_grad_sinks = []


def register_grad_sink(fn):
    """fn(w) -> a writable tensor of w's shape that will BE the parameter's gradient, or None."""
    _grad_sinks.append(fn)
    return fn


def unregister_grad_sink(fn):
    if fn in _grad_sinks:
        _grad_sinks.remove(fn)


def _param_grad(w):
    for sink in _grad_sinks:
        t = sink(w)
        if t is not None:
            return t
    return torch.empty_like(w)


_validate_indices = os.environ.get("CGAT_VALIDATE_INDICES", "1") != "0"


def set_validate_indices(flag):
    """Index validation of the CSR plans (default on): an edge_index / segment index entry outside [0, N) raises
    IndexError, as the reference's index_select / scatter would.  It costs one host sync per plan build; a training
    loop that trusts its collation can switch it off."""
    global _validate_indices
    _validate_indices = bool(flag)


def _check_plan_complete(rowptr, n, what):
    # the plan builder drops out-of-range keys, so the last row pointer falls short of n exactly when one exists
    if _validate_indices and n > 0:
        got = int(rowptr[-1])
        if got != n:
            raise IndexError(f"cgat_amd: {n - got} of {n} entries of {what} are outside [0, {rowptr.numel() - 1})")


class EdgePlan:
    """CSR plan of a batch (C struct cgat_plan): built once per edge_index, shared by all layers,
    forward and backward."""

    def __init__(self, edge_index, num_nodes):
        _require_gpu(edge_index)
        if edge_index.dtype != torch.int64 or edge_index.dim() != 2 or edge_index.shape[0] != 2:
            raise TypeError("edge_index must be int64 [2, E]")
        ei = edge_index.contiguous()
        E, N = int(ei.shape[1]), int(num_nodes)
        dev = ei.device
        i32 = dict(dtype=torch.int32, device=dev)
        self.N, self.E = N, E
        self.dst_rowptr = torch.empty(N + 1, **i32)
        self.dst_perm = torch.empty(E, **i32)
        self.dst_sorted = torch.empty(E, **i32)
        self.src_sorted = torch.empty(E, **i32)
        self.src_rowptr = torch.empty(N + 1, **i32)
        self.src_pos = torch.empty(E, **i32)
        nbytes = lib.cgat_plan_workspace_bytes(E, N)
        ws = workspace(nbytes, dev)
        with torch.cuda.device(dev):
            check(lib.cgat_plan_build(_ptr(ei), E, N, _ptr(self.dst_rowptr), _ptr(self.dst_perm),
                                      _ptr(self.dst_sorted), _ptr(self.src_sorted), _ptr(self.src_rowptr),
                                      _ptr(self.src_pos), _ptr(ws), ws.numel(), _stream()), "cgat_plan_build")
        _check_plan_complete(self.dst_rowptr, E, "edge_index[1]")
        _check_plan_complete(self.src_rowptr, E, "edge_index[0]")
        self.c = _lib.Plan(N, E, self.dst_rowptr.data_ptr(), self.dst_perm.data_ptr(), self.dst_sorted.data_ptr(),
                           self.src_sorted.data_ptr(), self.src_rowptr.data_ptr(), self.src_pos.data_ptr())


_plan_cache = {}


def get_plan(edge_index, num_nodes):
    """Small LRU keyed by the edge_index storage: every layer of a stack (and its backward)
    reuses the same plan."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), int(num_nodes), edge_index._version, edge_index.device.index)
    # a long-lived edge_index object carries its plan itself (the closed chunks of chunked.py: 8+ chunk tensors per 64 M
    # edges would evict each other from the 8-entry LRU and rebuild -- and re-validate, two host syncs -- their plan in
    # every forward, recomputation and backward)
    if hasattr(edge_index, "_cgat_plan"):                 # marked by its owner (chunked.closed_chunks): built on first use
        return attach_plan(edge_index, num_nodes)
    plan = _plan_cache.get(key)
    if plan is None:
        if len(_plan_cache) >= 8:
            _plan_cache.pop(next(iter(_plan_cache)))
        plan = EdgePlan(edge_index, num_nodes)
        plan._keepalive = edge_index
        _plan_cache[key] = plan
    return plan


def attach_plan(edge_index, num_nodes):
    """Build (and validate) the plan of `edge_index` once and keep it on the tensor object itself."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), int(num_nodes), edge_index._version, edge_index.device.index)
    own = getattr(edge_index, "_cgat_plan", None)
    if own is None or own[0] != key:
        edge_index._cgat_plan = (key, EdgePlan(edge_index, num_nodes))
    return edge_index._cgat_plan[1]


class SegmentPlan:
    """rowptr/perm for a generic int64 segment index (crystal index, Roost self index)."""

    def __init__(self, index, num_segments):
        _require_gpu(index)
        dev = index.device
        n, S = int(index.numel()), int(num_segments)
        keys = index.to(torch.int32).contiguous()
        self.n, self.S = n, S
        self.rowptr = torch.empty(S + 1, dtype=torch.int32, device=dev)
        self.perm = torch.empty(n, dtype=torch.int32, device=dev)
        ws = workspace(lib.cgat_csr_workspace_bytes(S), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_csr_from_keys(_ptr(keys), n, S, _ptr(self.rowptr), _ptr(self.perm), _ptr(ws), ws.numel(),
                                         _stream()), "cgat_csr_from_keys")
        _check_plan_complete(self.rowptr, n, "the segment index")
        self.perm64 = self.perm.long()


_seg_plan_cache = {}


def get_segment_plan(index, num_segments):
    """SegmentPlan through a small LRU keyed by the index storage (as get_plan): the Roost branch and the crystal
    pooling of one batch build four plans per forward; a batch that is evaluated again (validation epochs, the
    hipGraph-captured step) builds -- and validates, one host synchronisation each -- none."""
    key = (index.data_ptr(), tuple(index.shape), int(num_segments), index._version, index.device.index)
    plan = _seg_plan_cache.get(key)
    if plan is None:
        if len(_seg_plan_cache) >= 16:
            _seg_plan_cache.pop(next(iter(_seg_plan_cache)))
        plan = SegmentPlan(index, num_segments)
        plan._keepalive = index
        _seg_plan_cache[key] = plan
    return plan


# ----------------------------------------------------------------------------------------
# GATConvNodes message + softmax + aggregate
# ----------------------------------------------------------------------------------------
def _attn_params(x, edge_attr, H, ws_):
    A_in_w = ws_[0]
    C_, Ce = x.shape[1], edge_attr.shape[1]
    D = 2 * C_ + Ce
    HHd = A_in_w.shape[0]
    assert A_in_w.numel() == HHd * D and HHd % H == 0
    Hd = HHd // H
    p = _lib.AttnParams(C_, Ce, H, Hd, *[t.data_ptr() for t in ws_])
    return p, Hd


class NodesAttentionFn(torch.autograd.Function):
    """aggr[n] = mean_h sum_{e: dst(e)=n} softmax_dst(MH_A(m_e))[h] * MH_M(m_e)[h]
    (reference CGAT.py:319-329 + PyG aggregate), scalar attention."""

    @staticmethod
    def forward(ctx, x, edge_attr, plan, H, A_in_w, A_in_b, A_out_w, A_out_b, M_in_w, M_in_b, M_out_w, M_out_b):
        weights = [A_in_w, A_in_b, A_out_w, A_out_b, M_in_w, M_in_b, M_out_w, M_out_b]
        _require_gpu(x, edge_attr, *weights)
        x, edge_attr = _f32c(x), _f32c(edge_attr)
        weights = [_f32c(w.detach()) for w in weights]
        if A_out_w.numel() != H * (A_in_w.shape[0] // H):
            raise ValueError("NodesAttentionFn handles scalar attention (MH_A output_dim == 1)")
        N, E = plan.N, plan.E
        if x.shape[0] != N or edge_attr.shape[0] != E:
            raise ValueError(f"plan is for N={N}, E={E}; got x {tuple(x.shape)}, edge_attr {tuple(edge_attr.shape)}")
        p, Hd = _attn_params(x, edge_attr, H, weights)
        dev = x.device
        saved = _scratch(lib.cgat_nodes_attention_saved_floats(N, E, H, Hd), torch.float32, dev)
        aggr = torch.empty(N, x.shape[1], dtype=torch.float32, device=dev)
        ws = workspace(lib.cgat_nodes_attention_forward_workspace_bytes(C.byref(plan.c), C.byref(p)), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_nodes_attention_forward(C.byref(plan.c), C.byref(p), _ptr(x), _ptr(edge_attr), _ptr(aggr),
                                                   _ptr(saved), _ptr(ws), ws.numel(), _stream()),
                  "cgat_nodes_attention_forward")
        ctx.plan, ctx.H, ctx.storage = plan, H, lib.cgat_get_edge_storage()
        ctx.save_for_backward(x, edge_attr, saved, *weights)
        if debug.recording():
            debug.note_attention(A_in_w, M_in_w, plan, p, saved, 2 * H * Hd)
        return aggr

    @staticmethod
    def backward(ctx, g_aggr):
        x, edge_attr, saved, *weights = ctx.saved_tensors
        plan, H = ctx.plan, ctx.H
        g_aggr = _f32c(g_aggr)
        p, Hd = _attn_params(x, edge_attr, H, weights)
        dev = x.device
        g_x = torch.empty_like(x)
        g_e = torch.empty_like(edge_attr)
        grads = [_param_grad(w) for w in weights]
        g = _lib.AttnGrads(*[t.data_ptr() for t in grads])
        ws = workspace(lib.cgat_nodes_attention_backward_workspace_bytes(C.byref(plan.c), C.byref(p)), dev)
        with torch.cuda.device(dev), _storage_of(ctx.storage):
            check(lib.cgat_nodes_attention_backward(C.byref(plan.c), C.byref(p), _ptr(x), _ptr(edge_attr), _ptr(saved),
                                                    _ptr(g_aggr), _ptr(g_x), _ptr(g_e), C.byref(g), _ptr(ws),
                                                    ws.numel(), _stream()), "cgat_nodes_attention_backward")
        return (g_x, g_e, None, None, *grads)


class EdgeHiddenFn(torch.autograd.Function):
    """hidden[t] = LeakyReLU(w_in [x_i ; edge_attr ; x_j] + b_in) in destination-sorted slot order t (plan.dst_perm):
    the first layer of both message networks (reference CGAT.py:96,105-108 on the concatenated message of 316-318)
    with the operand split, for the vector-attention variants whose second layers need every hidden row."""

    @staticmethod
    def forward(ctx, x, edge_attr, plan, w_in, b_in):
        _require_gpu(x, edge_attr, w_in, b_in)
        x, edge_attr, w_in, b_in = _f32c(x), _f32c(edge_attr), _f32c(w_in.detach()), _f32c(b_in.detach())
        N, E = plan.N, plan.E
        Cn, Ce, W2 = x.shape[1], edge_attr.shape[1], w_in.shape[0]
        if x.shape[0] != N or edge_attr.shape[0] != E or w_in.shape[1] != 2 * Cn + Ce or b_in.numel() != W2:
            raise ValueError("EdgeHiddenFn: shapes do not match the plan / the stacked first-layer weight")
        dev = x.device
        hidden = torch.empty(E, W2, dtype=torch.float32, device=dev)
        hmax = torch.empty(1, dtype=torch.float32, device=dev)     # max |hidden|: the fp16 scale of the second layers
        ws = workspace(lib.cgat_edge_hidden_forward_workspace_bytes(C.byref(plan.c), Cn, Ce, W2), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_edge_hidden_forward(C.byref(plan.c), Cn, Ce, W2, _ptr(w_in), _ptr(b_in), _ptr(x),
                                               _ptr(edge_attr), _ptr(hidden), _ptr(hmax), _ptr(ws), ws.numel(), _stream()),
                  "cgat_edge_hidden_forward")
        ctx.plan = plan
        ctx.save_for_backward(x, edge_attr, w_in, hidden)
        ctx.mark_non_differentiable(hmax)
        return hidden, hmax

    @staticmethod
    def backward(ctx, g_hidden, _g_hmax=None):
        x, edge_attr, w_in, hidden = ctx.saved_tensors
        plan = ctx.plan
        g_hidden = _f32c(g_hidden)
        Cn, Ce, W2 = x.shape[1], edge_attr.shape[1], w_in.shape[0]
        dev = x.device
        g_x, g_e = torch.empty_like(x), torch.empty_like(edge_attr)
        g_w, g_b = torch.empty_like(w_in), torch.empty(W2, dtype=torch.float32, device=dev)
        ws = workspace(lib.cgat_edge_hidden_backward_workspace_bytes(C.byref(plan.c), Cn, Ce, W2), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_edge_hidden_backward(C.byref(plan.c), Cn, Ce, W2, _ptr(w_in), _ptr(x), _ptr(edge_attr),
                                                _ptr(hidden), _ptr(g_hidden), 0, None, _ptr(g_x), _ptr(g_e), _ptr(g_w),
                                                _ptr(g_b), _ptr(ws), ws.numel(), _stream()), "cgat_edge_hidden_backward")
        return g_x, g_e, None, g_w, g_b


# ----------------------------------------------------------------------------------------
# hypernetwork Pooling_NN
# ----------------------------------------------------------------------------------------
def _hnet_struct(cls, W, n_fc, n_hyper, flat, damping):
    s = cls()
    if cls is _lib.HnetParams:
        s.W, s.n_fc, s.n_hyper = W, n_fc, n_hyper
    per = 2 * n_fc + 2
    for l in range(n_hyper):
        t = flat[l * per:(l + 1) * per]
        for k in range(n_fc):
            s.layer[l].fc_w[k] = t[k].data_ptr()
            s.layer[l].fc_b[k] = t[n_fc + k].data_ptr()
        s.layer[l].head_w = t[2 * n_fc].data_ptr()
        s.layer[l].head_b = t[2 * n_fc + 1].data_ptr()
    s.damping = None if damping is None else damping.data_ptr()
    return s


# Side stream for the hypernetwork's weight-gradient contractions (cgat_hnet_backward_overlapped, used by NodeLayerFn):
# they are off the critical path of the backward pass and matrix-core bound, the attention backward is HBM bound.
# Decided by same-box A/Bs per arithmetic mode: in the 22-bit "f16x3" mode the overlap is worth 0.5 ms of a 23.9 ms step
# (the batched dT launch on half of the chip beside the HBM-bound attention backward) and is the default.  In the default
# 24-bit "f16x3c" mode, since its dT is ONE batched launch (round 5), it is worth 0.5-0.9 ms of 30.0 (29.0-29.5 vs
# 29.9-30.0 ms: the HBM-bound segment backward, source-side sums and dense weight gradients lose little on the other half
# of the chip; profiles/r05_side_stream_sweep.txt) -- opt-in there (CGAT_OVERLAP_WGRAD=1 / set_overlap_wgrad(True)): on
# half of the chip the dominant kernel's launch takes 8.2-8.8 ms instead of 4.6, and the step's per-kernel durations --
# the bench line's roofline, the rocprof statistics -- stop being statements about the kernels; the default keeps every
# kernel alone on the chip.  "bf16x6" (four six-pass dT launches, two to four times as long as the kernels they would
# run beside) gains nothing: 33.13 / 33.12 ms with, 33.04 / 33.08 without.
_side_streams = {}
_overlap_wgrad = {"0": False, "1": True}.get(os.environ.get("CGAT_OVERLAP_WGRAD", ""), None)


def set_overlap_wgrad(flag):
    global _overlap_wgrad
    _overlap_wgrad = None if flag is None else bool(flag)


def get_overlap_wgrad():
    """The raw setting (None = decided per arithmetic mode): what a caller that changes it temporarily must restore."""
    return _overlap_wgrad


def overlap_enabled():
    if _overlap_wgrad is None:
        return get_bilinear_mode() == "f16x3"
    return _overlap_wgrad


def side_stream(device, create=True):
    """The per-device side stream (None if it has never been used and create is False)."""
    dev = torch.device(device)
    key = dev.index if dev.index is not None else torch.cuda.current_device()
    if key not in _side_streams and create:
        # CGAT_SIDE_PRIORITY=-1: a high-priority side stream (its workgroups are placed ahead of the main stream's)
        _side_streams[key] = torch.cuda.Stream(device=dev, priority=int(os.environ.get("CGAT_SIDE_PRIORITY", "0")))
    return _side_streams.get(key)


_branch_streams = {}
_branch_max_edges = int(os.environ.get("CGAT_BRANCH_STREAM_MAX_EDGES", "262144"))   # 0: never


def branch_stream(device, n_edges, which=0):
    """The stream independent sub-networks run on beside the main stream at SMALL batches (CGAtNet: the composition
    branch beside the graph layers) -- or None when the batch is large enough for every kernel to fill the chip by
    itself.  At the harness' shipped batch (64 crystals) a kernel occupies 5-40 of the 256 CUs for as long as one
    workgroup needs for its serial chain, so two independent chains side by side cost the longer one, not the sum
    (SURVEY 8 f3).  Works eagerly and under hipGraph capture (fork / join by events); autograd runs each node's backward
    on its forward's stream, so the backward overlaps the same way."""
    if _branch_max_edges <= 0 or n_edges > _branch_max_edges:
        return None
    dev = torch.device(device)
    key = (dev.index if dev.index is not None else torch.cuda.current_device(), which)
    if key not in _branch_streams:
        _branch_streams[key] = torch.cuda.Stream(device=dev)
    return _branch_streams[key]


def aux_streams():
    """Every stream this package may have issued gradient-producing work on besides the caller's (dist.py makes a
    collective wait for them)."""
    return list(_branch_streams.values()) + list(_side_streams.values())


class HNetFn(torch.autograd.Function):
    """y = HyperFC(hyper_input)(v)  with hyper_input = h0 (H_Net_0) or d*h0 + (1-d)*v (H_Net);
    reference Hypernetworksmp.py:257-313.  `flat` = per predicted layer: n_fc trunk weights,
    n_fc trunk biases, head weight [W*W+W, W], head bias [W*W+W]."""

    @staticmethod
    def forward(ctx, h0, v, damping, n_fc, n_hyper, *flat):
        _require_gpu(h0, v, *flat)
        h0, v = _f32c(h0), _f32c(v)
        flat = [_f32c(t.detach()) for t in flat]
        d = None if damping is None else _f32c(damping.detach())
        rows, W = v.shape
        if h0.shape != v.shape:
            raise ValueError("hypernetwork: h0 and v must both be [rows, W]")
        for l in range(n_hyper):
            hw = flat[l * (2 * n_fc + 2) + 2 * n_fc]
            if tuple(hw.shape) != (W * W + W, W):
                raise ValueError(f"hypernetwork head weight must be [{W * W + W}, {W}] (all widths equal), got {tuple(hw.shape)}")
        p = _hnet_struct(_lib.HnetParams, W, n_fc, n_hyper, flat, d)
        dev = v.device
        saved = _scratch(lib.cgat_hnet_saved_floats(rows, C.byref(p)), torch.float32, dev)
        y = torch.empty_like(v)
        ws = workspace(lib.cgat_hnet_forward_workspace_bytes(rows, C.byref(p)), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_hnet_forward(rows, C.byref(p), _ptr(h0), _ptr(v), _ptr(y), _ptr(saved), _ptr(ws), ws.numel(),
                                        _stream()), "cgat_hnet_forward")
        ctx.n_fc, ctx.n_hyper, ctx.has_d = n_fc, n_hyper, d is not None
        ctx.save_for_backward(h0, v, saved, *( [d] if d is not None else []), *flat)
        return y

    @staticmethod
    def backward(ctx, g_y):
        h0, v, saved, *rest = ctx.saved_tensors
        d = rest.pop(0) if ctx.has_d else None
        flat = rest
        n_fc, n_hyper = ctx.n_fc, ctx.n_hyper
        rows, W = v.shape
        g_y = _f32c(g_y)
        p = _hnet_struct(_lib.HnetParams, W, n_fc, n_hyper, flat, d)
        grads = [_param_grad(t) for t in flat]
        g_d = torch.empty_like(d) if d is not None else None
        g = _hnet_struct(_lib.HnetGrads, W, n_fc, n_hyper, grads, g_d)
        g_h0, g_v = torch.empty_like(h0), torch.empty_like(v)
        dev = v.device
        ws = workspace(lib.cgat_hnet_backward_workspace_bytes(rows, C.byref(p)), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_hnet_backward(rows, C.byref(p), _ptr(h0), _ptr(v), _ptr(saved), _ptr(g_y), _ptr(g_h0),
                                         _ptr(g_v), C.byref(g), _ptr(ws), ws.numel(), _stream()), "cgat_hnet_backward")
        return (g_h0, g_v, g_d, None, None, *grads)


class NodeLayerFn(torch.autograd.Function):
    """One whole GATConvNodes layer: aggr = attention(x, edge_attr) (NodesAttentionFn), y = HyperFC(h0)(aggr) (HNetFn) --
    reference CGAT.py:307-340 -- as ONE autograd node, so that its backward can run the hypernetwork's four
    weight-gradient contractions (matrix-core bound, feeding nothing else) on a side stream BESIDE the attention
    backward (HBM bound) and join the two streams before returning.  Same kernels, same results as the two separate
    nodes; only the schedule differs."""

    @staticmethod
    def forward(ctx, x, edge_attr, h0, plan, H, damping, n_fc, n_hyper, *params):
        attn_w, flat = list(params[:8]), list(params[8:])
        _require_gpu(x, edge_attr, h0, *attn_w, *flat)
        x, edge_attr, h0 = _f32c(x), _f32c(edge_attr), _f32c(h0)
        attn_w = [_f32c(w.detach()) for w in attn_w]
        flat = [_f32c(t.detach()) for t in flat]
        d = None if damping is None else _f32c(damping.detach())
        N, E = plan.N, plan.E
        if x.shape[0] != N or edge_attr.shape[0] != E or h0.shape != x.shape:
            raise ValueError("NodeLayerFn: shapes do not match the plan")
        dev = x.device
        pa, Hd = _attn_params(x, edge_attr, H, attn_w)
        W = x.shape[1]
        ph = _hnet_struct(_lib.HnetParams, W, n_fc, n_hyper, flat, d)
        saved_a = _scratch(lib.cgat_nodes_attention_saved_floats(N, E, H, Hd), torch.float32, dev)
        saved_h = _scratch(lib.cgat_hnet_saved_floats(N, C.byref(ph)), torch.float32, dev)
        aggr, y = torch.empty(N, W, dtype=torch.float32, device=dev), torch.empty(N, W, dtype=torch.float32, device=dev)
        with torch.cuda.device(dev):
            ws = workspace(lib.cgat_nodes_attention_forward_workspace_bytes(C.byref(plan.c), C.byref(pa)), dev)
            check(lib.cgat_nodes_attention_forward(C.byref(plan.c), C.byref(pa), _ptr(x), _ptr(edge_attr), _ptr(aggr),
                                                   _ptr(saved_a), _ptr(ws), ws.numel(), _stream()),
                  "cgat_nodes_attention_forward")
            ws = workspace(lib.cgat_hnet_forward_workspace_bytes(N, C.byref(ph)), dev)
            check(lib.cgat_hnet_forward(N, C.byref(ph), _ptr(h0), _ptr(aggr), _ptr(y), _ptr(saved_h), _ptr(ws), ws.numel(),
                                        _stream()), "cgat_hnet_forward")
        ctx.plan, ctx.H, ctx.n_fc, ctx.n_hyper, ctx.has_d = plan, H, n_fc, n_hyper, d is not None
        ctx.storage = lib.cgat_get_edge_storage()
        ctx.save_for_backward(x, edge_attr, h0, aggr, saved_a, saved_h, *([d] if d is not None else []), *attn_w, *flat)
        if debug.recording():
            debug.note_attention(params[0], params[4], plan, pa, saved_a, 2 * H * Hd)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, edge_attr, h0, aggr, saved_a, saved_h, *rest = ctx.saved_tensors
        d = rest.pop(0) if ctx.has_d else None
        attn_w, flat = rest[:8], rest[8:]
        plan, H, n_fc, n_hyper = ctx.plan, ctx.H, ctx.n_fc, ctx.n_hyper
        N, W = x.shape
        dev = x.device
        g_y = _f32c(g_y)
        ph = _hnet_struct(_lib.HnetParams, W, n_fc, n_hyper, flat, d)
        g_flat = [_param_grad(t) for t in flat]
        g_d = torch.empty_like(d) if d is not None else None
        gh = _hnet_struct(_lib.HnetGrads, W, n_fc, n_hyper, g_flat, g_d)
        g_h0, g_aggr = torch.empty_like(h0), torch.empty_like(aggr)
        pa, Hd = _attn_params(x, edge_attr, H, attn_w)
        g_x, g_e = torch.empty_like(x), torch.empty_like(edge_attr)
        g_attn = [_param_grad(w) for w in attn_w]
        ga = _lib.AttnGrads(*[t.data_ptr() for t in g_attn])
        main = torch.cuda.current_stream(dev)
        s2 = side_stream(dev)
        with torch.cuda.device(dev):
            # side_ws (the side stream's inputs and scratch) is its own allocation: the attention backward reuses the
            # cached main workspace while the side stream is still running
            ws_h = workspace(lib.cgat_hnet_backward_workspace_bytes(N, C.byref(ph)), dev)
            side_ws = _scratch(lib.cgat_hnet_backward_side_workspace_bytes(N, C.byref(ph)), torch.uint8, dev)
            check(lib.cgat_hnet_backward_overlapped(N, C.byref(ph), _ptr(h0), _ptr(aggr), _ptr(saved_h), _ptr(g_y),
                                                    _ptr(g_h0), _ptr(g_aggr), C.byref(gh), _ptr(ws_h), ws_h.numel(),
                                                    main.cuda_stream, _ptr(side_ws), side_ws.numel(), s2.cuda_stream),
                  "cgat_hnet_backward_overlapped")
            ws_a = workspace(lib.cgat_nodes_attention_backward_workspace_bytes(C.byref(plan.c), C.byref(pa)), dev)
            with _storage_of(ctx.storage):
                check(lib.cgat_nodes_attention_backward(C.byref(plan.c), C.byref(pa), _ptr(x), _ptr(edge_attr), _ptr(saved_a),
                                                        _ptr(g_aggr), _ptr(g_x), _ptr(g_e), C.byref(ga), _ptr(ws_a),
                                                        ws_a.numel(), main.cuda_stream), "cgat_nodes_attention_backward")
            main.wait_stream(s2)     # every gradient is complete, in stream order, when this node returns
        return (g_x, g_e, g_h0, None, None, g_d, None, None, *g_attn, *g_flat)


# ----------------------------------------------------------------------------------------
# dense layer, segment ops
# ----------------------------------------------------------------------------------------
class LinearFn(torch.autograd.Function):
    """y = act(x W^T + b); x [M,K], W [N,K] (any leading stride), act in {none,tanh,leaky,relu}."""

    @staticmethod
    def forward(ctx, x, w, b, act):
        _require_gpu(x, w, b)
        # a column slice of a wider matrix (one head's hidden block) is passed with its row stride, not copied
        if not (x.dtype == torch.float32 and x.dim() == 2 and x.stride(1) == 1 and x.stride(0) % 4 == 0 and
                x.data_ptr() % 16 == 0):
            x = _f32c(x)
        w2 = w.detach().reshape(w.shape[0], -1)
        if w2.stride(1) != 1:
            w2 = w2.contiguous()
        if w2.dtype != torch.float32:
            raise TypeError("weights must be float32")
        bb = None if b is None else _f32c(b.detach())
        M, K = x.shape
        N = w2.shape[0]
        y = torch.empty(M, N, dtype=torch.float32, device=x.device)
        ws = workspace(lib.cgat_linear_forward_workspace_bytes(M, K, N), x.device)
        with torch.cuda.device(x.device):
            check(lib.cgat_linear_forward(_ptr(x), x.stride(0), _ptr(w2), w2.stride(0), _ptr(bb), _ptr(y), N, M, K, N, act,
                                          None, _ptr(ws), ws.numel(), _stream()), "cgat_linear_forward")
        ctx.act, ctx.has_b, ctx.wshape = act, b is not None, w.shape
        ctx.save_for_backward(x, w2, y)
        if act in (_lib.ACT_LEAKY, _lib.ACT_RELU) and debug.recording():
            debug.note(w, y > 0)
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, w2, y = ctx.saved_tensors
        g_y = _f32c(g_y)
        M, K = x.shape
        N = w2.shape[0]
        dev = x.device
        gpre = torch.empty_like(y) if ctx.act != _lib.ACT_NONE else None
        g_x = torch.empty(M, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[0] else None
        g_w = torch.empty(N, K, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        g_b = torch.empty(N, dtype=torch.float32, device=dev) if (ctx.has_b and ctx.needs_input_grad[2]) else None
        ws = workspace(lib.cgat_linear_backward_workspace_bytes(M, K, N), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_linear_backward(_ptr(x), x.stride(0), _ptr(w2), w2.stride(0), _ptr(y), N, _ptr(g_y), N, _ptr(gpre),
                                           _ptr(g_x), K, 0, _ptr(g_w), K, _ptr(g_b), M, K, N, ctx.act, _ptr(ws),
                                           ws.numel(), _stream()), "cgat_linear_backward")
        return g_x, (None if g_w is None else g_w.reshape(ctx.wshape)), g_b, None


def _cos(Co, n):
    """Output width per network: one int for all, or one per network (scalar attention: 1 for MH_A, C for MH_M)."""
    return tuple(Co) if isinstance(Co, (tuple, list)) else (Co,) * n


def _heads_forward(hid, nets, H, Hd, Co, hmax=None):
    """nets: list of (weight [H*Co, Hd], bias [H*Co] or None); network i reads columns [i*H*Hd, (i+1)*H*Hd) of hid.
    hmax: device scalar max |hid| (EdgeHiddenFn's second output), which lets the Hd > 128 products run in the f16x3 form."""
    E, W2 = hid.shape
    cos = _cos(Co, len(nets))
    outs = [torch.empty(E, H * co, dtype=torch.float32, device=hid.device) for co in cos]
    ws = workspace(max(lib.cgat_heads_linear_forward_workspace_bytes(E, Hd, co, H) for co in cos), hid.device)
    with torch.cuda.device(hid.device):
        for net, (w, b) in enumerate(nets):
            Co = cos[net]
            # the H heads of a network as ONE call: head h reads the column block (net * H + h) * Hd of hid, rows
            # [h * Co, (h + 1) * Co) of the weight / bias, and writes columns [h * Co, (h + 1) * Co) of the output
            check(lib.cgat_heads_linear_forward(_ptr(hid[:, net * H * Hd:]), W2, Hd, _ptr(w), Hd, Co * Hd, _ptr(b), Co,
                                                _ptr(outs[net]), H * Co, Co, E, Hd, Co, H, _ptr(hmax), _ptr(ws), ws.numel(),
                                                _stream()), "cgat_heads_linear_forward")
    return outs


def _heads_backward(hid, weights, has_b, grads, H, Hd, Co, need_hid):
    E, W2 = hid.shape
    dev = hid.device
    cos = _cos(Co, len(weights))
    gs = [_f32c(g.reshape(E, H * co)) for g, co in zip(grads, cos)]
    g_hid = torch.empty_like(hid) if need_hid else None
    g_w = [torch.empty(H * co, Hd, dtype=torch.float32, device=dev) for co in cos]
    g_b = [torch.empty(H * co, dtype=torch.float32, device=dev) if hb else None for hb, co in zip(has_b, cos)]
    ws = workspace(max(lib.cgat_linear_backward_workspace_bytes(E, Hd, co) for co in cos), dev)
    with torch.cuda.device(dev):
        for net, w in enumerate(weights):
            Co = cos[net]
            for h in range(H):
                col = (net * H + h) * Hd
                check(lib.cgat_linear_backward(_ptr(hid[:, col:col + Hd]), W2, _ptr(w[h * Co:(h + 1) * Co]), Hd, None, Co,
                                               _ptr(gs[net][:, h * Co:(h + 1) * Co]), H * Co, None,
                                               None if g_hid is None else _ptr(g_hid[:, col:col + Hd]), W2, 0,
                                               _ptr(g_w[net][h * Co:(h + 1) * Co]), Hd,
                                               None if g_b[net] is None else _ptr(g_b[net][h * Co:(h + 1) * Co]),
                                               E, Hd, Co, _lib.ACT_NONE, _ptr(ws), ws.numel(), _stream()),
                      "cgat_linear_backward")
    return g_hid, g_w, g_b


class HeadsLinearFn(torch.autograd.Function):
    """Second layers of two multi-head networks on one hidden matrix (vector attention, reference CGAT.py:97-98,
    103-109 applied per head): out_A[:, h, :] = hid[:, h*Hd:(h+1)*Hd] W_A[h]^T + b_A[h], out_M the same on the second
    half of the columns.  One autograd node for all 2H slices: the backward writes every head's input gradient
    straight into its column block of ONE [E, 2*H*Hd] buffer -- per-slice autograd nodes made torch zero-fill and add
    a full-size 6 GB gradient per head (21 ms of a 75 ms step at E = 1M)."""

    @staticmethod
    def forward(ctx, hid, wa, ba, wm, bm, H, Hd, Co, hmax=None):
        _require_gpu(hid, wa, wm)
        hid = _f32c(hid)
        E = hid.shape[0]
        cos = _cos(Co, 2)                   # (Co may be a pair: scalar attention's MH_A emits one logit per head)
        ws_ = [_f32c(w.detach().reshape(H * co, Hd)) for w, co in zip((wa, wm), cos)]
        bs_ = [None if b is None else _f32c(b.detach()) for b in (ba, bm)]
        outs = _heads_forward(hid, list(zip(ws_, bs_)), H, Hd, cos, hmax)
        ctx.dims = (H, Hd, cos)
        ctx.shapes = (wa.shape, wm.shape)
        ctx.has_b = (ba is not None, bm is not None)
        ctx.save_for_backward(hid, ws_[0], ws_[1])
        return outs[0].reshape(E, H, cos[0]), outs[1].reshape(E, H, cos[1])

    @staticmethod
    def backward(ctx, g_a, g_m):
        hid, w0, w1 = ctx.saved_tensors
        H, Hd, Co = ctx.dims
        g_hid, g_w, g_b = _heads_backward(hid, (w0, w1), ctx.has_b, (g_a, g_m), H, Hd, Co, ctx.needs_input_grad[0])
        return (g_hid, g_w[0].reshape(ctx.shapes[0]), g_b[0], g_w[1].reshape(ctx.shapes[1]), g_b[1], None, None, None, None)


class EdgeHiddenHeadsFn(torch.autograd.Function):
    """EdgeHiddenFn followed by HeadsLinearFn as ONE autograd node (vector attention: both networks' first layers on
    [x_i; edge_attr; x_j], then the 2H per-head second layers; reference CGAT.py:96-109 on the message of 316-318).
    Same forward kernels; the point is the backward: the second layers' input gradient g_hid = g W is the gradient of a
    LeakyReLU OUTPUT, and autograd turns it into the pre-activation gradient with an elementwise pass over [E, 2 H Hd]
    (30 GB per layer at the harness-default shape: 12 % of its step).  Here the product's epilogue multiplies by
    LeakyReLU'(sign of hidden) and takes the tensor maximum on the way (cgat_linear_backward_dact), and the first
    layers' backward consumes the result as it is (cgat_edge_hidden_backward, g_is_pre = 1)."""

    @staticmethod
    def eligible(hid_width, H, Hd, cos):
        return (get_bilinear_mode() != "f32" and Hd % 128 == 0 and all(co == 128 for co in cos) and
                hid_width == 2 * H * Hd)

    @staticmethod
    def forward(ctx, x, edge_attr, plan, w_in, b_in, wa, ba, wm, bm, H, Hd, Co):
        _require_gpu(x, edge_attr, w_in, b_in, wa, wm)
        x, edge_attr, w_in, b_in = _f32c(x), _f32c(edge_attr), _f32c(w_in.detach()), _f32c(b_in.detach())
        N, E = plan.N, plan.E
        Cn, Ce, W2 = x.shape[1], edge_attr.shape[1], w_in.shape[0]
        if x.shape[0] != N or edge_attr.shape[0] != E or w_in.shape[1] != 2 * Cn + Ce or b_in.numel() != W2:
            raise ValueError("EdgeHiddenHeadsFn: shapes do not match the plan / the stacked first-layer weight")
        dev = x.device
        hidden = torch.empty(E, W2, dtype=torch.float32, device=dev)
        hmax = torch.empty(1, dtype=torch.float32, device=dev)
        ws = workspace(lib.cgat_edge_hidden_forward_workspace_bytes(C.byref(plan.c), Cn, Ce, W2), dev)
        with torch.cuda.device(dev):
            check(lib.cgat_edge_hidden_forward(C.byref(plan.c), Cn, Ce, W2, _ptr(w_in), _ptr(b_in), _ptr(x),
                                               _ptr(edge_attr), _ptr(hidden), _ptr(hmax), _ptr(ws), ws.numel(), _stream()),
                  "cgat_edge_hidden_forward")
        cos = _cos(Co, 2)
        ws_ = [_f32c(w.detach().reshape(H * co, Hd)) for w, co in zip((wa, wm), cos)]
        bs_ = [None if b is None else _f32c(b.detach()) for b in (ba, bm)]
        outs = _heads_forward(hidden, list(zip(ws_, bs_)), H, Hd, cos, hmax)
        ctx.plan, ctx.dims = plan, (H, Hd, cos)
        ctx.shapes = (wa.shape, wm.shape)
        ctx.has_b = (ba is not None, bm is not None)
        ctx.save_for_backward(x, edge_attr, w_in, hidden, ws_[0], ws_[1])
        if debug.recording():
            debug.last_hidden = hidden             # the calling module records the derivative pattern under its weights' names
        return outs[0].reshape(E, H, cos[0]), outs[1].reshape(E, H, cos[1])

    @staticmethod
    def backward(ctx, g_a, g_m):
        x, edge_attr, w_in, hidden, w0, w1 = ctx.saved_tensors
        plan = ctx.plan
        H, Hd, cos = ctx.dims
        E, W2 = hidden.shape
        Cn, Ce = x.shape[1], edge_attr.shape[1]
        dev = x.device
        gs = [_f32c(g.reshape(E, H * co)) for g, co in zip((g_a, g_m), cos)]
        gpre = torch.empty_like(hidden)             # gradient of the PRE-activations, written head by head
        gmax = torch.zeros(1, dtype=torch.float32, device=dev)
        g_w = [torch.empty(H * co, Hd, dtype=torch.float32, device=dev) for co in cos]
        g_b = [torch.empty(H * co, dtype=torch.float32, device=dev) if hb else None for hb, co in zip(ctx.has_b, cos)]
        ws = workspace(max(lib.cgat_heads_linear_backward_dact_workspace_bytes(E, Hd, co, H) for co in cos), dev)
        with torch.cuda.device(dev):
            for net, w in enumerate((w0, w1)):
                Co = cos[net]
                col = net * H * Hd
                check(lib.cgat_heads_linear_backward_dact(
                    _ptr(hidden[:, col:]), W2, Hd, _ptr(w), Hd, Co * Hd, _ptr(gs[net]), H * Co, Co,
                    _ptr(gpre[:, col:]), W2, Hd, _ptr(hidden[:, col:]), W2, Hd, _ptr(gmax), _ptr(g_w[net]), Hd, Co * Hd,
                    _ptr(g_b[net]), Co, E, Hd, Co, H, _ptr(ws), ws.numel(), _stream()), "cgat_heads_linear_backward_dact")
            g_x, g_e = torch.empty_like(x), torch.empty_like(edge_attr)
            g_win, g_bin = torch.empty_like(w_in), torch.empty(W2, dtype=torch.float32, device=dev)
            ws2 = workspace(lib.cgat_edge_hidden_backward_workspace_bytes(C.byref(plan.c), Cn, Ce, W2), dev)
            check(lib.cgat_edge_hidden_backward(C.byref(plan.c), Cn, Ce, W2, _ptr(w_in), _ptr(x), _ptr(edge_attr),
                                                _ptr(hidden), _ptr(gpre), 1, _ptr(gmax), _ptr(g_x), _ptr(g_e), _ptr(g_win),
                                                _ptr(g_bin), _ptr(ws2), ws2.numel(), _stream()), "cgat_edge_hidden_backward")
        return (g_x, g_e, None, g_win, g_bin, g_w[0].reshape(ctx.shapes[0]), g_b[0], g_w[1].reshape(ctx.shapes[1]), g_b[1],
                None, None, None)


class HeadsLinear1Fn(torch.autograd.Function):
    """The same for one network: out[:, h, :] = hid[:, h*Hd:(h+1)*Hd] W[h]^T + b[h] (MultiHeadNetwork.fc_out)."""

    @staticmethod
    def forward(ctx, hid, w, b, H, Hd, Co):
        _require_gpu(hid, w)
        hid = _f32c(hid)
        E = hid.shape[0]
        w2 = _f32c(w.detach().reshape(H * Co, Hd))
        bb = None if b is None else _f32c(b.detach())
        out = _heads_forward(hid, [(w2, bb)], H, Hd, Co)[0]
        ctx.dims = (H, Hd, Co)
        ctx.wshape = w.shape
        ctx.has_b = b is not None
        ctx.save_for_backward(hid, w2)
        return out.reshape(E, H, Co)

    @staticmethod
    def backward(ctx, g):
        hid, w2 = ctx.saved_tensors
        H, Hd, Co = ctx.dims
        g_hid, g_w, g_b = _heads_backward(hid, (w2,), (ctx.has_b,), (g,), H, Hd, Co, ctx.needs_input_grad[0])
        return g_hid, g_w[0].reshape(ctx.wshape), g_b[0], None, None, None


class SmallEmbeddingFn(torch.autograd.Function):
    """table[idx] for a small table looked up once per edge (nn.Embedding(neighbor_number + 1, Ce) in CGAtNet): the
    forward is a row gather, the backward a deterministic per-class row sum (cgat_embedding_backward) instead of
    torch's sort + atomic scatter."""

    @staticmethod
    def forward(ctx, idx, table):
        _require_gpu(idx, table)
        ctx.save_for_backward(idx)
        ctx.tshape = table.shape
        return table.detach().index_select(0, idx.reshape(-1)).reshape(*idx.shape, table.shape[1])

    @staticmethod
    def backward(ctx, g):
        (idx,) = ctx.saved_tensors
        K, Cc = ctx.tshape
        g2 = _f32c(g.reshape(-1, Cc))
        ii = idx.reshape(-1).contiguous()
        out = torch.empty(K, Cc, dtype=torch.float32, device=g2.device)
        ws = workspace(lib.cgat_embedding_backward_workspace_bytes(K, Cc), g2.device)
        with torch.cuda.device(g2.device):
            check(lib.cgat_embedding_backward(_ptr(g2), Cc, _ptr(ii), g2.shape[0], K, Cc, _ptr(out), _ptr(ws), ws.numel(),
                                              _stream()), "cgat_embedding_backward")
        return None, out


def small_embedding(idx, table):
    """nn.Embedding lookup with the deterministic backward when the table is small enough, F.embedding otherwise."""
    if (table.is_cuda and idx.dtype == torch.int64 and table.dtype == torch.float32 and table.shape[1] <= 128 and
            table.shape[0] * table.shape[1] <= 8192):
        return SmallEmbeddingFn.apply(idx, table)
    return torch.nn.functional.embedding(idx, table)


def linear(x, w, b=None, act=_lib.ACT_NONE):
    lead = x.shape[:-1]
    x2 = x.reshape(-1, x.shape[-1])
    if rowprog.eligible(x2):
        # a few hundred rows: one wave per 16 x 16 output tile (csrc/rowprog.hip), one launch per direction
        (y,) = rowprog.RowNetsFn.apply(x2, None, (((act, 0, b is not None),),), w, b, None)
        return y.reshape(*lead, y.shape[-1])
    y = LinearFn.apply(x.reshape(-1, x.shape[-1]), w, b, act)
    return y.reshape(*lead, y.shape[-1])


class SegmentSoftmaxFn(torch.autograd.Function):
    """alpha = mult * exp(a - segmax) / (segsum + eps) over CSR-ordered rows [R, F]."""

    @staticmethod
    def forward(ctx, a, mult, rowptr, eps):
        _require_gpu(a, mult, rowptr)
        a = _f32c(a)
        m = None if mult is None else _f32c(mult)
        R, F = a.shape
        S = rowptr.numel() - 1
        alpha = torch.empty_like(a)
        with torch.cuda.device(a.device):
            check(lib.cgat_segment_softmax_forward(_ptr(a), _ptr(m), _ptr(rowptr), S, F, eps, _ptr(alpha), _stream()),
                  "cgat_segment_softmax_forward")
        ctx.has_m = m is not None
        ctx.save_for_backward(alpha, rowptr, *([m] if m is not None else []))
        return alpha

    @staticmethod
    def backward(ctx, g_alpha):
        alpha, rowptr, *rest = ctx.saved_tensors
        m = rest[0] if ctx.has_m else None
        g_alpha = _f32c(g_alpha)
        R, F = alpha.shape
        S = rowptr.numel() - 1
        g_a = torch.empty_like(alpha)
        g_m = torch.empty_like(m) if (m is not None and ctx.needs_input_grad[1]) else None
        with torch.cuda.device(alpha.device):
            check(lib.cgat_segment_softmax_backward(_ptr(alpha), _ptr(g_alpha), _ptr(m), _ptr(rowptr), S, F, _ptr(g_a),
                                                    _ptr(g_m), _stream()), "cgat_segment_softmax_backward")
        return g_a, g_m, None, None


class SegmentSumFn(torch.autograd.Function):
    """out[s] = sum of the CSR-ordered rows of segment s; backward = broadcast."""

    @staticmethod
    def forward(ctx, x, rowptr, seg_of_row):
        _require_gpu(x, rowptr)
        x = _f32c(x)
        R, F = x.shape
        S = rowptr.numel() - 1
        out = torch.empty(S, F, dtype=torch.float32, device=x.device)
        with torch.cuda.device(x.device):
            check(lib.cgat_segment_sum(_ptr(x), F, None, _ptr(rowptr), S, F, _ptr(out), F, _stream()), "cgat_segment_sum")
        ctx.save_for_backward(seg_of_row)
        return out

    @staticmethod
    def backward(ctx, g):
        (seg_of_row,) = ctx.saved_tensors
        return g.index_select(0, seg_of_row), None, None


def _chain_desc(rows, x, layers, in_dact=None, in_dact_type=0, in_store=None):
    """cgat_chain_desc from python values; `layers` = dicts with W, transposed, bias, act, dact, dact_type, resid, out,
    accumulate (tensors are 2-D row-major [rows,128] / [128,128])."""
    d = _lib.ChainDesc()
    d.n_layers, d.rows = len(layers), rows
    d.x, d.ldx = x.data_ptr(), x.stride(0)
    if in_dact is not None:
        d.in_dact, d.ld_in_dact, d.in_dact_type = in_dact.data_ptr(), in_dact.stride(0), in_dact_type
    if in_store is not None:
        d.in_store, d.ld_in_store = in_store.data_ptr(), in_store.stride(0)
    for i, L in enumerate(layers):
        c = d.layer[i]
        W = L["W"]
        c.W = W.data_ptr()
        c.w_so, c.w_sk = (1, W.stride(0)) if L.get("transposed") else (W.stride(0), 1)
        for k in ("dact", "resid", "out"):
            if L.get(k) is not None:
                setattr(c, k, L[k].data_ptr())
                setattr(c, "ld_" + k, L[k].stride(0))
        if L.get("bias") is not None:
            c.bias = L["bias"].data_ptr()
        c.act, c.dact_type, c.accumulate = L.get("act", 0), L.get("dact_type", 0), int(L.get("accumulate", False))
    return d


def _run_chain(d, device):
    ws = workspace(lib.cgat_mlp_chain_workspace_bytes(d.n_layers), device)
    with torch.cuda.device(device):
        check(lib.cgat_mlp_chain(C.byref(d), _ptr(ws), ws.numel(), _stream()), "cgat_mlp_chain")


class ChainMLPFn(torch.autograd.Function):
    """out = resid + W_n act(... act(W_1 x + b_1) ...) + b_n for width-128 layers in ONE launch per direction
    (csrc/chain.hip): SimpleNetwork (message_changed.py:36-63) on [rows,128] inputs, with the residual of CGAtNet's
    `edge_attr + Edge(...)` (CGAT.py:580-585) folded in.  Hidden activations are kept for backward; the backward chain
    runs on the transposed weights with the activation derivatives folded in and leaves every pre-activation gradient
    for the one-pass weight + bias gradient kernel."""

    @staticmethod
    def eligible(x, weights, resid=None):
        if not (x.is_cuda and x.dtype == torch.float32 and x.dim() == 2 and x.shape[1] == 128 and x.shape[0] > 0):
            return False
        if get_bilinear_mode() not in ("f16x3", "f16x3c", "bf16x6") or not (1 <= len(weights) <= 5):
            return False
        if resid is not None and (resid.shape != x.shape or resid.dtype != torch.float32):
            return False
        return all(tuple(w.shape) == (128, 128) and w.dtype == torch.float32 for w in weights)

    @staticmethod
    def forward(ctx, x, resid, act, *wb):
        n = len(wb) // 2
        ws_, bs_ = [_f32c(w.detach()) for w in wb[:n]], [_f32c(b.detach()) for b in wb[n:]]
        # x + f(x): the backward chain then adds g_out in its last layer.  Callers pass two separate reshape() results
        # of one tensor (SimpleNetwork.forward), so identity is decided on storage, shape and strides
        ctx.same = resid is not None and (resid is x or (resid.data_ptr() == x.data_ptr() and resid.shape == x.shape and
                                                         resid.stride() == x.stride() and resid.dtype == x.dtype))
        x = _f32c(x)
        r = None if resid is None else (x if ctx.same else _f32c(resid))
        rows, dev = x.shape[0], x.device
        hid = [torch.empty(rows, 128, dtype=torch.float32, device=dev) for _ in range(n - 1)]
        out = torch.empty(rows, 128, dtype=torch.float32, device=dev)
        layers = [dict(W=ws_[i], bias=bs_[i], act=act, out=hid[i]) for i in range(n - 1)]
        layers.append(dict(W=ws_[-1], bias=bs_[-1], act=_lib.ACT_NONE, resid=r, out=out))
        _run_chain(_chain_desc(rows, x, layers), dev)
        if act in (_lib.ACT_LEAKY, _lib.ACT_RELU) and debug.recording():
            for i in range(n - 1):
                debug.note(wb[i], hid[i] > 0)
        ctx.act, ctx.n, ctx.has_r = act, n, resid is not None
        ctx.save_for_backward(x, *hid, *ws_)
        return out

    @staticmethod
    def backward(ctx, g_out):
        n, act = ctx.n, ctx.act
        x, *rest = ctx.saved_tensors
        hid, ws_ = rest[:n - 1], rest[n - 1:]
        g_out = _f32c(g_out)
        rows, dev = x.shape[0], x.device
        gpre = [torch.empty(rows, 128, dtype=torch.float32, device=dev) for _ in range(n - 1)] + [g_out]
        g_x = torch.empty(rows, 128, dtype=torch.float32, device=dev)
        layers = []
        for i in range(n):                               # chain layer i multiplies by W_(n-1-i)
            sl = n - 1 - i
            L = dict(W=ws_[sl], transposed=True)
            if sl > 0:
                L.update(dact=hid[sl - 1], dact_type=act, out=gpre[sl - 1])
            else:
                L.update(out=g_x, resid=g_out if ctx.same else None)
            layers.append(L)
        _run_chain(_chain_desc(rows, g_out, layers), dev)
        g_w, g_b = [], []
        wsz = workspace(lib.cgat_linear_backward_workspace_bytes(rows, 128, 128), dev)
        with torch.cuda.device(dev):
            for sl in range(n):                          # dW_sl = gpre_sl^T input_sl, db_sl = column sums of gpre_sl
                xin = x if sl == 0 else hid[sl - 1]
                gw = torch.empty(128, 128, dtype=torch.float32, device=dev)
                gb = torch.empty(128, dtype=torch.float32, device=dev)
                check(lib.cgat_linear_backward(_ptr(xin), 128, _ptr(ws_[sl]), 128, None, 128, _ptr(gpre[sl]), 128, None,
                                               None, 128, 0, _ptr(gw), 128, _ptr(gb), rows, 128, 128, _lib.ACT_NONE,
                                               _ptr(wsz), wsz.numel(), _stream()), "cgat_linear_backward")
                g_w.append(gw); g_b.append(gb)
        return (g_x, g_out if (ctx.has_r and not ctx.same) else None, None, *g_w, *g_b)


class AttentionPoolFn(torch.autograd.Function):
    """out[s, f] = sum_{r in seg s} alpha[r, f // fw] * m[r, f] with alpha = mult * exp(a - segmax) / (segsum + eps) -- the
    reference's softmax -> multiply -> scatter_add (CGAT.py:323-329, 59-61; roost_message.py:305-317) as one kernel per
    direction.  `a` [R, aF], `m` [R, F] (fw = F // aF), `mult` [R] or None; `perm` = rows of each segment in CSR order
    (None: the rows already are), so nothing is gathered or permuted."""

    @staticmethod
    def supported(a, m):
        aF, F = a.shape[1], m.shape[1]
        if F % 4 or F % aF or not (a.is_cuda and a.dtype == torch.float32 and m.dtype == torch.float32):
            return False
        fw = F // aF
        return (fw == 1 and aF % 4 == 0) or (fw % 4 == 0 and fw // 4 <= 64 and (fw // 4) & (fw // 4 - 1) == 0)

    @staticmethod
    def forward(ctx, a, mult, m, rowptr, perm, eps):
        _require_gpu(a, m, rowptr)
        a, m = _f32c(a), _f32c(m)
        mu = None if mult is None else _f32c(mult.reshape(-1))
        R, aF = a.shape
        F = m.shape[1]
        S = rowptr.numel() - 1
        dev = a.device
        out = torch.empty(S, F, dtype=torch.float32, device=dev)
        mx = torch.empty(S, aF, dtype=torch.float32, device=dev)
        inv = torch.empty(S, aF, dtype=torch.float32, device=dev)
        # low part of the fp64 sum: backward centres on out + out_lo (not needed, hence not computed, without a backward)
        need_lo = any(ctx.needs_input_grad[:3])
        out_lo = torch.empty(S, F, dtype=torch.float32, device=dev) if need_lo else None
        with torch.cuda.device(dev):
            check(lib.cgat_segment_attention_pool_forward(_ptr(a), aF, _ptr(mu), _ptr(m), F, _ptr(rowptr), _ptr(perm), S, F,
                                                          eps, _ptr(out), _ptr(mx), _ptr(inv), _ptr(out_lo), _stream()),
                  "cgat_segment_attention_pool_forward")
        ctx.has_mu, ctx.has_perm, ctx.mshape = mu is not None, perm is not None, None if mult is None else mult.shape
        ctx.save_for_backward(a, m, out, mx, inv, out_lo, rowptr, *([mu] if mu is not None else []),
                              *([perm] if perm is not None else []))
        return out

    @staticmethod
    def backward(ctx, g_out):
        a, m, out, mx, inv, out_lo, rowptr, *rest = ctx.saved_tensors
        mu = rest.pop(0) if ctx.has_mu else None
        perm = rest.pop(0) if ctx.has_perm else None
        g_out = _f32c(g_out)
        R, aF = a.shape
        F = m.shape[1]
        S = rowptr.numel() - 1
        g_a = torch.empty_like(a)
        g_m = torch.empty_like(m) if ctx.needs_input_grad[2] else None
        g_mu = torch.empty_like(mu) if (mu is not None and ctx.needs_input_grad[1]) else None
        with torch.cuda.device(a.device):
            check(lib.cgat_segment_attention_pool_backward(_ptr(a), aF, _ptr(mu), _ptr(m), F, _ptr(rowptr), _ptr(perm), S, F,
                                                           _ptr(out), _ptr(mx), _ptr(inv), _ptr(out_lo), _ptr(g_out), _ptr(g_a),
                                                           _ptr(g_m), F, _ptr(g_mu), _stream()),
                  "cgat_segment_attention_pool_backward")
        return g_a, (None if g_mu is None else g_mu.reshape(ctx.mshape)), g_m, None, None, None


def attention_pool(a, m, index_plan, index, mult=None, eps=1e-16):
    """softmax(a over the segments of index) * m, summed per segment: rows in their ORIGINAL order, `index_plan` the
    SegmentPlan of `index`.  One launch per direction when the shape allows, the generic three-step path otherwise."""
    n = a.shape[0]
    a2, m2 = a.reshape(n, -1), m.reshape(n, -1)
    if n > 0 and AttentionPoolFn.supported(a2, m2) and (mult is None or a2.shape[1] == 1):
        return AttentionPoolFn.apply(a2, mult, m2, index_plan.rowptr, index_plan.perm, eps)
    alpha = segment_softmax(a2, index_plan, mult=mult, eps=eps)
    fw = m2.shape[1] // a2.shape[1]
    w = alpha if fw == 1 else alpha.repeat_interleave(fw, dim=1)
    return segment_sum(w * m2, index_plan, index)


class GatherRowsFn(torch.autograd.Function):
    """x[index] whose backward is an atomics-free segment sum over the index's CSR plan
    (deterministic, unlike index_add)."""

    @staticmethod
    def forward(ctx, x, index, plan):
        _require_gpu(x, index)
        ctx.plan, ctx.rows = plan, x.shape[0]
        return x.index_select(0, index)

    @staticmethod
    def backward(ctx, g):
        plan = ctx.plan
        g = _f32c(g)
        F = g.shape[1]
        out = torch.empty(ctx.rows, F, dtype=torch.float32, device=g.device)
        with torch.cuda.device(g.device):
            check(lib.cgat_segment_sum(_ptr(g), F, _ptr(plan.perm), _ptr(plan.rowptr), plan.S, F, _ptr(out), F,
                                       _stream()), "cgat_segment_sum")
        return out, None, None


def gather_rows(x, index, plan):
    return GatherRowsFn.apply(x, index, plan)


def segment_softmax(a, index_plan, mult=None, eps=1e-16):
    """softmax over the segments of `index_plan` for rows in their ORIGINAL order [R, F]."""
    perm = index_plan.perm64
    a_s = a.index_select(0, perm)
    m_s = None if mult is None else mult.reshape(-1).index_select(0, perm)
    al_s = SegmentSoftmaxFn.apply(a_s, m_s, index_plan.rowptr, eps)
    out = torch.empty_like(al_s)
    return out.index_copy(0, perm, al_s)


def segment_sum(x, index_plan, index):
    """scatter_add(x, index, dim=0, dim_size=S) for rows in their original order."""
    perm = index_plan.perm64
    return SegmentSumFn.apply(x.index_select(0, perm), index_plan.rowptr, index.index_select(0, perm))


_MODES = {"f32": 0, "bf16x6": 6, "bf16x3": 3, "f16x3": 2, "f16x3c": 4}
# the mode a fresh process starts in (env CGAT_BILINEAR_MODE overrides the built-in default)
DEFAULT_MODE = os.environ.get("CGAT_BILINEAR_MODE", "f16x3c")
if DEFAULT_MODE not in _MODES:
    DEFAULT_MODE = "f16x3c"


def set_bilinear_mode(mode):
    """Arithmetic of the width-128 matrix-core kernels.
    "f16x3c" (default, round 4): 24-bit operands.  Every fp32 operand is scaled by a power of two and split EXACTLY into
    three pieces x = h + l + t (two fp16 pieces and the 24th bit); a product is hh + hl + lh on three fp16-MFMA passes
    (exact) plus ll + ht + th (weight <= 2^-22) on three 6-bit MFMA passes of K = 128 (csrc/mfma_bf16.h), fp32 accumulate:
    1.25 x the matrix time of "f16x3" where "bf16x6" needs 2 x.  Kernels without that form run their "bf16x6" form.
    "bf16x6": three bf16 pieces (24 bits), six bf16-MFMA passes.  "f16x3": two fp16 pieces = 22-bit operands, three passes
    (the fastest; not the default because its operands are narrower than fp32's).  "f32": f32-input MFMA, exact fp32
    fmaf chains.  "bf16x3": three bf16 passes, ~4e-6 relative; fails the parity tests (diagnostic)."""
    lib.cgat_set_bilinear_mode(_MODES[mode])


def get_bilinear_mode():
    m = lib.cgat_get_bilinear_mode()
    return {v: k for k, v in _MODES.items()}[m]


# launch-timer tags of the contraction kernels that have the 3.75-pass h + l + t form in the f16x3c mode (bench.py prices a
# kernel against 2500 / 3.75 TFLOP/s only if it is listed here, against 2500 / 6 otherwise)
F16C_KERNELS = ("bilinear_rows", "bilinear_dual", "bilinear_wgrad")


def set_edge_storage(mode):
    """Storage of the per-edge intermediates Z / gZ of the fused scalar-attention path: "f32" (default) or "bf16"
    (BASELINE configs[4]'s "bf16 activations": half the HBM bytes of the edge phase, tolerance 1e-2; logits, softmax
    statistics and all products unchanged).  Exists at C = Ce = 128 in every split arithmetic mode (f16x3c, bf16x6, f16x3);
    a scalar-attention layer without a bf16 form (other widths, the f32 mode) raises instead of silently running in fp32
    storage.  "bf16-mma" (round 6, the 24-bit modes): bf16 storage AND bf16 OPERANDS in the two per-edge
    BACKWARD products (grad edge_attr, grad W_e: ONE matrix pass on round-to-nearest bf16 images of the rebuilt gZ rows
    and of the weight / edge rows, fp32 accumulation) -- BASELINE configs[4]'s "bf16 activations with MFMA edge-MLP",
    tolerance 1e-2.  The forward product keeps its six passes: on bf16 operands the pre-activations move by ~3e-3 of
    their scale, 0.3 % of the LeakyReLU derivatives land on the other side, and the gradients of edge_attr and of the
    attention network's first layer came out 1.6 % ... 5.7 % off (tools/bf16mma_probe.py; measured 775 ms instead of 830
    per 64 M-edge step) -- outside the mode's stated tolerance."""
    lib.cgat_set_edge_storage({"f32": 0, "bf16": 1, "f32+gz": 2, "bf16-mma": 3}[mode])


def get_edge_storage():
    return {0: "f32", 1: "bf16", 2: "f32+gz", 3: "bf16-mma"}[lib.cgat_get_edge_storage()]


class _storage_of:
    """Run a backward under the edge-storage mode its forward ran under."""

    def __init__(self, mode):
        self.mode = mode

    def __enter__(self):
        self.prev = lib.cgat_get_edge_storage()
        lib.cgat_set_edge_storage(self.mode)

    def __exit__(self, *a):
        lib.cgat_set_edge_storage(self.prev)


def prof_enable(on=True):
    lib.cgat_prof_enable(1 if on else 0)


def prof_reset():
    lib.cgat_prof_reset()


def prof_launches():
    """Kernel launches the library has issued so far in this process (all streams)."""
    return int(lib.cgat_prof_launches())


def prof_get(tag):
    n, ms = C.c_int(0), C.c_float(0.0)
    lib.cgat_prof_get(tag.encode(), C.byref(n), C.byref(ms))
    return n.value, ms.value
