"""Parity instrumentation (used by tests/, never on the hot path): records, per activation layer, the derivative
pattern the HIP backward will use -- `post-activation > 0` for every LeakyReLU(0.01) / ReLU of the path (reference
CGAT.py:95, message_changed.py:52,101, roost_message.py:340), keyed by the NAME of the layer's weight in the shared
state_dict layout.

Why: LeakyReLU' and ReLU' jump at 0.  Two correct fp32 evaluations of a pre-activation with |z| ~ 1e-7 max|z| can land
on different sides, and every gradient upstream then differs by a finite amount (the reference's own fp32 and fp64 runs
do).  With the oracle's derivative pattern forced to the recorded one (oracle.cgat_oracle.forced_masks) the comparison
needs no allowance for such flips.

    with cgat_amd.debug.record_masks(model) as masks:
        y = model(...)
    # masks: {"graphs.0.Node.MH_A.fc_in.weight": [bool tensor [rows, units] on the CPU, one per call], ...}
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check

last_hidden = None      # ops.EdgeHiddenHeadsFn leaves its hidden tensor here while a recorder is active
_active = None          # the recorder of the innermost `with`, or None (the normal state: every hook is one `is None` test)


class record_masks:
    def __init__(self, model):
        self.names = {p.data_ptr(): n for n, p in model.named_parameters()}
        self.masks = {}
        self.dropout = []      # keep-masks of the attention dropout (scaled by 1 / (1 - p)), original edge order, in call order

    def __enter__(self):
        global _active
        self.prev, _active = _active, self
        return self.masks

    def __exit__(self, *exc):
        global _active, last_hidden
        _active = self.prev
        last_hidden = None      # the [E, 2 H Hd] activation (GBs at 1 M edges) must not outlive the recorder


def recording():
    return _active is not None


def note(weight, mask):
    """`weight`: the layer's weight parameter (or any view sharing its first element); `mask`: bool [rows, units]."""
    if _active is None:
        return
    name = _active.names.get(weight.data_ptr())
    if name is not None:
        _active.masks.setdefault(name, []).append(mask.detach().to("cpu", torch.bool))


def note_dropout(keep):
    if _active is not None:
        _active.dropout.append(keep.detach().to("cpu"))


def note_attention(a_in_w, m_in_w, plan, attn_params, saved, W2):
    """The fused scalar-attention layer: signs of the saved pre-activations of MH_A | MH_M through the C ABI
    (cgat_debug_nodes_attention_signs), original edge order."""
    if _active is None:
        return
    mask = torch.empty(plan.E, W2, dtype=torch.uint8, device=saved.device)
    with torch.cuda.device(saved.device):
        check(lib.cgat_debug_nodes_attention_signs(C.byref(plan.c), C.byref(attn_params), C.c_void_p(saved.data_ptr()),
                                                   C.c_void_p(mask.data_ptr()),
                                                   C.c_void_p(torch.cuda.current_stream().cuda_stream)),
              "cgat_debug_nodes_attention_signs")
    half = W2 // 2
    note(a_in_w, mask[:, :half] != 0)
    note(m_in_w, mask[:, half:] != 0)


def note_sorted_hidden(weights, plan, hidden):
    """EdgeHiddenFn's output (post-activation, destination-sorted slots) -> per-network masks in original edge order.
    `weights`: the first-layer weights of the stacked networks, equal column shares of `hidden`."""
    global last_hidden
    last_hidden = None          # consumed: drop the module-global reference (the caller's `hidden` argument keeps it for this call)
    if _active is None or hidden is None:
        return
    perm = plan.dst_perm.long()
    m = torch.empty(hidden.shape, dtype=torch.bool, device=hidden.device)
    m[perm] = hidden.detach() > 0
    share = hidden.shape[1] // len(weights)
    for k, w in enumerate(weights):
        note(w, m[:, k * share:(k + 1) * share])
