"""Roost composition branch (reference CGAT/roost_message.py:88-321) on the HIP primitives:
dense layers in the MFMA GEMM kernel, the weighted softmax (w**pow * exp(g - segmax) /
(segsum + 1e-13)) and the scatter-adds in the atomics-free segment kernels."""
import torch
import torch.nn as nn

from . import _lib, rowprog
from .mlp import SimpleNetwork
from .ops import SegmentPlan, attention_pool, gather_rows, get_segment_plan


class WeightedAttention(nn.Module):
    """Weighted softmax attention (roost_message.py:286-317)."""

    def __init__(self, gate_nn, message_nn, num_heads=1):
        super().__init__()
        self.gate_nn = gate_nn
        self.message_nn = message_nn
        self.pow = torch.nn.Parameter(torch.randn((1)))

    def forward(self, fea, index, weights, plan=None, dim_size=None):
        if plan is None:
            n = int(index.max()) + 1 if dim_size is None else dim_size   # reference: scatter's implicit size
            plan = get_segment_plan(index, n)
        if (type(self.gate_nn) is SimpleNetwork and type(self.message_nn) is SimpleNetwork and fea.dim() == 2 and
                rowprog.eligible(fea)):
            # both networks read the same rows: one launch per direction for the pair (csrc/rowprog.hip)
            nets, flat = [], []
            for net in (self.gate_nn, self.message_nn):
                nets.append(rowprog.mlp_spec(len(net.fcs), _lib.ACT_LEAKY))
                for fc in list(net.fcs) + [net.fc_out]:
                    flat += [fc.weight, fc.bias, None]
            gate, fea = rowprog.RowNetsFn.apply(fea, None, tuple(nets), *flat)
        else:
            gate = self.gate_nn(fea)                                                   # [M,1]
            fea = self.message_nn(fea)
        # (weights ** pow) * exp(gate - segmax) / (segsum + 1e-13), times the message, summed per segment (305-317)
        return attention_pool(gate, fea, plan, index, mult=weights ** self.pow, eps=1e-13)

    def __repr__(self):
        return '{}(gate_nn={})'.format(self.__class__.__name__, self.gate_nn)


class MessageLayer(nn.Module):
    """Message passing on the composition graph (roost_message.py:88-153)."""

    def __init__(self, fea_len, num_heads=1):
        super().__init__()
        hidden_ele = [256]
        hidden_msg = [256]
        self.pooling = nn.ModuleList([WeightedAttention(
            gate_nn=SimpleNetwork(2 * fea_len, 1, hidden_ele),
            message_nn=SimpleNetwork(2 * fea_len, fea_len, hidden_msg),
        ) for _ in range(num_heads)])

    def forward(self, elem_weights, elem_in_fea, self_fea_idx, nbr_fea_idx, plans=None):
        n = elem_in_fea.shape[0]
        if plans is None:
            plans = (get_segment_plan(self_fea_idx, n), get_segment_plan(nbr_fea_idx, n))
        self_plan, nbr_plan = plans
        elem_nbr_weights = elem_weights.index_select(0, nbr_fea_idx)
        elem_nbr_fea = gather_rows(elem_in_fea, nbr_fea_idx, nbr_plan)
        elem_self_fea = gather_rows(elem_in_fea, self_fea_idx, self_plan)
        fea = torch.cat([elem_self_fea, elem_nbr_fea], dim=1)
        head_fea = [att(fea=fea, index=self_fea_idx, weights=elem_nbr_weights, plan=self_plan)
                    for att in self.pooling]
        fea = torch.mean(torch.stack(head_fea), dim=0)
        return fea + elem_in_fea

    def __repr__(self):
        return self.__class__.__name__


class Roost(nn.Module):
    """Composition model used for the global pooling context (roost_message.py:159-264)."""

    def __init__(self, orig_elem_fea_len, elem_fea_len, n_graph):
        super().__init__()
        self.embedding = nn.Linear(orig_elem_fea_len, elem_fea_len - 1)
        msg_heads = 1
        self.graphs = nn.ModuleList([MessageLayer(elem_fea_len, msg_heads) for _ in range(n_graph)])
        mat_heads = 1
        mat_hidden = [256]
        self.cry_pool = nn.ModuleList([WeightedAttention(
            gate_nn=SimpleNetwork(elem_fea_len, 1, mat_hidden),
            message_nn=nn.Identity(),
        ) for _ in range(mat_heads)])

    def forward(self, elem_weights, orig_elem_fea, self_fea_idx, nbr_fea_idx, crystal_elem_idx, num_crystals=None):
        from .ops import linear
        elem_fea = linear(orig_elem_fea, self.embedding.weight, self.embedding.bias)
        elem_fea = torch.cat([elem_fea, elem_weights], dim=1)                  # C-1 learned + the weight itself
        n = elem_fea.shape[0]
        plans = (get_segment_plan(self_fea_idx, n), get_segment_plan(nbr_fea_idx, n))   # shared by all message layers
        for graph_func in self.graphs:
            elem_fea = graph_func(elem_weights, elem_fea, self_fea_idx, nbr_fea_idx, plans=plans)
        G = int(crystal_elem_idx.max()) + 1 if num_crystals is None else num_crystals
        cplan = get_segment_plan(crystal_elem_idx, G)
        head_fea = [att(fea=elem_fea, index=crystal_elem_idx, weights=elem_weights, plan=cplan)
                    for att in self.cry_pool]
        return torch.mean(torch.stack(head_fea), dim=0)

    def __repr__(self):
        return self.__class__.__name__
