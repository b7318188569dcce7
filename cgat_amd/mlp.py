"""SimpleNetwork / Rezero / ResidualNetwork with the reference's parameter layout
(reference CGAT/message_changed.py:31-138, CGAT/roost_message.py:324-355); every Linear runs in
the fp32 MFMA GEMM kernel with its activation fused into the epilogue."""
import torch
import torch.nn as nn

from . import _lib, rowprog
from .ops import ChainMLPFn, linear


class SimpleNetwork(nn.Module):
    """Linear -> LeakyReLU(0.01) per hidden layer, then Linear (message_changed.py:36-63)."""

    def __init__(self, input_dim, output_dim, hidden_layer_dims):
        super().__init__()
        dims = [input_dim] + list(hidden_layer_dims)
        self.fcs = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.acts = nn.ModuleList([nn.LeakyReLU() for _ in range(len(dims) - 1)])
        self.fc_out = nn.Linear(dims[-1], output_dim)

    def forward(self, fea, residual=None):
        """`residual` (optional, same shape as the output) is added to the result: CGAtNet's `edge_attr + Edge(...)`."""
        ws = [fc.weight for fc in self.fcs] + [self.fc_out.weight]
        bs = [fc.bias for fc in self.fcs] + [self.fc_out.bias]
        x2 = fea.reshape(-1, fea.shape[-1])
        if rowprog.eligible(x2):
            # a few hundred rows (the composition branch, per-crystal rows): the whole network is one launch per direction
            r2 = None if residual is None else residual.reshape(x2.shape[0], -1)
            spec = (rowprog.mlp_spec(len(self.fcs), _lib.ACT_LEAKY),)
            flat = [t for w, b in zip(ws, bs) for t in (w, b, None)]
            (out,) = rowprog.RowNetsFn.apply(x2, r2, spec, *flat)
            return out.reshape(*fea.shape[:-1], out.shape[-1])
        if all(b is not None for b in bs) and ChainMLPFn.eligible(x2, ws, None if residual is None else residual.reshape(x2.shape)):
            # all layers of width 128: one launch per direction, hidden rows never leave the registers (csrc/chain.hip)
            r2 = None if residual is None else residual.reshape(x2.shape)
            out = ChainMLPFn.apply(x2, r2, _lib.ACT_LEAKY, *ws, *bs)
            return out.reshape(*fea.shape[:-1], 128)
        for fc in self.fcs:
            fea = linear(fea, fc.weight, fc.bias, _lib.ACT_LEAKY)
        out = linear(fea, self.fc_out.weight, self.fc_out.bias)
        return out if residual is None else out + residual

    def __repr__(self):
        return self.__class__.__name__


class Rezero(nn.Module):
    def __init__(self):
        super().__init__()
        self.alpha = nn.Parameter(torch.zeros(1))

    def forward(self, x):
        return self.alpha * x

    def __repr__(self):
        return self.__class__.__name__


class ResidualNetwork(nn.Module):
    """fea = [rezero](relu(fc(fea))) + res_fc(fea) per layer; `last_layer=False` returns the last
    hidden representation (message_changed.py:86-135)."""

    def __init__(self, input_dim, output_dim, hidden_layer_dims, if_rezero=False):
        super().__init__()
        dims = [input_dim] + list(hidden_layer_dims)
        self.fcs = nn.ModuleList([nn.Linear(dims[i], dims[i + 1]) for i in range(len(dims) - 1)])
        self.res_fcs = nn.ModuleList([nn.Linear(dims[i], dims[i + 1], bias=False) if dims[i] != dims[i + 1]
                                      else nn.Identity() for i in range(len(dims) - 1)])
        self.acts = nn.ModuleList([nn.ReLU() for _ in range(len(dims) - 1)])
        self.fc_out = nn.Linear(dims[-1], output_dim)
        self.if_rezero = if_rezero
        if self.if_rezero:
            self.rezeros = nn.ModuleList([Rezero() for _ in range(len(dims) - 1)])

    def forward(self, fea, *, last_layer=True):
        x2 = fea.reshape(-1, fea.shape[-1])
        if not self.if_rezero and rowprog.eligible(x2):
            # the head at G rows: every layer with its skip product is one phase of ONE launch per direction
            spec, flat = [], []
            for fc, res in zip(self.fcs, self.res_fcs):
                ident = isinstance(res, nn.Identity)
                spec.append((_lib.ACT_RELU, 1 if ident else 2, True))
                flat += [fc.weight, fc.bias, None if ident else res.weight]
            if last_layer:
                spec.append((_lib.ACT_NONE, 0, True))
                flat += [self.fc_out.weight, self.fc_out.bias, None]
            (out,) = rowprog.RowNetsFn.apply(x2, None, (tuple(spec),), *flat)
            return out.reshape(*fea.shape[:-1], out.shape[-1])
        for k, (fc, res) in enumerate(zip(self.fcs, self.res_fcs)):
            h = linear(fea, fc.weight, fc.bias, _lib.ACT_RELU)
            if self.if_rezero:
                h = self.rezeros[k](h)
            skip = fea if isinstance(res, nn.Identity) else linear(fea, res.weight, None)
            fea = h + skip
        return linear(fea, self.fc_out.weight, self.fc_out.bias) if last_layer else fea

    def __repr__(self):
        return self.__class__.__name__
