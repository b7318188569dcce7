#!/usr/bin/env python3
"""Generate golden vectors by running the UNMODIFIED reference (hyllios/CGAT at
/root/reference) on CPU.  Runs only in the authoring container; the resulting
`*.npz` fixtures are committed, the reference never travels.

    python tests/golden/make_golden.py            # writes tests/golden/*.npz

The reference's hot path imports three packages that are not installed and not vendored
(torch_scatter, torch_geometric, torchvision).  Only four symbols of them are used on the
path; they are stood in for below with pure-torch definitions of their *documented*
semantics:

  * torch_scatter.scatter_add / scatter_max / scatter_mean       (segment reductions, dim=0)
  * torch_geometric.utils.softmax                                 exp(a-max)/(sum+1e-16)
  * torch_geometric.nn.MessagePassing.propagate (flow source_to_target, aggr='add', node_dim=0)
        x_j = x[edge_index[0]], x_i = x[edge_index[1]], aggregate at edge_index[1]
  * torchvision.utils                                             (imported, never used)

Everything else that runs -- MultiHeadNetwork, GATConvNodes.message/update, GATConvEdges,
MHAttention, CGAtNet.forward, every class of Hypernetworksmp.py, message_changed.py and
roost_message.py -- is the reference's own code.  The reference has no tests, so there is
no upstream vector pinning the third-party boundary itself; that boundary is pinned only by
the documented semantics restated here.
"""
import inspect
import os
import sys
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import recipe  # noqa: E402

REF = "/root/reference"


# ----------------------------------------------------------------------------------------
# stand-ins for the absent third-party packages
# ----------------------------------------------------------------------------------------
def _expand(index, src):
    return index.view(-1, *([1] * (src.dim() - 1))).expand_as(src)


def scatter_add(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    size = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.zeros((size,) + tuple(src.shape[1:]), dtype=src.dtype)
    return res.scatter_add_(0, _expand(index, src), src)


def scatter_max(src, index, dim=0, out=None, dim_size=None):
    assert dim == 0
    size = int(index.max()) + 1 if dim_size is None else dim_size
    res = torch.zeros((size,) + tuple(src.shape[1:]), dtype=src.dtype)
    res = res.scatter_reduce(0, _expand(index, src), src, reduce="amax", include_self=False)
    return res, None


def scatter_mean(src, index, dim=0, out=None, dim_size=None):
    s = scatter_add(src, index, dim, None, dim_size)
    c = scatter_add(torch.ones_like(src), index, dim, None, dim_size).clamp(min=1)
    return s / c


def pyg_softmax(src, index, ptr=None, num_nodes=None):
    n = int(index.max()) + 1 if num_nodes is None else num_nodes
    mx = scatter_max(src, index, 0, None, n)[0][index]
    ex = (src - mx).exp()
    den = scatter_add(ex, index, 0, None, n)[index]
    return ex / (den + 1e-16)


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr="add", flow="source_to_target", node_dim=-2, **kw):
        super().__init__()
        assert aggr == "add" and flow == "source_to_target"
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim

    def propagate(self, edge_index, size=None, **kwargs):
        assert self.node_dim == 0
        j, i = edge_index[0], edge_index[1]
        x = kwargs["x"]
        n = x.shape[0]
        margs = {}
        for name in inspect.signature(self.message).parameters:
            if name == "edge_index_i":
                margs[name] = i
            elif name == "edge_index_j":
                margs[name] = j
            elif name.endswith("_i"):
                margs[name] = kwargs[name[:-2]].index_select(0, i)
            elif name.endswith("_j"):
                margs[name] = kwargs[name[:-2]].index_select(0, j)
            else:
                margs[name] = kwargs[name]
        out = self.message(**margs)
        agg = scatter_add(out, i, 0, None, n)
        uargs = {k: kwargs[k] for k in list(inspect.signature(self.update).parameters)[1:]}
        return self.update(agg, **uargs)


def install_shims():
    ts = types.ModuleType("torch_scatter")
    ts.scatter_add, ts.scatter_max, ts.scatter_mean = scatter_add, scatter_max, scatter_mean
    tg = types.ModuleType("torch_geometric")
    tgn = types.ModuleType("torch_geometric.nn")
    tgn.MessagePassing = MessagePassing
    tgu = types.ModuleType("torch_geometric.utils")
    tgu.softmax = pyg_softmax
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = object
    tgd.Batch = object
    tg.nn, tg.utils, tg.data = tgn, tgu, tgd
    tv = types.ModuleType("torchvision")
    tvu = types.ModuleType("torchvision.utils")
    tv.utils = tvu
    for name, mod in [("torch_scatter", ts), ("torch_geometric", tg), ("torch_geometric.nn", tgn),
                      ("torch_geometric.utils", tgu), ("torch_geometric.data", tgd),
                      ("torchvision", tv), ("torchvision.utils", tvu)]:
        sys.modules[name] = mod


def import_reference():
    install_shims()
    sys.path.insert(0, REF)
    # CGAT/__init__.py pulls in CGAT.CGAT only
    import CGAT.CGAT as ref_cgat
    import CGAT.Hypernetworksmp as ref_hyper
    import CGAT.message_changed as ref_mlp
    import CGAT.roost_message as ref_roost
    return ref_cgat, ref_hyper, ref_mlp, ref_roost


# ----------------------------------------------------------------------------------------
# driver
# ----------------------------------------------------------------------------------------
def _np(t):
    return t.detach().cpu().numpy()


def record(case, *, full_param_grads):
    """fp32 run of the reference (outputs, gradients) + fp64 run of the same.  For every tensor
    the fp64 run yields the reference's own rounding deviation `nf.<name>` = ||fp32 - fp64||_inf:
    the noise floor below which a difference from the fp32 reference carries no information."""
    out = {}
    y, grads, _ = recipe.run_case(case, torch.float32)
    y64, grads64, _ = recipe.run_case(case, torch.float64)
    out["out"] = _np(y)
    out["out_f64"] = _np(y64)
    for name, g in grads.items():
        if g is None:
            out[name + ".none"] = np.zeros(0, dtype=np.float32)
            continue
        g64 = grads64[name]
        out["nf." + name] = np.array([(g.double() - g64).abs().max().item(), g64.abs().max().item()])
        if name.startswith("gp.") and (g.numel() > recipe.PROBE_ABOVE or not full_param_grads):
            out["gpn." + name[3:]] = recipe.grad_probe(g)
            out["gpn64." + name[3:]] = recipe.grad_probe(g64)
        else:
            out[name] = _np(g)
    return out


def main():
    ref_cgat, ref_hyper, ref_mlp, ref_roost = import_reference()
    ns = types.SimpleNamespace(
        MultiHeadNetwork=ref_cgat.MultiHeadNetwork, GATConvNodes=ref_cgat.GATConvNodes,
        GATConvEdges=ref_cgat.GATConvEdges, MHAttention=ref_cgat.MHAttention, CGAtNet=ref_cgat.CGAtNet,
        H_Net_0=ref_hyper.H_Net_0, H_Net=ref_hyper.H_Net,
        SimpleNetwork=ref_mlp.SimpleNetwork, ResidualNetwork=ref_mlp.ResidualNetwork,
        WeightedAttention=ref_roost.WeightedAttention, MessageLayer=ref_roost.MessageLayer,
        Roost=ref_roost.Roost, RoostSimpleNetwork=ref_roost.SimpleNetwork)
    for fname, table, full in (("tiny.npz", recipe.tiny_cases(ns), True), ("base.npz", recipe.base_cases(ns), False)):
        blob = {}
        for cname, case in table.items():
            for k, v in record(case, full_param_grads=full).items():
                blob[f"{cname}/{k}"] = v
        np.savez_compressed(os.path.join(HERE, fname), **blob)
        print(fname, len(table), "cases", len(blob), "arrays")


if __name__ == "__main__":
    main()
