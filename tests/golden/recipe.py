"""Closed-form, RNG-light recipes shared by the golden-vector generator and the tests.

Everything here is *data recipe*, not model code: how parameters are filled, how the
tiny / BASELINE-shaped crystal graphs are laid out.  The generator
(`make_golden.py`, runs only in the authoring container where /root/reference exists)
applies these recipes to the *reference* modules; the tests apply the very same
recipes to the oracle (`oracle/cgat_oracle.py`) and to the HIP product
(`cgat_amd`) and compare against the stored reference outputs.

Input layout follows the reference's data layer:
  * edge_index[0] = centre atom, sorted, K consecutive entries per atom
    (reference CGAT/data.py:116-120,140), edge_index[1] = neighbour atom of the same crystal;
  * edge_attr = distance-rank shell id in [1, K] (reference CGAT/prepare_data.py:163-169);
  * roost tuple = (weights [Nc,1], elem_fea [Nc,200], self_idx [Mc], nbr_idx [Mc], crystal_idx [Nc])
    (reference CGAT/data.py:81-103, CGAT/roost_message.py:400-458).
"""
import math

import numpy as np
import torch

ORIG_FEA = 200  # matscholar embedding width used by the reference (lightning_module.py:166)


# ----------------------------------------------------------------------------------------
# parameters
# ----------------------------------------------------------------------------------------
def fill_params(module, dtype=None):
    """RNG-free parameter fill applied to a state_dict *in key order*.

    p.flat[i] = s * sin(0.37*i + 1.3*t + 0.11)   (t = position of the tensor in state_dict)
    s = 1/sqrt(fan_in) for >=2-d tensors (x0.1 for the hypernetwork head `net.4.weight`,
    mimicking last_hyper_layer_init), 0.1 for 1-d tensors; scalars get fixed values.
    Works on any module whose state_dict layout equals the reference's.
    """
    sd = module.state_dict()
    with torch.no_grad():
        for t, (name, p) in enumerate(sd.items()):
            n = p.numel()
            if name.endswith("damping"):
                # graphs.1 -> 0.3, graphs.2 -> 0.45 ... ; stays inside (0,1) so clamp is a no-op,
                val = 0.3 + 0.15 * (t % 4)
                p.copy_(torch.full_like(p, val))
                continue
            if name.endswith("pow"):
                p.copy_(torch.full_like(p, 0.7 - 0.2 * (t % 3)))
                continue
            if name.endswith("alpha") and n == 1:  # Rezero
                p.copy_(torch.full_like(p, 0.5 + 0.1 * (t % 3)))
                continue
            if p.dim() >= 2:
                fan_in = int(np.prod(p.shape[1:]))
                s = 1.0 / math.sqrt(fan_in)
                if name.endswith("net.4.weight"):
                    s *= 0.1
            else:
                s = 0.1
            i = torch.arange(n, dtype=torch.float64)
            v = s * torch.sin(0.37 * i + 1.3 * t + 0.11)
            p.copy_(v.reshape(p.shape).to(p.dtype))
    if dtype is not None:
        module.to(dtype)
    return module


def sin_tensor(shape, phase, scale=1.0, dtype=torch.float32):
    n = int(np.prod(shape))
    i = torch.arange(n, dtype=torch.float64)
    return (scale * torch.sin(0.73 * i + phase)).reshape(shape).to(dtype)


# ----------------------------------------------------------------------------------------
# graphs
# ----------------------------------------------------------------------------------------
class GraphBatch:
    """Duck-typed stand-in for torch_geometric.data.Batch: the fields CGAtNet.forward reads
    (reference CGAT/CGAT.py:566-570)."""

    def __init__(self, x, edge_index, edge_attr, batch, y=None):
        self.x, self.edge_index, self.edge_attr, self.batch, self.y = x, edge_index, edge_attr, batch, y
        self.num_nodes = x.shape[0]

    def to(self, device):
        return GraphBatch(*(None if t is None else t.to(device)
                            for t in (self.x, self.edge_index, self.edge_attr, self.batch, self.y)))


def element_table(n_elem=103, width=ORIG_FEA):
    """Synthetic stand-in for embeddings/matscholar-embedding.json with its statistics
    (103 x 200, mean 0.0035, std 0.0706, range [-0.247, 0.253]); closed form."""
    i = torch.arange(n_elem * width, dtype=torch.float64)
    t = 0.0035 + 0.0706 * math.sqrt(2.0) * torch.sin(1.917 * i + 0.5 * torch.sin(0.013 * i))
    return t.clamp(-0.247, 0.253).reshape(n_elem, width).to(torch.float32)


def build_graphs(atoms_per_graph, K, species_per_graph, seed, quirks=False):
    """atoms_per_graph: list of atom counts; species_per_graph: list of species counts.
    Returns (GraphBatch, roost_tuple, shell ids are in edge_attr)."""
    rs = np.random.RandomState(seed)
    table = element_table()
    xs, ei0, ei1, ea, bidx = [], [], [], [], []
    rw, rfea, rself, rnbr, rcry = [], [], [], [], []
    base = 0
    rbase = 0
    for g, (A, S) in enumerate(zip(atoms_per_graph, species_per_graph)):
        S = min(S, A)
        species = rs.choice(103, size=S, replace=False)
        z = np.concatenate([species, rs.choice(species, size=A - S)]) if A > S else species.copy()
        xs.append(table[torch.as_tensor(z, dtype=torch.long)])
        # exactly K out-edges per centre atom, neighbours drawn with replacement inside the crystal
        src = np.repeat(np.arange(A), K)
        dst = rs.randint(0, A, size=A * K)
        # shell ids: non-decreasing per atom, start at 1, +1 w.p. 0.4, clamp K
        inc = (rs.rand(A, K) < 0.4).astype(np.int64)
        inc[:, 0] = 0
        shell = np.minimum(1 + np.cumsum(inc, axis=1), K).reshape(-1)
        ei0.append(torch.as_tensor(src + base))
        ei1.append(torch.as_tensor(dst + base))
        ea.append(torch.as_tensor(shell))
        bidx.append(torch.full((A,), g, dtype=torch.long))
        # roost composition graph (reference data.py:81-103)
        uniq, counts = np.unique(z, return_counts=True)
        order = sorted(range(len(uniq)), key=lambda j: list(z).index(uniq[j]))  # first-appearance order
        uniq, counts = uniq[order], counts[order]
        ne = len(uniq)
        rw.append(torch.as_tensor(counts / float(A), dtype=torch.float32))
        rfea.append(table[torch.as_tensor(uniq, dtype=torch.long)])
        for i in range(ne):
            others = [j for j in range(ne) if j != i]
            rself.append(torch.full((len(others),), i + rbase, dtype=torch.long))
            rnbr.append(torch.as_tensor(others, dtype=torch.long) + rbase)
        rcry.append(torch.full((ne,), g, dtype=torch.long))
        base += A
        rbase += ne
    x = torch.cat(xs)
    ei = torch.stack([torch.cat(ei0), torch.cat(ei1)]).long()
    ea = torch.cat(ea).long()
    if quirks:
        # force the edge cases the path must survive: a node nobody points to, a self loop,
        # a duplicated (multi-)edge.  Graph 1 (>= 3 atoms) hosts them.
        a0 = atoms_per_graph[0]
        tgt = ei[1]
        n_iso = a0 + 1                      # second atom of graph 1 gets in-degree 0
        tgt[tgt == n_iso] = a0
        kk = a0 * K                         # first edge of graph 1's first atom
        tgt[kk] = ei[0][kk]                 # self loop
        tgt[kk + 1] = a0 + 2
        tgt[kk + 2] = a0 + 2                # duplicated edge
        ei = torch.stack([ei[0], tgt])
    batch = GraphBatch(x, ei, ea, torch.cat(bidx))
    N = x.shape[0]
    batch.y = sin_tensor((len(atoms_per_graph),), 0.4) * torch.as_tensor(atoms_per_graph, dtype=torch.float32)
    roost = (torch.cat(rw).view(-1, 1), torch.cat(rfea),
             torch.cat(rself) if rself else torch.zeros(0, dtype=torch.long),
             torch.cat(rnbr) if rnbr else torch.zeros(0, dtype=torch.long),
             torch.cat(rcry))
    return batch, roost


def tiny_graphs():
    """G=3 ragged crystals A in {2,5,7}, K=4; graph 0 is a single-element crystal (its roost
    graph has one node and no composition edges); graph 1 carries the quirks."""
    return build_graphs([2, 5, 7], K=4, species_per_graph=[1, 3, 2], seed=7, quirks=True)


def base_graphs(G=8, A=20, K=12, seed=0):
    rs = np.random.RandomState(seed + 1000)
    return build_graphs([A] * G, K=K, species_per_graph=list(rs.randint(2, 5, size=G)), seed=seed)


TINY = dict(C=16, Ce=16, H=3, K=4, L=2)
BASE = dict(C=128, Ce=128, H=3, K=12, L=4)


# ----------------------------------------------------------------------------------------
# case table: one definition, three users (reference generator, oracle test, HIP test)
# ----------------------------------------------------------------------------------------
class Case:
    """mk() -> module ; inputs(dtype) -> dict ; call(module, inputs) -> tensor."""

    def __init__(self, mk, inputs, call):
        self.mk, self.inputs, self.call = mk, inputs, call


def _net_call(**fkw):
    def call(m, i):
        b = GraphBatch(i["x"], i["edge_index"], i["edge_attr"], i["batch"])
        roost = (t for t in (i["r0"], i["r1"], i["r2"], i["r3"], i["r4"]))  # one-shot generator, as the harness passes
        return m(b, roost, **fkw)
    return call


def _net_inputs(b, roost):
    w, fea, sidx, nidx, cidx = roost

    def inp(dt):
        return {"x": b.x.to(dt), "edge_index": b.edge_index, "edge_attr": b.edge_attr, "batch": b.batch,
                "r0": w.to(dt), "r1": fea.to(dt), "r2": sidx, "r3": nidx, "r4": cidx}
    return inp


def tiny_cases(ns):
    """ns: namespace with the reference's class names (MultiHeadNetwork, GATConvNodes, ...;
    `RoostSimpleNetwork` = roost_message.SimpleNetwork)."""
    T = TINY
    C, Ce, H, K = T["C"], T["Ce"], T["H"], T["K"]
    D = 2 * C + Ce
    tb, troost = tiny_graphs()
    N, E = tb.num_nodes, tb.edge_index.shape[1]
    G = int(tb.batch.max()) + 1
    w, fea, sidx, nidx, cidx = troost
    Nc = w.shape[0]
    cases = {}

    # a1 MultiHeadNetwork (CGAT.py:65-112); view=False gets a non-contiguous input as MHAttention makes (55-58)
    def mhn(view):
        def inp(dt):
            return {"fea": sin_tensor((E, D), 0.2, 1.0, dt) if view else sin_tensor((2, E, D // 2), 0.2, 1.0, dt)}
        return Case(lambda: ns.MultiHeadNetwork(D, C, int(D / 1.5), H, view=view), inp,
                    lambda m, i: m(i["fea"] if view else i["fea"].transpose(1, 0)))
    cases["mhn_view1"], cases["mhn_view0"] = mhn(True), mhn(False)

    # a2-a8 GATConvNodes (CGAT.py:233-340)
    def node_inputs(dt):
        return {"x": sin_tensor((N, C), 0.1, 1.0, dt), "edge_index": tb.edge_index,
                "edge_attr": sin_tensor((E, Ce), 0.3, 1.0, dt), "x_0": sin_tensor((N, C), 0.5, 1.0, dt)}

    def nodes(first, vec):
        return Case(lambda: ns.GATConvNodes(C, C, Ce, H, concat=True, vector_attention=vec, first=first),
                    node_inputs, lambda m, i: m(i["x"], i["edge_index"], i["edge_attr"], i["x_0"]))
    for first in (True, False):
        for vec in (False, True):
            cases[f"nodes_first{int(first)}_vec{int(vec)}"] = nodes(first, vec)

    # a9 GATConvEdges (CGAT.py:115-230)
    def edge_inputs(dt):
        return {"x": sin_tensor((N, C), 0.1, 1.0, dt), "edge_index": tb.edge_index,
                "edge_attr": sin_tensor((E, Ce), 0.3, 1.0, dt), "x_0": sin_tensor((E, Ce), 0.6, 1.0, dt)}

    def edges(kw):
        return Case(lambda: ns.GATConvEdges(C, Ce, Ce, H, concat=True, **kw), edge_inputs,
                    lambda m, i: m(i["x"], i["edge_index"], i["edge_attr"], i["x_0"]))
    for tag, kw in [("nohyper", dict(no_hyper=True)), ("hyper_first", dict(no_hyper=False, first=True)),
                    ("hyper", dict(no_hyper=False)), ("hyper_vec", dict(no_hyper=False, vector_attention=True))]:
        cases[f"edges_{tag}"] = edges(kw)

    # a8 hypernetworks (Hypernetworksmp.py:257-313)
    cases["hnet0"] = Case(lambda: ns.H_Net_0(C, 3, C, C, 2, C, C),
                          lambda dt: {"h_0": sin_tensor((N, C), 0.1, 1.0, dt), "x": sin_tensor((N, C), 0.7, 1.0, dt)},
                          lambda m, i: m(i["h_0"], i["x"]))
    cases["hnet"] = Case(lambda: ns.H_Net(C, 3, C, C, 2, C, C),
                         lambda dt: {"h_0": sin_tensor((N, C), 0.1, 1.0, dt), "h_t": sin_tensor((N, C), 0.2, 1.0, dt),
                                     "x": sin_tensor((N, C), 0.7, 1.0, dt)},
                         lambda m, i: m(i["h_0"], i["h_t"], i["x"]))

    # a12 MHAttention (CGAT.py:14-62)
    def mhatt(vec):
        return Case(lambda: ns.MHAttention(C, C, H, vector_attention=vec),
                    lambda dt: {"fea": sin_tensor((N, C), 0.15, 1.0, dt), "cry_fea": sin_tensor((G, C), 0.25, 1.0, dt),
                                "index": tb.batch},
                    lambda m, i: m(i["fea"], i["cry_fea"], i["index"]))
    cases["mhatt_vec0"], cases["mhatt_vec1"] = mhatt(False), mhatt(True)

    # a13 ResidualNetwork (message_changed.py:81-138)
    def resnet(rez, last):
        return Case(lambda: ns.ResidualNetwork(C, 2, [32, 32, 24, 24, 16], if_rezero=rez),
                    lambda dt: {"fea": sin_tensor((G, C), 0.35, 1.0, dt)},
                    lambda m, i: m(i["fea"], last_layer=last))
    for rez in (False, True):
        for last in (True, False):
            cases[f"resnet_rez{int(rez)}_last{int(last)}"] = resnet(rez, last)

    # a10 SimpleNetwork
    cases["simplenet"] = Case(lambda: ns.SimpleNetwork(Ce, Ce, [Ce]),
                              lambda dt: {"fea": sin_tensor((E, Ce), 0.45, 1.0, dt)}, lambda m, i: m(i["fea"]))

    # a14 Roost branch (roost_message.py:88-321)
    cases["wattn"] = Case(
        lambda: ns.WeightedAttention(gate_nn=ns.RoostSimpleNetwork(2 * C, 1, [24]),
                                     message_nn=ns.RoostSimpleNetwork(2 * C, C, [24])),
        lambda dt: {"fea": sin_tensor((sidx.shape[0], 2 * C), 0.55, 1.0, dt), "index": sidx, "weights": w[nidx].to(dt)},
        lambda m, i: m(fea=i["fea"], index=i["index"], weights=i["weights"]))
    cases["msglayer"] = Case(
        lambda: ns.MessageLayer(C, 1),
        lambda dt: {"elem_weights": w.to(dt), "elem_in_fea": sin_tensor((Nc, C), 0.65, 1.0, dt),
                    "self_fea_idx": sidx, "nbr_fea_idx": nidx},
        lambda m, i: m(i["elem_weights"], i["elem_in_fea"], i["self_fea_idx"], i["nbr_fea_idx"]))
    cases["roost"] = Case(
        lambda: ns.Roost(ORIG_FEA, C, 3),
        lambda dt: {"elem_weights": w.to(dt), "orig_elem_fea": fea.to(dt), "self_fea_idx": sidx,
                    "nbr_fea_idx": nidx, "crystal_elem_idx": cidx},
        lambda m, i: m(i["elem_weights"], i["orig_elem_fea"], i["self_fea_idx"], i["nbr_fea_idx"],
                       i["crystal_elem_idx"]))

    # a11 full CGAtNet (CGAT.py:343-613)
    variants = {
        "net_mean": (dict(mean_pooling=True), {}),
        "net_concat": (dict(mean_pooling=False), {}),
        "net_embed": (dict(mean_pooling=True), dict(return_graph_embedding=True)),
        "net_hidden": (dict(mean_pooling=False, rezero=True), dict(last_layer=False)),
        "net_vec": (dict(mean_pooling=False, vector_attention=True, global_vector_attention=True), {}),
        "net_edgehyper": (dict(mean_pooling=True, no_hyper=False), {}),
    }

    def net(ckw, fkw):
        return Case(lambda: ns.CGAtNet(ORIG_FEA, C, T["L"], nbr_embedding_size=Ce, neighbor_number=K, msg_heads=H,
                                       update_edges=True, n_graph_roost=2, **ckw),
                    _net_inputs(tb, troost), _net_call(**fkw))
    for tag, (ckw, fkw) in variants.items():
        cases[tag] = net(ckw, fkw)
    return cases


def base_cases(ns):
    """BASELINE-shaped: C=Ce=128, H=3, K=12, L=4, G=8 crystals of 20 atoms (N=160, E=1920)."""
    B = BASE
    C, Ce, H, K, L = B["C"], B["Ce"], B["H"], B["K"], B["L"]
    bb, broost = base_graphs(G=8, A=20, K=K, seed=0)
    N, E = bb.num_nodes, bb.edge_index.shape[1]
    cases = {}

    def node_inputs(dt):
        return {"x": sin_tensor((N, C), 0.1, 1.0, dt), "edge_index": bb.edge_index,
                "edge_attr": sin_tensor((E, Ce), 0.3, 1.0, dt), "x_0": sin_tensor((N, C), 0.5, 1.0, dt)}

    def nodes(first):
        return Case(lambda: ns.GATConvNodes(C, C, Ce, H, concat=True, first=first), node_inputs,
                    lambda m, i: m(i["x"], i["edge_index"], i["edge_attr"], i["x_0"]))
    cases["nodes_first1"], cases["nodes_first0"] = nodes(True), nodes(False)
    cases["net_mean"] = Case(
        lambda: ns.CGAtNet(ORIG_FEA, C, L, nbr_embedding_size=Ce, neighbor_number=K, msg_heads=H, update_edges=True,
                           mean_pooling=True, n_graph_roost=3),
        _net_inputs(bb, broost), _net_call())
    return cases


def cotangent(y):
    return sin_tensor(tuple(y.shape), 0.9, dtype=y.dtype)


def grad_probe(g):
    """Compact fingerprint of a large gradient: (sum, L2 norm, max|.|, dot with a sin ramp,
    first 8 entries), computed in float64."""
    g = g.detach().double().flatten().cpu()
    ramp = torch.sin(0.011 * torch.arange(g.numel(), dtype=torch.float64) + 0.3)
    return np.array([g.sum(), g.norm(), g.abs().max(), (g * ramp).sum()] + g[:8].tolist())


PROBE_ABOVE = 4096


def run_case(case, dtype=torch.float32, device="cpu", want_grads=True, ctx=None):
    """Build the module, fill its parameters by recipe, run forward and the backward of
    sum(out * cotangent).  Returns (out, {name: grad or None}) with names `gin.<input>` and
    `gp.<param>`; the module's parameter names must equal the reference's."""
    torch.manual_seed(1)
    mod = fill_params(case.mk()).to(dtype).to(device)
    inputs = {k: (v.to(device) if torch.is_tensor(v) else v) for k, v in case.inputs(dtype).items()}
    leaves = {k: v for k, v in inputs.items() if torch.is_tensor(v) and v.is_floating_point()}
    for v in leaves.values():
        v.requires_grad_(True)
    if ctx is None:
        y = case.call(mod, inputs)
    else:
        with ctx(mod):                 # tests: record / force the activation derivative patterns around the forward
            y = case.call(mod, inputs)
    if not want_grads:
        return y, {}, mod
    params = dict(mod.named_parameters())
    targets = list(leaves.values()) + list(params.values())
    names = ["gin." + k for k in leaves] + ["gp." + k for k in params]
    grads = torch.autograd.grad((y * cotangent(y).to(device)).sum(), targets, allow_unused=True)
    return y, dict(zip(names, grads)), mod
