"""Batch container and the seeded synthetic crystal-graph generator used by bench.py and the
tests.  Layout mirrors what the reference's data layer hands to CGAtNet:

  x          [N,200] fp32   element embedding of each atom            (CGAT/data.py:107-114)
  edge_index [2,E]   int64  row 0 = centre atom (sorted, exactly K per atom), row 1 = neighbour
                            of the same crystal                         (CGAT/data.py:116-120,140)
  edge_attr  [E]     int64  distance-rank shell id in [1,K]            (CGAT/prepare_data.py:163-169)
  batch      [N]     int64  crystal index of each atom (sorted)
  roost      5-tuple        composition graph                          (CGAT/data.py:81-103,
                                                                         CGAT/roost_message.py:400-458)
Everything is generated with vectorised torch ops from a torch.Generator seed (no per-graph
Python loops), so million-edge batches build in well under a second.
"""
import math

import torch

ORIG_FEA = 200
N_ELEMENTS = 103


class GraphBatch:
    """The fields of torch_geometric.data.Batch that CGAtNet.forward reads (CGAT.py:566-570)."""

    def __init__(self, x, edge_index, edge_attr, batch, y=None, num_graphs=None):
        self.x, self.edge_index, self.edge_attr, self.batch, self.y = x, edge_index, edge_attr, batch, y
        self.num_nodes = x.shape[0]
        self.num_graphs = num_graphs

    def to(self, device):
        mv = lambda t: None if t is None else t.to(device)
        return GraphBatch(mv(self.x), mv(self.edge_index), mv(self.edge_attr), mv(self.batch), mv(self.y),
                          self.num_graphs)


def element_table(seed=1234):
    """103 x 200 table with the statistics of the matscholar embedding the reference ships
    (mean 0.0035, std 0.0706, clipped to [-0.247, 0.253]); synthetic, seeded."""
    g = torch.Generator().manual_seed(seed)
    t = 0.0035 + 0.0706 * torch.randn(N_ELEMENTS, ORIG_FEA, generator=g)
    return t.clamp_(-0.247, 0.253)


def synthetic_batch(num_graphs, atoms_per_graph=20, K=12, seed=0, species_range=(2, 4)):
    """Returns (GraphBatch, roost_tuple) on CPU.  E = num_graphs * atoms_per_graph * K."""
    g = torch.Generator().manual_seed(seed)
    G, A = int(num_graphs), int(atoms_per_graph)
    N = G * A
    table = element_table()
    # species: each crystal draws S in [lo,hi] distinct-ish elements, atoms pick among them
    lo, hi = species_range
    S = torch.randint(lo, hi + 1, (G,), generator=g)
    pool = torch.randint(0, N_ELEMENTS, (G, hi), generator=g)
    pool = (pool + torch.arange(hi).view(1, -1) * 17) % N_ELEMENTS        # decorrelate columns
    pick = (torch.rand(G, A, generator=g) * S.view(-1, 1)).long().clamp_(max=hi - 1)
    pick[:, :hi] = torch.minimum(torch.arange(hi).view(1, -1).expand(G, -1), (S - 1).view(-1, 1))  # every species used
    z = pool.gather(1, pick)                                              # [G,A] element ids
    x = table[z.reshape(-1)]
    # edges
    centre = torch.arange(N).repeat_interleave(K)
    nbr_local = torch.randint(0, A, (N * K,), generator=g)
    nbr = (centre // A) * A + nbr_local
    edge_index = torch.stack([centre, nbr])
    inc = (torch.rand(N, K, generator=g) < 0.4).long()
    inc[:, 0] = 0
    shell = (1 + inc.cumsum(dim=1)).clamp_(max=K).reshape(-1)
    batch = torch.arange(N) // A
    y = torch.randn(G, generator=g) * A
    # roost composition graph: unique species per crystal with fractional weights, fully connected w/o self edges
    onehot = torch.zeros(G, N_ELEMENTS)
    onehot.scatter_add_(1, z, torch.ones(G, A))
    present = onehot > 0
    cry_idx, elem_id = present.nonzero(as_tuple=True)                     # sorted by crystal
    weights = (onehot[cry_idx, elem_id] / A).view(-1, 1)
    n_per = present.sum(dim=1)                                            # species per crystal
    start = torch.cumsum(n_per, 0) - n_per
    Nc = int(cry_idx.numel())
    local = torch.arange(Nc) - start[cry_idx]
    # pairs (i, j != i) inside each crystal
    reps = (n_per[cry_idx] - 1).clamp_(min=0)
    self_idx = torch.arange(Nc).repeat_interleave(reps)
    kk = torch.arange(int(reps.sum())) - (torch.cumsum(reps, 0) - reps).repeat_interleave(reps)
    loc_self = local[self_idx]
    nbr_local2 = kk + (kk >= loc_self).long()
    nbr_idx = start[cry_idx[self_idx]] + nbr_local2
    roost = (weights, table[elem_id], self_idx, nbr_idx, cry_idx)
    return GraphBatch(x, edge_index, shell, batch, y, num_graphs=G), roost


def shard_graphs(num_graphs, rank, world_size):
    """Contiguous, balanced split of crystal ids across ranks (graphs are independent units:
    no edge crosses crystals, so sharding needs no halo and no data-path collective)."""
    base, rem = divmod(num_graphs, world_size)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


# element symbols in the order of the reference's embeddings/matscholar-embedding.json (Z = 1 .. 103)
ELEMENT_SYMBOLS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb "
                   "Sr Y Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf "
                   "Ta W Re Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr").split()


def synthetic_dataset_dict(num_graphs, atoms_per_graph=20, stored_nbrs=24, seed=0):
    """A dataset in the reference's on-disk dictionary layout 0 (prepare_data.py:92-98; data.py:47-50):
    {'input': object array [3, n] of per-crystal [n_atoms, 24] tables (shell id, centre, neighbour), 'comps': per-atom
    element symbols, 'batch_comp': formula strings, 'target': {name: array}} and the embedding dict symbol -> vector.
    Used by bench.py --workload collate and the collation tests."""
    import numpy as np
    rs = np.random.RandomState(seed)
    if isinstance(atoms_per_graph, (tuple, list)):
        return _ragged_dataset_dict(rs, int(num_graphs), int(atoms_per_graph[0]), int(atoms_per_graph[1]), int(stored_nbrs))
    G, A, K = int(num_graphs), int(atoms_per_graph), int(stored_nbrs)
    z = rs.randint(0, N_ELEMENTS, size=(G, 1)) + rs.randint(0, 4, size=(G, A)) * 17
    z %= N_ELEMENTS
    shell = np.minimum(1 + np.cumsum(rs.rand(G, A, K) < 0.4, axis=2), 12).astype(np.int64)   # shell ids clamped as prepare_data.py:163-169
    centre = np.broadcast_to(np.arange(A)[None, :, None], (G, A, K)).astype(np.int64)
    nbr = rs.randint(0, A, size=(G, A, K)).astype(np.int64)
    inp = np.empty((3, G), dtype=object)
    comps, formulas = [], []
    for g in range(G):
        inp[0][g], inp[1][g], inp[2][g] = shell[g], centre[g], nbr[g]
        syms = [ELEMENT_SYMBOLS[k] for k in z[g]]
        comps.append(syms)
        formulas.append("".join(f"{el}{syms.count(el)}" for el in dict.fromkeys(syms)))
    table = element_table().numpy()
    emb = {el: table[k].astype("float64").tolist() for k, el in enumerate(ELEMENT_SYMBOLS)}
    return {"input": inp, "comps": comps, "batch_comp": formulas,
            "target": {"e_above_hull": rs.randn(G).round(4)}}, emb


def _ragged_dataset_dict(rs, G, a_lo, a_hi, K):
    """The DCGAT-shaped variant (BASELINE configs[3]): crystals of a_lo .. a_hi atoms (uniform), K stored neighbours per
    atom with periodic images (an atom of a 2-atom cell has 24 neighbours among 2 atoms: multi-edges and self loops)."""
    import numpy as np
    inp = np.empty((3, G), dtype=object)
    comps, formulas = [], []
    for g in range(G):
        A = int(rs.randint(a_lo, a_hi + 1))
        z = (rs.randint(0, N_ELEMENTS) + rs.randint(0, 4, size=A) * 17) % N_ELEMENTS
        inp[0][g] = np.minimum(1 + np.cumsum(rs.rand(A, K) < 0.4, axis=1), 12).astype(np.int64)
        inp[1][g] = np.broadcast_to(np.arange(A)[:, None], (A, K)).astype(np.int64)
        inp[2][g] = rs.randint(0, A, size=(A, K)).astype(np.int64)
        syms = [ELEMENT_SYMBOLS[k] for k in z]
        comps.append(syms)
        formulas.append("".join(f"{el}{syms.count(el)}" for el in dict.fromkeys(syms)))
    table = element_table().numpy()
    emb = {el: table[k].astype("float64").tolist() for k, el in enumerate(ELEMENT_SYMBOLS)}
    return {"input": inp, "comps": comps, "batch_comp": formulas,
            "target": {"e_above_hull": rs.randn(G).round(4)}}, emb
