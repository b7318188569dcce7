"""Synthetic datasets in the reference's on-disk dictionary format (reference prepare_data.py:92-98,
CGAT/data.py:47-50) for the collation fixtures and tests.  Closed form / seeded: the same data is rebuilt by the
fixture generator, the oracle test and the GPU test."""
import json
import os

import numpy as np

# element symbols in the order of embeddings/matscholar-embedding.json (Z = 1 .. 103)
ELEMENTS = ("H He Li Be B C N O F Ne Na Mg Al Si P S Cl Ar K Ca Sc Ti V Cr Mn Fe Co Ni Cu Zn Ga Ge As Se Br Kr Rb Sr Y "
            "Zr Nb Mo Tc Ru Rh Pd Ag Cd In Sn Sb Te I Xe Cs Ba La Ce Pr Nd Pm Sm Eu Gd Tb Dy Ho Er Tm Yb Lu Hf Ta W Re "
            "Os Ir Pt Au Hg Tl Pb Bi Po At Rn Fr Ra Ac Th Pa U Np Pu Am Cm Bk Cf Es Fm Md No Lr").split()
N_NBR_STORED = 24          # neighbour columns stored per atom (prepare_data.py keeps 24)
FEA = 200                  # embedding width of the reference's matscholar table


def embedding_table():
    """103 x 200 table with the statistics of the reference's embedding file; closed form (float64 values that are
    exactly representable in fp32, so that JSON round trips and torch.Tensor conversion are exact)."""
    i = np.arange(len(ELEMENTS) * FEA, dtype=np.float64)
    t = 0.0035 + 0.0706 * np.sqrt(2.0) * np.sin(1.917 * i + 0.5 * np.sin(0.013 * i))
    return np.clip(t, -0.247, 0.253).astype(np.float32).reshape(len(ELEMENTS), FEA)


def write_embedding_json(path):
    tab = embedding_table()
    with open(path, "w") as f:
        json.dump({el: [float(v) for v in tab[k]] for k, el in enumerate(ELEMENTS)}, f)
    return path


def make_dataset(n_graphs, seed, fmt=0, comps_as="list"):
    """Dictionary with the keys the reference's CompositionData expects.  Ragged crystals (1..40 atoms, incl.
    single-element ones), neighbour tables [n_atoms, 24] = (shell id, self index, neighbour index).
    fmt 0: data['input'] is an object array [3, n_graphs]; fmt 1: [n_graphs, 3]  (data.py:47-50).
    comps_as: 'list' (per-atom symbols), 'tuple' (per-atom (symbol, site) tuples) or 'str' (parsed from batch_comp)."""
    rs = np.random.RandomState(seed)
    shells, selfs, nbrs, comps, batch_comp, targets = [], [], [], [], [], []
    for g in range(n_graphs):
        n_atoms = int(rs.randint(1, 41)) if g % 7 else int(rs.randint(1, 4))
        n_species = int(rs.randint(1, min(n_atoms, 5) + 1))
        species = rs.choice(len(ELEMENTS), size=n_species, replace=False)
        z = np.concatenate([species, rs.choice(species, size=n_atoms - n_species)])
        if comps_as == "str":
            z = np.sort(z)       # a formula string can only describe grouped elements
        rs.shuffle(z) if comps_as != "str" else None
        sh = np.cumsum(rs.rand(n_atoms, N_NBR_STORED) < 0.4, axis=1) + 1
        shells.append(sh.astype(np.int64))
        selfs.append(np.repeat(np.arange(n_atoms)[:, None], N_NBR_STORED, axis=1).astype(np.int64))
        nbrs.append(rs.randint(0, n_atoms, size=(n_atoms, N_NBR_STORED)).astype(np.int64))
        syms = [ELEMENTS[k] for k in z]
        # formula in order of first appearance
        uniq = list(dict.fromkeys(syms))
        formula = "".join(f"{el}{syms.count(el)}" for el in uniq)
        batch_comp.append(formula)
        if comps_as == "list":
            comps.append(syms)
        elif comps_as == "tuple":
            comps.append([(el, i) for i, el in enumerate(syms)])
        else:
            comps.append(formula)
        targets.append(float(np.round(rs.randn(), 4)))
    inp = np.empty((3, n_graphs), dtype=object)
    for g in range(n_graphs):
        inp[0][g], inp[1][g], inp[2][g] = shells[g], selfs[g], nbrs[g]
    if fmt == 1:
        inp = np.ascontiguousarray(inp.T)
        assert inp.shape[0] > 3
    return {"input": inp, "comps": comps, "batch_comp": batch_comp,
            "target": {"e_above_hull": np.array(targets), "volume": np.abs(np.array(targets)) + 1.0}}


CASES = [  # name, n_graphs, seed, fmt, comps_as, max_nbr, target, batches (lists of dataset indices)
    ("fmt0_list_k12", 23, 1, 0, "list", 12, "e_above_hull", [[0, 1, 2, 3], [22, 5, 5, 7, 11, 13], list(range(23))]),
    ("fmt1_tuple_k24", 9, 2, 1, "tuple", 24, "e_above_hull", [[8, 0, 4], list(range(9))]),
    ("fmt0_str_k6_volume", 11, 3, 0, "str", 6, "volume", [[1, 2, 3, 10], [6]]),
]
