"""Device-side batch collation: the step right before the hot path.

The reference assembles every batch on the host with per-crystal Python loops
(`CompositionData.__getitem__`, CGAT/data.py:61-144; PyG `Batch.from_data_list`, CGAT/lightning_module.py:200;
`collate_batch`, CGAT/roost_message.py:400-458) and ships it to the GPU each step.  Here the dataset dictionary is
packed ONCE into int32 arrays that live in HBM (`PackedDataset`), and a batch is a list of crystal ids:
`PackedDataset.collate(ids)` launches one HIP kernel (csrc/collate.hip) that writes the tensors of the collated batch
-- same values, shapes and dtypes as the reference's -- directly on the device.  Per batch the host only computes
three prefix sums over the batch's crystals and uploads them with the ids (a few KB).

The element parsing rules of `__getitem__` (string formulas via batch_comp, per-atom tuples, ndarray inputs) are
applied at pack time; they are host logic and mirror data.py:62-80.
"""
import json
import re

import numpy as np
import torch

from . import _lib
from .graph import GraphBatch

C = _lib.C


def _element_list(data, idx):
    elements = data["comps"][idx]
    if isinstance(elements, str):
        pattern = re.compile(r"([a-z]+)(\d+)", re.IGNORECASE)
        try:
            matches = pattern.findall(data["batch_comp"][idx])
        except TypeError:
            matches = pattern.findall(data["batch_comp"][idx][0])
        elements = []
        for el, count in matches:
            elements += [el] * int(count)
    if hasattr(elements, "tolist"):
        elements = elements.tolist()
    if isinstance(elements[0], (list, tuple)):
        elements = [el[0] for el in elements]
    return list(elements)


class PackedDataset:
    """Packed, device-resident form of the reference's dataset dictionary
    (`{'input', 'comps', 'batch_comp', 'target'}`, either `input` layout of data.py:47-50)."""

    def __init__(self, arrays, host, device):
        self.t, self.host, self.device = arrays, host, device
        s = _lib.PackedDatasetStruct()
        s.n_graphs, s.fea, s.max_nbr, s.n_elem = host["n_graphs"], host["fea"], host["max_nbr"], host["n_elem"]
        for k in ("table", "atom_ptr", "atom_elem", "shell", "self_idx", "nbr_idx", "comp_ptr", "comp_elem",
                  "comp_weight", "y_val"):
            setattr(s, k, arrays[k].data_ptr())
        self.c = s

    @classmethod
    def from_dict(cls, data, embedding, max_neighbor_number=12, target="e_above_hull", device="cuda:0"):
        """`embedding`: path of the element-embedding JSON (reference `fea_path`) or a dict symbol -> vector."""
        if isinstance(embedding, str):
            with open(embedding) as f:
                embedding = json.load(f)
        symbols = list(embedding.keys())
        elem_id = {el: k for k, el in enumerate(symbols)}
        table = np.asarray([embedding[el] for el in symbols], dtype=np.float64).astype(np.float32)
        fmt = 1 if data["input"].shape[0] > 3 else 0
        G = len(data["target"][target])
        K = int(max_neighbor_number)
        atom_ptr, comp_ptr = np.zeros(G + 1, np.int64), np.zeros(G + 1, np.int64)
        atom_elem, shell, self_idx, nbr_idx, comp_elem, comp_weight = [], [], [], [], [], []
        y_val = np.zeros(G, np.float32)
        for g in range(G):
            elements = _element_list(data, g)
            n = len(elements)
            for el in elements:
                if el not in elem_id:
                    raise AssertionError(f"{el} is not an allowed atom type")     # Featuriser.get_fea
            counts = {}
            for el in elements:
                counts[el] = counts.get(el, 0) + 1
            atom_elem.append(np.array([elem_id[el] for el in elements], np.int32))
            comp_elem.append(np.array([elem_id[el] for el in counts], np.int32))
            comp_weight.append(np.array([v / n for v in counts.values()], np.float32))
            tabs = (data["input"][0][g], data["input"][1][g], data["input"][2][g]) if fmt == 0 else \
                   (data["input"][g][0], data["input"][g][1], data["input"][g][2])
            sh, se, nb = (np.asarray(t)[:, 0:K].astype(np.int64) for t in tabs)
            if sh.shape[0] != n:
                raise ValueError(f"crystal {g}: {n} elements but {sh.shape[0]} rows of neighbours")
            if sh.shape[1] != K:
                raise ValueError(f"crystal {g}: {sh.shape[1]} stored neighbours < max_neighbor_number {K}")
            shell.append(sh.astype(np.int32)); self_idx.append(se.astype(np.int32)); nbr_idx.append(nb.astype(np.int32))
            t = np.float32(data["target"][target][g])
            y_val[g] = t * np.float32(n) if target != "volume" else t
            atom_ptr[g + 1] = atom_ptr[g] + n
            comp_ptr[g + 1] = comp_ptr[g] + len(counts)
        if atom_ptr[-1] * max(K, 1) >= 2 ** 31:
            raise ValueError("dataset too large for int32 offsets")
        cat = lambda xs, dt, shape: (np.concatenate(xs).astype(dt) if xs else np.zeros(shape, dt))
        arrays = {"table": table, "atom_ptr": atom_ptr.astype(np.int32), "atom_elem": cat(atom_elem, np.int32, (0,)),
                  "shell": cat(shell, np.int32, (0, K)), "self_idx": cat(self_idx, np.int32, (0, K)),
                  "nbr_idx": cat(nbr_idx, np.int32, (0, K)), "comp_ptr": comp_ptr.astype(np.int32),
                  "comp_elem": cat(comp_elem, np.int32, (0,)), "comp_weight": cat(comp_weight, np.float32, (0,)),
                  "y_val": y_val}
        host = {"n_graphs": G, "fea": table.shape[1], "max_nbr": K, "n_elem": table.shape[0],
                "natoms": np.diff(atom_ptr), "nuniq": np.diff(comp_ptr)}
        dev = {k: torch.from_numpy(np.ascontiguousarray(v)).to(device) for k, v in arrays.items()}
        return cls(dev, host, torch.device(device))

    def __len__(self):
        return self.host["n_graphs"]

    def collate(self, ids):
        """Returns (GraphBatch, roost_tuple) for the crystals `ids` (sequence of ints, in batch order): the
        tensors `Batch.from_data_list(...)` / `collate_batch(...)` produce in the reference, already on the device."""
        if self.device.type != "cuda":
            raise RuntimeError("PackedDataset.collate needs the HIP library and a GPU-resident dataset; there is no CPU path")
        ids = np.asarray(ids, dtype=np.int64).reshape(-1)
        if ids.size and (ids.min() < 0 or ids.max() >= len(self)):
            raise IndexError("crystal id out of range")
        B = ids.size
        na, nu = self.host["natoms"][ids], self.host["nuniq"][ids]
        meta = np.zeros((4, B + 1), np.int32)
        meta[0, :B] = ids
        meta[1, 1:] = np.cumsum(na); meta[2, 1:] = np.cumsum(nu); meta[3, 1:] = np.cumsum(nu * (nu - 1))
        N, Nc, Ec = int(meta[1, B]), int(meta[2, B]), int(meta[3, B])
        K, F, dev = self.host["max_nbr"], self.host["fea"], self.device
        E = N * K
        # pinned staging: a pageable source makes the upload a synchronous copy -- the one host sync per training step that
        # round 4's bench line still showed
        m = torch.from_numpy(meta).pin_memory().to(dev, non_blocking=True)
        f32, i64 = dict(dtype=torch.float32, device=dev), dict(dtype=torch.int64, device=dev)
        x, ei, ea = torch.empty(N, F, **f32), torch.empty(2, E, **i64), torch.empty(E, **i64)
        y, batch = torch.empty(B, **f32), torch.empty(N, **i64)
        cw, cf = torch.empty(Nc, 1, **f32), torch.empty(Nc, F, **f32)
        cs, cn, cc = torch.empty(Ec, **i64), torch.empty(Ec, **i64), torch.empty(Nc, **i64)
        out = _lib.CollatedStruct(*[t.data_ptr() for t in (x, ei, ea, y, batch, cw, cf, cs, cn, cc)])
        with torch.cuda.device(dev):
            _lib.check(_lib.lib.cgat_collate_batch(C.byref(self.c), m[0].data_ptr(), m[1].data_ptr(), m[2].data_ptr(),
                                                   m[3].data_ptr(), B, E, Ec, C.byref(out),
                                                   torch.cuda.current_stream().cuda_stream), "cgat_collate_batch")
        return GraphBatch(x, ei, ea, batch, y, B), (cw, cf, cs, cn, cc)
