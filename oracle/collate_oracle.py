"""CPU restatement (numpy) of the reference's batch collation -- the step right before the hot path.
TEST INFRASTRUCTURE ONLY: imported by tests/ and by bench.py's cpu_baseline leg, never by the product.

Follows, line by line in behaviour:
  * CGAT/data.py:61-144  CompositionData.__getitem__  (element list -> composition dict in first-appearance order,
    weights = count / n_atoms, fully connected composition graph without self loops, per-atom embedding rows,
    neighbour tables sliced to max_neighbor_number columns and flattened atom-major, y = target * n_atoms unless the
    target is 'volume');
  * CGAT/roost_message.py:400-458  collate_batch  (concatenate, offset the composition indices by the number of
    composition nodes before the crystal, crystal index per composition node);
  * PyG Batch.from_data_list (CGAT/lightning_module.py:200) -- third-party, un-vendored (torch_geometric 2.0.3):
    restated from its documented semantics (concatenate, offset edge_index by the nodes before the graph, `batch`
    = graph id per node).  UNPINNED UPSTREAM for that one call; everything else is pinned by tests/golden/collate.npz,
    recorded from the unmodified reference (tests/golden/make_collate_golden.py).
"""
import re

import numpy as np


def element_list(data, idx):
    """data.py:62-80: per-atom element symbols of crystal idx."""
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
    try:
        elements = elements.tolist()
    except AttributeError:
        pass
    if isinstance(elements[0], (list, tuple)):
        elements = [el[0] for el in elements]
    return list(elements)


def get_item(data, fmt, idx, table, elem_id, max_nbr, target):
    """One crystal: (x, edge_index, edge_attr, y), (weights, comp_fea, comp_self, comp_nbr)   (data.py:61-144)."""
    elements = element_list(data, idx)
    n = len(elements)
    comp = {}
    for el in elements:                         # dict keeps first-appearance order (data.py:84-90)
        comp[el] = elements.count(el)
    uniq = list(comp.keys())
    weights = np.array([v / n for v in comp.values()], dtype=np.float32)
    u = len(uniq)
    comp_self, comp_nbr = [], []
    for i in range(u):                          # data.py:91-96
        comp_self += [i] * (u - 1)
        comp_nbr += list(range(i)) + list(range(i + 1, u))
    comp_fea = table[[elem_id[el] for el in uniq]].astype(np.float32)
    x = table[[elem_id[el] for el in elements]].astype(np.float32)
    if fmt == 0:                                # data.py:105-122 / 123-138
        shell, self_i, nbr_i = data["input"][0][idx], data["input"][1][idx], data["input"][2][idx]
    else:
        shell, self_i, nbr_i = data["input"][idx][0], data["input"][idx][1], data["input"][idx][2]
    edge_attr = np.asarray(shell)[:, 0:max_nbr].flatten().astype(np.int64)
    ei = np.stack([np.asarray(self_i)[:, 0:max_nbr].flatten().astype(np.int64),
                   np.asarray(nbr_i)[:, 0:max_nbr].flatten().astype(np.int64)])
    t = np.float32(data["target"][target][idx])
    y = np.array([t * n if target != "volume" else t], dtype=np.float32)   # data.py:139-144
    return (x, ei, edge_attr, y), (weights, comp_fea, np.array(comp_self, dtype=np.int64), np.array(comp_nbr, dtype=np.int64))


def collate(data, ids, table, elem_id, max_nbr, target):
    """The tensors the hot path consumes for the crystals `ids` (in this order)."""
    fmt = 1 if data["input"].shape[0] > 3 else 0      # data.py:47-50
    xs, eis, eas, ys, bs = [], [], [], [], []
    ws, cf, cs, cn, ci = [], [], [], [], []
    base = cbase = 0
    for g, idx in enumerate(ids):
        (x, ei, ea, y), (w, f, s, nb) = get_item(data, fmt, idx, table, elem_id, max_nbr, target)
        xs.append(x); eis.append(ei + base); eas.append(ea); ys.append(y)
        bs.append(np.full(x.shape[0], g, dtype=np.int64))
        base += x.shape[0]
        ws.append(w); cf.append(f); cs.append(s + cbase); cn.append(nb + cbase)   # roost_message.py:437-452
        ci.append(np.full(f.shape[0], g, dtype=np.int64))
        cbase += f.shape[0]
    return {"x": np.concatenate(xs), "edge_index": np.concatenate(eis, axis=1), "edge_attr": np.concatenate(eas),
            "y": np.concatenate(ys), "batch": np.concatenate(bs),
            "comp0": np.concatenate(ws).reshape(-1, 1), "comp1": np.concatenate(cf), "comp2": np.concatenate(cs),
            "comp3": np.concatenate(cn), "comp4": np.concatenate(ci)}
