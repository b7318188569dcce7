"""Generates tests/golden/collate.npz from the UNMODIFIED reference at /root/reference: for synthetic
datasets in the reference's dictionary format (collate_recipe.py) it runs the reference's own
CGAT.data.CompositionData.__getitem__ (data.py:61-144) and CGAT.roost_message.collate_batch (400-458) for a few
batches and stores every tensor of the collated batch.  PyG's Batch.from_data_list (lightning_module.py:200) is an
un-vendored third-party dependency (torch_geometric 2.0.3): it is stood in by its documented semantics here
(concatenate x / edge_attr / y along dim 0, edge_index along dim 1 with each graph's indices offset by the number
of nodes before it, batch = graph id per node) -- unpinned upstream, as stated in the oracle.

    python tests/golden/make_collate_golden.py          (authoring container only; needs /root/reference)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import collate_recipe as R

REF = "/root/reference"


class Data:  # stand-in for torch_geometric.data.Data: a plain attribute bag, as data.py uses it
    def __init__(self, **kw):
        self.__dict__.update(kw)


def batch_from_data_list(items):
    xs, eis, eas, ys, bs = [], [], [], [], []
    base = 0
    for g, d in enumerate(items):
        xs.append(d.x); eas.append(d.edge_attr); ys.append(d.y)
        eis.append(d.edge_index + base)
        bs.append(torch.full((d.x.shape[0],), g, dtype=torch.long))
        base += d.x.shape[0]
    return Data(x=torch.cat(xs), edge_index=torch.cat(eis, dim=1), edge_attr=torch.cat(eas), y=torch.cat(ys),
                batch=torch.cat(bs))


def main():
    tgd = types.ModuleType("torch_geometric.data")
    tgd.Data = Data
    tg = types.ModuleType("torch_geometric")
    tg.data = tgd
    sys.modules["torch_geometric"], sys.modules["torch_geometric.data"] = tg, tgd
    # CGAT/__init__.py imports CGAT.CGAT, which needs the layer shims of make_golden.py
    import make_golden
    make_golden.install_shims()
    sys.modules["torch_geometric.data"].Data = Data
    sys.path.insert(0, REF)
    from CGAT.data import CompositionData
    from CGAT.roost_message import collate_batch

    out = {}
    with tempfile.TemporaryDirectory() as tmp:
        fea_path = R.write_embedding_json(os.path.join(tmp, "emb.json"))
        for name, n_graphs, seed, fmt, comps_as, max_nbr, target, batches in R.CASES:
            data = R.make_dataset(n_graphs, seed, fmt, comps_as)
            ds = CompositionData(data, fea_path, max_neighbor_number=max_nbr, target=target)
            assert ds.format == fmt
            for bi, ids in enumerate(batches):
                items = [ds[i] for i in ids]
                batch = batch_from_data_list([it[0] for it in items])
                comp = collate_batch([it[1] for it in items])
                k = f"{name}.b{bi}."
                out[k + "x"] = batch.x.numpy(); out[k + "edge_index"] = batch.edge_index.numpy()
                out[k + "edge_attr"] = batch.edge_attr.numpy(); out[k + "y"] = batch.y.numpy()
                out[k + "batch"] = batch.batch.numpy()
                for j, t in enumerate(comp):
                    out[k + f"comp{j}"] = t.numpy()
    np.savez_compressed(os.path.join(HERE, "collate.npz"), **out)
    print("wrote", os.path.join(HERE, "collate.npz"), len(out), "arrays,",
          os.path.getsize(os.path.join(HERE, "collate.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
