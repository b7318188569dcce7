"""Dev tool: where one training step synchronises the host -- the step under torch's sync-debug mode with the warning
turned into an error, so that the traceback names the line."""
import os, sys, traceback, warnings
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import cgat_amd as P
from cgat_amd import ops
from cgat_amd.graph import synthetic_dataset_dict
from cgat_amd.trainer import DataParallelTrainer

dev = torch.device("cuda:0")
data, emb = synthetic_dataset_dict(128, (2, 40), 24, seed=100)
ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device=dev)
torch.manual_seed(1)
net = P.CGAtNet(200, 128, 4, msg_heads=3, neighbor_number=12, update_edges=True).to(dev)
tr = DataParallelTrainer(net, ds, lr=1e-4, weight_decay=1e-6, rank=0, world=1)
ops.set_validate_indices(False)
ids = np.random.RandomState(0).permutation(128)[:64]
for _ in range(2):
    tr.step(ids)
torch.cuda.synchronize()
torch.cuda.set_sync_debug_mode("warn")
with warnings.catch_warnings():
    warnings.simplefilter("error")
    try:
        tr.step(ids)
        print("no host synchronisation in the step")
    except Exception:
        traceback.print_exc()
torch.cuda.set_sync_debug_mode("default")
