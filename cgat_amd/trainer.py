"""One data-parallel training step of the reference harness on this path (BASELINE configs[3]):

    LightningModel.training_step / evaluate / configure_optimizers        CGAT/lightning_module.py:185-259, 306-355
    Trainer(strategy='ddp', accumulate_grad_batches=...)                  CGAT/train.py:53-79

collate the rank's crystals on the device (PackedDataset) -> CGAtNet forward -> `.chunk(2, dim=1)` into
(output, log_std) -> robust loss against the normalised target -> backward with the bucketed gradient all-reduce
overlapped (GradientAverager: RCCL over xGMI) -> one fused AdamW launch.  Nothing here touches the host per crystal;
per step the host uploads the batch's crystal ids and three prefix sums.
"""
import numpy as np
import torch

from .dist import GradientAverager, shard_range
from .optim import FusedAdamW, RobustL1, RobustL2


class Normalizer:
    """target -> (target - mean) / std and back (CGAT/utils.py Normalizer as used at lightning_module.py:205-211)."""

    def __init__(self, mean=0.0, std=1.0):
        self.mean, self.std = float(mean), float(std)

    def norm(self, t):
        return (t - self.mean) / self.std

    def denorm(self, t):
        return t * self.std + self.mean


class DataParallelTrainer:
    def __init__(self, model, dataset, lr=1e-3, weight_decay=1e-2, loss="L1", normalizer=None, accumulate_grad_batches=1,
                 rank=0, world=1, bucket_bytes=64 << 20, force_averager=False, static_graph=True):
        self.model, self.dataset = model, dataset
        self.rank, self.world = rank, world
        self.params = [p for p in model.parameters() if p.requires_grad]
        self.optimizer = FusedAdamW(self.params, lr=lr, weight_decay=weight_decay)
        # static_graph: the set of never-used parameters is fixed by the architecture (Edge.MH_A / Edge.MH_M under
        # no_hyper=True), so the averager may freeze it after two steps and stop synchronising with the host
        # force_averager: the bucketed all-reduce also at world size 1 (a one-rank RCCL communicator; dist.GradientAverager)
        # (created only when it will be active: force=True without a process group raises in GradientAverager)
        self.averager = (GradientAverager(self.params, bucket_bytes=bucket_bytes, force=force_averager, static_graph=static_graph)
                         if (world > 1 or force_averager) else None)
        self.criterion = RobustL1 if loss == "L1" else RobustL2
        self.normalizer = normalizer or Normalizer()
        self.accumulate = int(accumulate_grad_batches)

    def local_ids(self, global_ids):
        """This rank's contiguous share of a global batch of crystal ids (graphs are independent: no halo)."""
        lo, hi = shard_range(len(global_ids), self.rank, self.world)
        return np.asarray(global_ids)[lo:hi]

    def _loss(self, ids):
        batch, roost = self.dataset.collate(ids)
        output, log_std = self.model(batch, roost).chunk(2, dim=1)
        target = self.normalizer.norm(batch.y.view(-1, 1))
        return self.criterion(output, log_std, target), batch

    def step(self, ids):
        """`ids`: this rank's crystal ids for the step (or a list of `accumulate_grad_batches` id arrays).
        Returns (loss tensor of the last micro-batch, edges processed on this rank)."""
        micro = list(ids) if self.accumulate > 1 else [ids]
        if self.averager is not None:
            self.averager.zero_grad()                  # gradients accumulate straight into the all-reduce buckets
        else:
            for p in self.params:
                p.grad = None
        edges = 0
        for k, mb in enumerate(micro):
            last = k == len(micro) - 1
            loss, batch = self._loss(mb)
            loss = loss / len(micro)
            if self.averager is not None and not last:
                with self.averager.no_sync():
                    loss.backward()
            else:
                loss.backward()
            edges += int(batch.edge_index.shape[1])
        if self.averager is not None:
            self.averager.finish()
        self.optimizer.step()
        return loss.detach(), edges
