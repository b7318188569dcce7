"""Data parallelism for the CGAT path: crystals are independent units (no edge crosses
crystals; reference lightning_module.py:200 / roost_message.py:445-452 offset indices per
graph), so ranks shard *graphs* -- no halo, no activation exchange.  The only collective is the
gradient mean after backward, what Lightning's strategy='ddp' does in the reference
(CGAT/train.py:56); here it is a bucketed all-reduce over torch.distributed (backend "nccl" =
RCCL over xGMI on MI355X, "gloo" in the CPU tests), launched from autograd hooks as soon as a
bucket's gradients exist so that it overlaps the rest of backward.
"""
import os

import torch
import torch.distributed as dist


def init_from_env(backend=None):
    """One process per GPU, launched by torch.distributed.run.  Returns (rank, world, device)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    use_gpu = torch.cuda.is_available()
    backend = backend or os.environ.get("CGAT_DIST_BACKEND")          # test hook: "gloo" on a 1-GPU box
    if use_gpu and os.environ.get("CGAT_DIST_SHARE_GPU") == "1":      # test hook: all ranks on cuda:0
        local = 0
    device = torch.device(f"cuda:{local}") if use_gpu else torch.device("cpu")
    if use_gpu:
        torch.cuda.set_device(device)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend or ("nccl" if use_gpu else "gloo"), rank=rank, world_size=world)
    return rank, world, device


def shard_range(n_items, rank, world):
    """Contiguous balanced split [lo, hi) of n_items graphs."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


class GradientAverager:
    """Bucketed gradient mean across ranks, overlapped with backward.

    Parameters are packed (in reverse registration order ~ the order backward produces them) into
    flat buckets of <= bucket_bytes.  A post-accumulate-grad hook counts arrivals; when a bucket is
    complete its gradients are copied into the flat buffer and an async all-reduce starts.
    `finish()` waits, scales by 1/world and copies back.  Parameters that received no gradient in
    this step (the reference's never-trained Edge.MH_A / Edge.MH_M) contribute zeros,
    which is what DDP's find_unused_parameters does."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and cur_bytes + nb > bucket_bytes:
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device)
                     for b in self.buckets]
        self._bucket_of = {}
        for bi, b in enumerate(self.buckets):
            for p in b:
                self._bucket_of[id(p)] = bi
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)
        self._works = [None] * len(self.buckets)
        self._handles = []
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _pack_and_launch(self, bi):
        flat, off = self.flat[bi], 0
        for p in self.buckets[bi]:
            n = p.numel()
            if p.grad is None:
                flat[off:off + n].zero_()
            else:
                flat[off:off + n].copy_(p.grad.reshape(-1))
            off += n
        self._works[bi] = dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._launched[bi] = True

    def _on_grad(self, p):
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        if self._pending[bi] == 0 and not self._launched[bi]:
            self._pack_and_launch(bi)

    def finish(self):
        """Call after backward().  Leaves the averaged gradients in p.grad."""
        if self.world == 1:
            return
        for bi in range(len(self.buckets)):
            if not self._launched[bi]:      # some parameter of the bucket never got a gradient
                self._pack_and_launch(bi)
        for bi, b in enumerate(self.buckets):
            self._works[bi].wait()
            flat, off = self.flat[bi], 0
            flat.mul_(1.0 / self.world)
            for p in b:
                n = p.numel()
                if p.grad is not None:
                    p.grad.copy_(flat[off:off + n].view_as(p.grad))
                off += n
        self._pending = [len(b) for b in self.buckets]
        self._launched = [False] * len(self.buckets)

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []
