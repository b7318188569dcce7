"""Data parallelism for the CGAT path: crystals are independent units (no edge crosses
crystals; reference lightning_module.py:200 / roost_message.py:445-452 offset indices per
graph), so ranks shard *graphs* -- no halo, no activation exchange.  The only collective is the
gradient mean after backward, what Lightning's strategy='ddp' does in the reference
(CGAT/train.py:56); here it is a bucketed all-reduce over torch.distributed (backend "nccl" =
RCCL over xGMI on MI355X, "gloo" in the CPU tests), launched from autograd hooks as soon as a
bucket's gradients exist so that it overlaps the rest of backward.
"""
import contextlib
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
    """Bucketed gradient mean across ranks, overlapped with backward (the DDP of the reference's
    Lightning harness, train.py:56-62 incl. accumulate_grad_batches).

    * Gradients live IN the buckets: `p.grad` is a view of its bucket's flat buffer
      (`zero_grad()` installs the views), autograd accumulates into it in place, and the
      all-reduce runs on the flat buffer -- no pack / unpack passes.  If something replaced
      `p.grad` by a tensor of its own (`optimizer.zero_grad(set_to_none=True)` makes autograd
      allocate a fresh one), the hook folds that tensor into the view and re-installs it.
    * Buckets are filled in reverse registration order (~ the order backward produces them) and
      launched strictly in bucket order -- bucket i waits for bucket i-1 -- so every rank issues
      the same sequence of collectives whatever order its autograd engine ran in.
    * Gradient accumulation: inside `with averager.no_sync():` backward only accumulates; the
      backward outside it reduces the sum of all micro-batches (as DDP.no_sync).
    * Parameters that received no gradient on ANY rank since the last `finish()` (the reference's
      never-trained Edge.MH_A / Edge.MH_M) end with `p.grad = None`, so optimisers
      skip them on every rank alike; one that was used on some rank gets the mean on all of them
      (zeros contributed where it was unused) -- DDP's find_unused_parameters, with the same
      used-bitmap all-reduce."""

    def __init__(self, params, bucket_bytes=64 << 20, group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.buckets = []
        cur, cur_bytes = [], 0
        for p in reversed(self.params):
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > bucket_bytes or p.dtype != cur[0].dtype or p.device != cur[0].device):
                self.buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(p)
            cur_bytes += nb
        if cur:
            self.buckets.append(cur)
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device)
                     for b in self.buckets]
        self._view, self._bucket_of, self._index = {}, {}, {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self._view[id(p)] = self.flat[bi][off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = bi
                off += p.numel()
        for i, p in enumerate(self.params):
            self._index[id(p)] = i
        self._sync = True
        self._handles = []
        self._reset()
        self._stale = [True] * len(self.params)        # the view holds last step's mean, not this step's sum
        if self.world > 1:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._next = 0                                 # buckets [0, _next) have been launched
        self._used = [0] * len(self.params)
        self._stale = [True] * len(self.params)

    # -- public ---------------------------------------------------------------------------
    def zero_grad(self):
        """Zero every gradient and point `p.grad` at its bucket view (the fast path: autograd
        then accumulates straight into the bucket)."""
        for f in self.flat:
            f.zero_()
        for i, p in enumerate(self.params):
            p.grad = self._view[id(p)]
            self._stale[i] = False

    @contextlib.contextmanager
    def no_sync(self):
        """Accumulate gradients without communication (micro-batches before the last one)."""
        old, self._sync = self._sync, False
        try:
            yield
        finally:
            self._sync = old

    def finish(self):
        """Call after the (last) backward().  Leaves the mean gradients in p.grad."""
        if self.world == 1:
            return
        for i, p in enumerate(self.params):            # never reached by autograd in this step
            if not self._used[i]:
                self._adopt(i, p, arrived=False)
        while self._next < len(self.buckets):          # buckets some parameter of which never got a gradient
            self._launch(self._next)
        dev = self.flat[0].device if self.flat else torch.device("cpu")
        used = torch.tensor(self._used, dtype=torch.int32, device=dev)
        dist.all_reduce(used, op=dist.ReduceOp.SUM, group=self.group)
        used = used.tolist()
        for bi in range(len(self.buckets)):
            self._works[bi].wait()
            self.flat[bi].mul_(1.0 / self.world)
        for i, p in enumerate(self.params):
            p.grad = self._view[id(p)] if used[i] else None
        self._reset()

    def close(self):
        for h in self._handles:
            h.remove()
        self._handles = []

    # -- internals ------------------------------------------------------------------------
    def _adopt(self, i, p, arrived):
        """Make p.grad the bucket view holding this step's local gradient sum."""
        view = self._view[id(p)]
        g = p.grad
        if g is None or not arrived:
            if g is None or g.data_ptr() != view.data_ptr():
                if g is None:
                    view.zero_()                       # no local gradient: contributes zeros
                else:
                    view.copy_(g)                      # a gradient left from an earlier no_sync() backward
            elif self._stale[i]:
                view.zero_()                           # view still holds last step's mean and nothing was added
        elif g.data_ptr() != view.data_ptr():
            if self._stale[i]:
                view.copy_(g)
            else:
                view.add_(g)
        p.grad = view
        self._stale[i] = False

    def _launch(self, bi):
        self._works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        self._next = bi + 1

    def _on_grad(self, p):
        i = self._index[id(p)]
        self._adopt(i, p, arrived=True)
        self._used[i] = 1
        if not self._sync:
            return
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        # strictly in bucket order: the same sequence of collectives on every rank
        while self._next < len(self.buckets) and self._pending[self._next] <= 0:
            self._launch(self._next)
