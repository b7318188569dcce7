"""Data parallelism for the CGAT path: crystals are independent units (no edge crosses
crystals; reference lightning_module.py:200 / roost_message.py:445-452 offset indices per
graph), so ranks shard *graphs* -- no halo, no activation exchange.  The only collective is the
gradient mean after backward, what Lightning's strategy='ddp' does in the reference
(CGAT/train.py:56); here it is a bucketed all-reduce over torch.distributed (backend "nccl" =
RCCL over xGMI on MI355X, "gloo" in the CPU tests), launched from autograd hooks as soon as a
bucket's gradients exist so that it overlaps the rest of backward.
"""
import collections
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
    force = os.environ.get("CGAT_DIST_FORCE") == "1"                  # a one-rank communicator (RCCL on a 1-GPU box)
    if (world > 1 or force) and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29512")
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
    * Parameters that received no gradient on ANY rank in the last step (the reference's
      never-trained Edge.MH_A / Edge.MH_M, SURVEY §5: 2.7-5.5 MB in every 64-MB bucket of the
      shipped network) are COLD: they do not gate a bucket.  The globally all-reduced used-bitmap
      of a step -- identical on every rank -- decides the layout of the next one: hot parameters
      fill the leading buckets, which launch from the hooks while backward is still running;
      cold ones sit in trailing buckets that are reduced in `finish()` only if the bitmap shows
      that some rank used one of them after all (it is hot again from the next step on), and
      cost nothing otherwise.  The first step has no history: every parameter is hot and the
      buckets holding never-used ones launch in `finish()`.
    * Gradient accumulation: inside `with averager.no_sync():` backward only accumulates; the
      backward outside it reduces the sum of all micro-batches (as DDP.no_sync).
    * A parameter without gradient on ANY rank since the last `finish()` ends with
      `p.grad = None`, so optimisers skip it on every rank alike; one that was used on some rank
      gets the mean on all of them (zeros contributed where it was unused) -- DDP's
      find_unused_parameters, with the same used-bitmap all-reduce.
    * `static_graph=True` (as DDP's): once two consecutive steps have produced the same global bitmap the cold set is
      FROZEN -- `finish()` then issues no bitmap all-reduce and, above all, no host synchronisation (reading the bitmap
      drains the launch queue: +2.3 ms on the 24.4 ms layer step, measured over a one-rank RCCL communicator); the
      all-reduces are only stream-ordered.  A cold parameter that receives a gradient on some rank after that raises on
      EVERY rank in the same finish() call, VIOL_LAG = 2 steps after the step that did it (an 8-byte flag is all-reduced
      asynchronously each step and its pinned copy is read two steps later: no rank is left waiting in a collective).
      Until then the replicas have already taken two optimiser steps on the un-reduced -- rank-divergent -- gradient of
      that parameter: the error means "these replicas have diverged, restart with static_graph=False", it does not
      prevent the divergence.  Reading the two-step-old flag blocks the host on that step's event, which also bounds the
      host's run-ahead to two steps.  A hot parameter that receives no gradient contributes zeros and keeps its
      (zero-mean) gradient instead of None.
    * `force=True` keeps all of this alive at world size 1 (a one-rank communicator): the mean is
      then the identity, and the hooks, the asynchronous collectives and their interplay with the
      layer's side stream can be exercised over RCCL on a single GPU (tests/test_rccl_one_rank.py).

    * `bucket_bytes` (default 8 MB since round 6; 64 MB before): the headline layer's gradients are 37.9 MB -- one 64-MB
      bucket launched behind the LAST gradient, i.e. no overlap with backward at all; at 8 MB the four 8.45-MB hypernetwork
      head weights are buckets of their own and the rest packs into a fifth and sixth, each launched from the hook of its
      last gradient while backward is still running.  A ring all-reduce over 7 xGMI links moves 8 MB in ~50 us: still
      bandwidth-, not latency-bound.

    `stats` after each `finish()`: buckets launched from the hooks (i.e. overlapped with backward)
    and in finish(), cold buckets reduced / skipped, bytes reduced."""

    def __init__(self, params, bucket_bytes=8 << 20, group=None, force=False, static_graph=False):
        self.params = [p for p in params if p.requires_grad]
        self.group = group
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        if force and not dist.is_initialized():
            # an inactive averager would still install zero bucket views in zero_grad() and never turn the unused
            # parameters' gradients back into None: optimisers would then decay parameters the plain path skips
            raise RuntimeError("GradientAverager(force=True) needs an initialised process group (a one-rank group will do)")
        self.active = (self.world > 1 or bool(force)) and dist.is_initialized()
        self.bucket_bytes = int(bucket_bytes)
        self.static_graph, self._frozen = bool(static_graph), False
        self._index = {id(p): i for i, p in enumerate(self.params)}
        self._cold = frozenset()
        self._last_cold = None                         # the previous step's global cold set (None: no step observed yet)
        # frozen mode: one entry (step, flag tensor, pinned copy, event) per step, oldest first; an entry is read -- blocking on
        # its event -- exactly VIOL_LAG steps after it was made, so every rank reads step k's (global) flag in the same finish()
        self._viol_q = collections.deque()
        self._home = None
        self._viol_step = 0
        self._viol_sticky = None                       # device tensor: the last global flag (a violation stays visible)
        self.stats = {"launched_in_backward": 0, "launched_in_finish": 0, "cold_reduced": 0, "cold_skipped": 0,
                      "bytes_reduced": 0, "rebuilds": 0}
        self._layout(preserve=False)
        self._sync = True
        self._handles = []
        self._reset()
        self._by_ptr = {p.data_ptr(): i for i, p in enumerate(self.params)}
        if self.active:
            for p in self.params:
                self._handles.append(p.register_post_accumulate_grad_hook(self._on_grad))
            try:                                       # (the CPU tests run this file without the HIP package)
                from . import ops
                self._ops = ops
                ops.register_grad_sink(self._sink)
            except Exception:
                self._ops = None

    # -- bucket layout --------------------------------------------------------------------
    def _pack(self, idxs):
        """Consecutive parameters of one dtype/device into buckets of <= bucket_bytes."""
        buckets, cur, cur_bytes = [], [], 0
        for i in idxs:
            p = self.params[i]
            nb = p.numel() * p.element_size()
            if cur and (cur_bytes + nb > self.bucket_bytes or p.dtype != self.params[cur[0]].dtype or
                        p.device != self.params[cur[0]].device):
                buckets.append(cur)
                cur, cur_bytes = [], 0
            cur.append(i)
            cur_bytes += nb
        if cur:
            buckets.append(cur)
        return buckets

    def _layout(self, preserve):
        """(Re)build the buckets: hot parameters in reverse registration order first, then the cold ones.  With
        `preserve` the gradients held by the old views move to the new ones."""
        order = list(range(len(self.params) - 1, -1, -1))
        hot = self._pack([i for i in order if i not in self._cold])
        cold = self._pack([i for i in order if i in self._cold])
        idx_buckets = hot + cold
        self.n_hot = len(hot)
        old_view = getattr(self, "_view", None)
        self.buckets = [[self.params[i] for i in b] for b in idx_buckets]
        self.flat = [torch.zeros(sum(p.numel() for p in b), dtype=b[0].dtype, device=b[0].device) for b in self.buckets]
        self._view, self._bucket_of = {}, {}
        for bi, b in enumerate(self.buckets):
            off = 0
            for p in b:
                self._view[id(p)] = self.flat[bi][off:off + p.numel()].view_as(p)
                self._bucket_of[id(p)] = bi
                off += p.numel()
        if preserve and old_view is not None:
            for p in self.params:
                if p.grad is not None and p.grad.data_ptr() == old_view[id(p)].data_ptr():
                    self._view[id(p)].copy_(p.grad)
                    p.grad = self._view[id(p)]

    def _reset(self):
        self._pending = [len(b) for b in self.buckets]
        self._works = [None] * len(self.buckets)
        self._next = 0                                 # buckets [0, _next) have been launched
        self._used = [0] * len(self.params)
        self._claimed = [False] * len(self.params)     # handed out as a gradient destination in this step (_sink)
        self._stale = [True] * len(self.params)        # the view holds last step's mean, not this step's sum
        self._in_finish = False

    # -- public ---------------------------------------------------------------------------
    def zero_grad(self):
        """Drop every gradient.  The next backward leaves each gradient IN its bucket: this package's backward passes ask
        for the destination of a parameter's gradient (cgat_amd.ops.register_grad_sink) and write it straight into the
        bucket view, which autograd then installs as `p.grad` without a copy; a gradient that arrives in a tensor of its
        own (any other module) is copied into its view by the hook -- one pass.  (Until round 6 this zero-filled the
        buckets and installed the views up front, so that autograd ADDED every gradient into them: a fill, a read of the
        gradient and a read-modify-write of the bucket per parameter, 0.44 ms per 38-MB layer step at world size 1.)"""
        if self.flat and self.flat[0].is_cuda:
            self._home = torch.cuda.current_stream(self.flat[0].device)    # (hooks may fire under another stream)
        for i, p in enumerate(self.params):
            p.grad = None
            self._stale[i] = True

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
        if not self.active:
            return
        self._in_finish = True
        for i, p in enumerate(self.params):            # never reached by autograd in this step
            if not self._used[i]:
                self._adopt(i, p, arrived=False)
        while self._next < self.n_hot:                 # hot buckets some parameter of which never got a gradient
            self._launch(self._next)
        if self._frozen:
            return self._finish_frozen()
        dev = self.flat[0].device if self.flat else torch.device("cpu")
        used = torch.tensor(self._used, dtype=torch.int32, device=dev)
        dist.all_reduce(used, op=dist.ReduceOp.SUM, group=self.group)
        used = used.tolist()
        # cold buckets: reduced only if the (global, hence rank-independent) bitmap shows one of their parameters in use
        for bi in range(self.n_hot, len(self.buckets)):
            if any(used[self._index[id(p)]] for p in self.buckets[bi]):
                self._launch(bi)
                self.stats["cold_reduced"] += 1
            else:
                self.stats["cold_skipped"] += 1
        inv = 1.0 / self.world
        for bi in range(len(self.buckets)):
            if self._works[bi] is not None:
                self._works[bi].wait()
                if inv != 1.0:
                    self.flat[bi].mul_(inv)
        for i, p in enumerate(self.params):
            p.grad = self._view[id(p)] if used[i] else None
        cold = frozenset(i for i in range(len(self.params)) if not used[i])
        same_as_last = self._last_cold is not None and cold == self._last_cold
        self._last_cold = cold
        if cold != self._cold:                         # same decision on every rank: the bitmap is global
            self._cold = cold
            self._layout(preserve=True)
            self.stats["rebuilds"] += 1
        elif self.static_graph and same_as_last:
            # TWO consecutive steps have produced this global bitmap (the first step never matches: the initial empty
            # cold set is a default, not an observation): no more bitmap / host sync
            self._frozen = True
        self._reset()

    VIOL_LAG = 2   # steps between a flag's all-reduce and the finish() that reads it (its event is long complete by then)

    def _finish_frozen(self):
        """static_graph after the cold set froze: stream-ordered waits only, no collective on the bitmap, no host sync.

        A cold parameter that receives a gradient on SOME rank violates the contract.  Raising on that rank alone would
        leave the others waiting in their next collective, so the violation travels: every step all-reduces one flag
        (8 bytes, asynchronous, stream-ordered; the flag also carries the previous step's global result, so a violation
        stays visible) and copies the sum to pinned memory behind an event.  The entries form a FIFO -- none is ever
        overwritten -- and the entry of step k is read in the finish() of step k + VIOL_LAG ON EVERY RANK (blocking the
        HOST on an event that is two steps old: no GPU idle time, but the host cannot run more than two steps ahead of
        the device): the value is global, the step is fixed, so all ranks raise in the same call and none is left
        waiting in a collective.  (Round 4 looked at a single slot with event.query(): a
        not-yet-ready flag was overwritten by the next step's, and readiness differed between ranks.)"""
        try:
            self._check_violation()
        except RuntimeError:
            # leave the averager usable (and this step's collectives consumed) before the error travels up: every rank
            # raises in this same call, so every rank waits the same hot-bucket works
            for bi in range(self.n_hot):
                if self._works[bi] is not None:
                    self._works[bi].wait()
            self._reset()
            raise
        local = any(self._used[i] for i in self._cold)
        dev = self.flat[0].device if self.flat else torch.device("cpu")
        flag = torch.full((1,), 1 if local else 0, dtype=torch.int64, device=dev)
        if self._viol_sticky is not None:
            flag = torch.maximum(flag, (self._viol_sticky > 0).to(torch.int64))
        work = dist.all_reduce(flag, op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        work.wait()                                    # stream-ordered on GPU backends
        self._viol_sticky = flag
        if dev.type == "cuda":
            host = torch.empty(1, dtype=torch.int64, pin_memory=True)
            host.copy_(flag, non_blocking=True)
            ev = torch.cuda.Event()
            ev.record()
            self._viol_q.append((self._viol_step, flag, host, ev))
        else:
            self._viol_q.append((self._viol_step, flag, flag, None))
        self._viol_step += 1
        inv = 1.0 / self.world
        for bi in range(self.n_hot):
            self._works[bi].wait()
            if inv != 1.0:
                self.flat[bi].mul_(inv)
        for bi in range(self.n_hot, len(self.buckets)):
            self.stats["cold_skipped"] += 1
        for i, p in enumerate(self.params):
            p.grad = None if i in self._cold else self._view[id(p)]
        self._reset()

    def _check_violation(self, drain=False):
        """Frozen mode: raise (on every rank in the same call) if some rank saw a gradient on a cold parameter VIOL_LAG
        steps ago (drain: in any step still queued)."""
        bad = False
        while self._viol_q and (drain or self._viol_q[0][0] <= self._viol_step - self.VIOL_LAG):
            _, _, host, ev = self._viol_q.popleft()
            if ev is not None:
                ev.synchronize()
            bad = bad or int(host[0]) > 0
        if bad:
            self._viol_q.clear()
            raise RuntimeError("GradientAverager(static_graph=True): a parameter that was unused on every rank when the "
                               "graph froze received a gradient on some rank; construct the averager with "
                               "static_graph=False")

    def close(self):
        self._check_violation(drain=True)
        for h in self._handles:
            h.remove()
        self._handles = []
        if getattr(self, "_ops", None) is not None:
            self._ops.unregister_grad_sink(self._sink)

    def _sink(self, w):
        """Destination of the gradient of parameter `w` (cgat_amd.ops._param_grad): its bucket view -- a fresh alias, so
        that autograd can take it over as p.grad without a copy -- while nothing has been accumulated into it in this
        step (a second use of the parameter, or a later micro-batch under no_sync(), gets a tensor of its own and is
        ADDED by autograd, as before)."""
        i = self._by_ptr.get(w.data_ptr())
        if i is None or self._used[i] or self._claimed[i]:
            return None
        p = self.params[i]
        if p.grad is not None or w.numel() != p.numel() or w.dtype != p.dtype:
            return None
        # one claim per step: two backward nodes of one parameter may both run before its hook does (autograd sums their
        # results in its input buffer first) -- the second must not overwrite the first's
        self._claimed[i] = True
        return self._view[id(p)].view(w.shape)

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
        if self.flat[bi].is_cuda:
            # gradients of one bucket may have been written on more than one stream (cgat_amd.ops.branch_stream: the
            # composition branch runs beside the graph layers at small batches; the f16x3 mode's side stream): the
            # collective is ordered behind the CURRENT stream only, so the current stream waits for the others first
            from .ops import aux_streams
            cur = torch.cuda.current_stream(self.flat[bi].device)
            others = aux_streams() + ([self._home] if self._home is not None else [])   # _home: the caller's stream
            for s in others:
                if s != cur and s.device == self.flat[bi].device:
                    cur.wait_stream(s)
        self._works[bi] = dist.all_reduce(self.flat[bi], op=dist.ReduceOp.SUM, group=self.group, async_op=True)
        if bi < self.n_hot:
            self._next = bi + 1
        self.stats["launched_in_finish" if self._in_finish else "launched_in_backward"] += 1
        self.stats["bytes_reduced"] += self.flat[bi].numel() * self.flat[bi].element_size()

    def _on_grad(self, p):
        i = self._index[id(p)]
        self._adopt(i, p, arrived=True)
        self._used[i] = 1
        if not self._sync:
            return
        bi = self._bucket_of[id(p)]
        self._pending[bi] -= 1
        # strictly in bucket order: the same sequence of collectives on every rank (cold buckets wait for finish())
        while self._next < self.n_hot and self._pending[self._next] <= 0:
            self._launch(self._next)
