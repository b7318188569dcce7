"""Fused optimiser steps, robust losses and the cyclical learning-rate schedule -- the per-step work
after the hot path, behind the reference's own names.

  FusedAdamW  <->  torch.optim.AdamW(parameters, lr, weight_decay)        CGAT/lightning_module.py:328-331
  FusedLamb   <->  CGAT.lambs.JITLamb(parameters, lr, weight_decay)       CGAT/lightning_module.py:332-335, lambs.py:155-262
  RobustL1 / RobustL2                                                     CGAT/utils.py:30-47
  cyclical_lr                                                             CGAT/utils.py:50-64

Each optimiser step is ONE kernel launch over all parameter tensors (three for LAMB, which needs per-tensor norms)
through the C ABI (`cgat_adamw_step`, `cgat_lamb_step`): a device table of (param, grad, exp_avg, exp_avg_sq, n) plus a
list of fixed-size chunks.  No CPU fallback."""
import math

import numpy as np
import torch

from . import _lib

C = _lib.C


class _MultiTensor(torch.optim.Optimizer):
    def __init__(self, params, defaults):
        super().__init__(params, defaults)
        self._plan = {}
        self._live_grads = []

    def _tensors(self, group):
        ps = [p for p in group["params"] if p.grad is not None]
        for p in ps:
            if not p.is_cuda or p.dtype != torch.float32 or p.grad.is_sparse:
                raise RuntimeError(f"{type(self).__name__} handles dense fp32 parameters on the GPU (no CPU fallback)")
            if not p.is_contiguous():
                raise RuntimeError("parameters must be contiguous")
            st = self.state[p]
            if len(st) == 0:
                st["step"] = 0
                st["exp_avg"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
                st["exp_avg_sq"] = torch.zeros_like(p, memory_format=torch.contiguous_format)
            # a state loaded from torch.optim.AdamW carries `step` as a tensor: same name, normalised to an int here
            st["step"] = int(st["step"]) + 1
        return ps

    def _launch_plan(self, ps):
        """Chunk list (cached per set of parameter sizes) and the per-step pointer table."""
        key = tuple(p.numel() for p in ps)
        dev = ps[0].device
        if key not in self._plan:
            ch = _lib.lib.cgat_mt_chunk_elems()
            ct, co, first = [], [], [0]
            for i, n in enumerate(key):
                offs = np.arange(0, max(n, 1), ch, dtype=np.int64) if n > 0 else np.zeros(0, np.int64)
                ct.append(np.full(offs.size, i, np.int32)); co.append(offs)
                first.append(first[-1] + offs.size)
            self._plan[key] = (torch.from_numpy(np.concatenate(ct)).to(dev), torch.from_numpy(np.concatenate(co)).to(dev),
                               torch.from_numpy(np.asarray(first, np.int32)).to(dev), first[-1])
        tab = np.empty((len(ps), 5), np.int64)
        # re-laid (non-contiguous) gradients stay alive until the NEXT step's launch has been queued; not in
        # self.state: state_dict() must hold exactly torch.optim.AdamW's entries (step, exp_avg, exp_avg_sq)
        live = []
        for i, p in enumerate(ps):
            st = self.state[p]
            g = p.grad
            if not g.is_contiguous():
                g = g.contiguous()
                live.append(g)
            tab[i] = (p.data_ptr(), g.data_ptr(), st["exp_avg"].data_ptr(), st["exp_avg_sq"].data_ptr(), p.numel())
        self._live_grads = live
        # pinned staging: from pageable memory the upload is a synchronous copy -- the one host synchronisation a training
        # step still had (tools/sync_probe.py); the caching host allocator keeps the pinned block until the copy has run
        return self._plan[key], torch.from_numpy(tab).pin_memory().to(dev, non_blocking=True)


class FusedAdamW(_MultiTensor):
    """torch.optim.AdamW semantics (lr, betas, eps, weight_decay; no amsgrad / maximize), one launch per step."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = self._tensors(group)
            if not ps:
                continue
            steps = {self.state[p]["step"] for p in ps}
            if len(steps) != 1:
                raise RuntimeError("FusedAdamW: parameters of a group must share their step count")
            (ct, co, _, n_chunks), tab = self._launch_plan(ps)
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                _lib.check(_lib.lib.cgat_adamw_step(tab.data_ptr(), ct.data_ptr(), co.data_ptr(), n_chunks, group["lr"],
                                                    b1, b2, group["eps"], group["weight_decay"], steps.pop(),
                                                    torch.cuda.current_stream().cuda_stream), "cgat_adamw_step")
        return loss


class FusedLamb(_MultiTensor):
    """The reference's JITLamb (CGAT/lambs.py): LAMB without bias correction, weight norm clamped to [0, 10]."""

    def __init__(self, params, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0, adam=False):
        if adam:
            raise NotImplementedError("adam=True (trust ratio forced to 1) is not used by the reference's harness")
        super().__init__(params, dict(lr=lr, betas=betas, eps=eps, weight_decay=weight_decay))

    @torch.no_grad()
    def step(self, closure=None):
        loss = closure() if closure is not None else None
        for group in self.param_groups:
            ps = self._tensors(group)
            if not ps:
                continue
            (ct, co, first, n_chunks), tab = self._launch_plan(ps)
            ws = torch.empty(2 * n_chunks + len(ps), dtype=torch.float32, device=ps[0].device)
            b1, b2 = group["betas"]
            with torch.cuda.device(ps[0].device):
                _lib.check(_lib.lib.cgat_lamb_step(tab.data_ptr(), ct.data_ptr(), co.data_ptr(), n_chunks, first.data_ptr(),
                                                   len(ps), group["lr"], b1, b2, group["eps"], group["weight_decay"],
                                                   ws.data_ptr(), torch.cuda.current_stream().cuda_stream),
                           "cgat_lamb_step")
        return loss


class _RobustLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, output, log_std, target, kind):
        for t in (output, log_std, target):
            if not t.is_cuda:
                raise RuntimeError("robust losses run on the GPU (no CPU fallback)")
        shape = torch.broadcast_shapes(output.shape, log_std.shape, target.shape)
        o, s, t = (x.to(torch.float32).expand(shape).contiguous().reshape(-1) for x in (output, log_std, target))
        n = o.numel()
        terms, go, gs = (torch.empty(n, dtype=torch.float32, device=o.device) for _ in range(3))
        with torch.cuda.device(o.device):
            _lib.check(_lib.lib.cgat_robust_loss(o.data_ptr(), s.data_ptr(), t.data_ptr(), n, kind, terms.data_ptr(),
                                                 go.data_ptr(), gs.data_ptr(), torch.cuda.current_stream().cuda_stream),
                       "cgat_robust_loss")
        ctx.save_for_backward(go, gs)
        ctx.shapes = (output.shape, log_std.shape, shape, n)
        return terms.mean()

    @staticmethod
    def backward(ctx, g):
        go, gs = ctx.saved_tensors
        so, ss, shape, n = ctx.shapes
        scale = g / n
        return ((go * scale).reshape(shape).sum_to_size(so), (gs * scale).reshape(shape).sum_to_size(ss), None, None)


def RobustL1(output, log_std, target):
    """mean( sqrt(2) |output - target| exp(-log_std) + log_std )      (CGAT/utils.py:30-37)"""
    return _RobustLoss.apply(output, log_std, target, 1)


def RobustL2(output, log_std, target):
    """mean( 0.5 (output - target)^2 exp(-2 log_std) + log_std )      (CGAT/utils.py:40-47)"""
    return _RobustLoss.apply(output, log_std, target, 2)


def cyclical_lr(period=100, cycle_mul=0.2, tune_mul=0.05):
    """Triangular cyclical schedule as a LambdaLR multiplier (CGAT/utils.py:50-64); `tune_mul` is accepted and unused,
    as in the reference."""
    def relative(it):
        cycle = math.floor(1 + it / period)
        x = abs(2 * (it / period - cycle) + 1)
        return max(0, (1 - x))

    return lambda it: cycle_mul + (1. - cycle_mul) * relative(it)
