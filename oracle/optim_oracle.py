"""CPU restatement (torch fp32 on the host) of the optimiser steps, losses and schedule after the hot path
(SURVEY 8 f4).  TEST INFRASTRUCTURE ONLY.

  adamw_step : the update rule of torch.optim.AdamW, which the reference constructs at
               CGAT/lightning_module.py:328-331 (third-party torch; restated from its documented algorithm in the op
               order of torch 2.x's single-tensor path and pinned by running torch.optim.AdamW itself in the fixtures)
  lamb_step  : CGAT/lambs.py:155-181 lamb_kernel, as JITLamb.step (226-262) drives it
  robust_l1/l2, cyclical_lr : CGAT/utils.py:30-64
Pinned by tests/golden/optim.npz, recorded from the unmodified reference (tests/golden/make_optim_golden.py)."""
import math

import numpy as np
import torch


def adamw_step(p, g, m, v, step, lr=1e-3, betas=(0.9, 0.999), eps=1e-8, weight_decay=1e-2):
    b1, b2 = betas
    p = p * (1 - lr * weight_decay)
    m = torch.lerp(m, g, 1 - b1)
    v = v * b2 + (1 - b2) * g * g
    bc1, bc2 = 1 - b1 ** step, 1 - b2 ** step
    denom = v.sqrt() / math.sqrt(bc2) + eps
    p = p - (lr / bc1) * (m / denom)
    return p, m, v


def lamb_step(p, g, m, v, lr=1e-3, betas=(0.9, 0.999), eps=1e-6, weight_decay=0.0):
    b1, b2 = betas
    m = m * b1 + (1 - b1) * g
    v = v * b2 + (1 - b2) * (g * g)
    s = m / (v.sqrt() + eps) + weight_decay * p
    wn = p.norm(p=2).clamp(0, 10)
    an = s.norm(p=2)
    r = wn / (an + eps)
    r = torch.where(wn == 0, torch.ones_like(r), r)
    r = torch.where(an == 0, torch.ones_like(r), r)
    return p - lr * r * s, m, v


def robust_l1(output, log_std, target):
    return torch.mean(np.sqrt(2.0) * torch.abs(output - target) * torch.exp(-log_std) + log_std)


def robust_l2(output, log_std, target):
    return torch.mean(0.5 * torch.pow(output - target, 2.0) * torch.exp(-2.0 * log_std) + log_std)


def cyclical_lr(period=100, cycle_mul=0.2, tune_mul=0.05):
    def relative(it):
        cycle = math.floor(1 + it / period)
        x = abs(2 * (it / period - cycle) + 1)
        return max(0, (1 - x))
    return lambda it: cycle_mul + (1. - cycle_mul) * relative(it)
