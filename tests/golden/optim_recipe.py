"""Closed-form parameters, gradients and loss inputs for the optimiser / loss fixtures."""
import torch

LR, WD, STEPS = 3e-3, 1e-2, 5
SHAPES = [(128, 128), (384,), (3, 256, 1), (1,), (40000,), (7, 33)]   # incl. a >1-chunk tensor, a scalar-like one
CLR_PERIOD, CLR_ITS = 70, [0, 1, 17, 35, 69, 70, 71, 104, 105, 139, 140, 1000, 12345]


def _sin(n, a, b, scale):
    i = torch.arange(n, dtype=torch.float64)
    return (scale * torch.sin(a * i + b)).to(torch.float32)


def params():
    out = []
    for k, sh in enumerate(SHAPES):
        n = 1
        for d in sh:
            n *= d
        t = _sin(n, 0.37 + 0.01 * k, 0.1 * k, 0.5 if k != 4 else 0.02).reshape(sh)
        if k == 3:
            t = torch.zeros(sh)            # zero weight norm: LAMB's trust-ratio guard
        out.append(t)
    return out


def grad(i, step, shape):
    n = 1
    for d in shape:
        n *= d
    return _sin(n, 0.91 + 0.07 * i, 0.3 * step + i, 0.2).reshape(shape)


def loss_inputs():
    n = 257
    o, s, t = _sin(n, 0.31, 0.0, 2.0), _sin(n, 0.17, 1.0, 0.7), _sin(n, 0.53, 2.0, 2.0)
    o[5] = t[5]                            # |o - t| = 0: the sign(0) = 0 branch of RobustL1's gradient
    return o.reshape(-1, 1), s.reshape(-1, 1), t.reshape(-1, 1)
