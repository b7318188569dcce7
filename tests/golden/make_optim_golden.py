"""Generates tests/golden/optim.npz from the UNMODIFIED reference: five steps of CGAT.lambs.JITLamb and
of torch.optim.AdamW (constructed as CGAT/lightning_module.py:328-335 does) on closed-form parameters/gradients,
CGAT.utils.RobustL1/RobustL2 values and gradients, and CGAT.utils.cyclical_lr samples.

    python tests/golden/make_optim_golden.py          (authoring container only; needs /root/reference)"""
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden
import optim_recipe as R


def main():
    make_golden.install_shims()
    sys.path.insert(0, "/root/reference")
    from CGAT.lambs import JITLamb
    from CGAT import utils as U
    out = {}
    for name, mk in (("adamw", lambda ps: torch.optim.AdamW(ps, lr=R.LR, weight_decay=R.WD)),
                     ("lamb", lambda ps: JITLamb(ps, lr=R.LR, weight_decay=R.WD))):
        ps = [torch.nn.Parameter(t.clone()) for t in R.params()]
        opt = mk(ps)
        for step in range(R.STEPS):
            for i, p in enumerate(ps):
                p.grad = R.grad(i, step, p.shape)
            opt.step()
            if step in (0, R.STEPS - 1):
                for i, p in enumerate(ps):
                    out[f"{name}.s{step}.p{i}"] = p.detach().numpy().copy()
    o, s, t = R.loss_inputs()
    for name, fn in (("l1", U.RobustL1), ("l2", U.RobustL2)):
        oo, ss = o.clone().requires_grad_(True), s.clone().requires_grad_(True)
        v = fn(oo, ss, t)
        go, gs = torch.autograd.grad(v, [oo, ss])
        out[f"{name}.value"], out[f"{name}.go"], out[f"{name}.gs"] = v.detach().numpy(), go.numpy(), gs.numpy()
    f = U.cyclical_lr(period=R.CLR_PERIOD, cycle_mul=0.1, tune_mul=0.05)
    out["clr"] = np.array([f(it) for it in R.CLR_ITS], dtype=np.float64)
    np.savez_compressed(os.path.join(HERE, "optim.npz"), **out)
    print("wrote optim.npz", len(out), "arrays", os.path.getsize(os.path.join(HERE, "optim.npz")) // 1024, "KiB")


if __name__ == "__main__":
    main()
