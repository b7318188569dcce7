"""Comparison helpers for the golden fixtures (tests only)."""
import os
import types

import numpy as np
import torch

import recipe

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load(fname):
    if fname not in _cache:
        _cache[fname] = np.load(os.path.join(GOLDEN_DIR, fname))
    return _cache[fname]


def case_arrays(fname, cname):
    z = load(fname)
    pre = cname + "/"
    return {k[len(pre):]: z[k] for k in z.files if k.startswith(pre)}


def maxnorm_rel(a, b):
    """||a-b||_inf / ||b||_inf  -- the parity metric of BASELINE.md §4 (element-wise relative error is
    meaningless on near-zero entries: the reference differs from its own fp64 run by up to 0.2 there)."""
    a = np.asarray(a, dtype=np.float64)
    b = np.asarray(b, dtype=np.float64)
    den = np.abs(b).max()
    if den == 0:
        return float(np.abs(a).max())
    return float(np.abs(a - b).max() / den)


def check_case(fname, cname, case, *, device="cpu", tol=1e-4, grad_tol=None):
    """Run `case` (module built from the namespace under test) and compare output, input grads,
    parameter grads (full or probe) and None-ness of grads with the reference's fixture."""
    grad_tol = tol if grad_tol is None else grad_tol
    ref = case_arrays(fname, cname)
    y, grads, _ = recipe.run_case(case, torch.float32, device=device)
    errs = {"out": maxnorm_rel(y.detach().cpu().numpy(), ref["out"])}
    assert errs["out"] <= tol, f"{cname}: out err {errs['out']:.3e}"
    for name, g in grads.items():
        if name + ".none" in ref:
            # reference leaves this gradient unset; we may return None or exact zeros
            assert g is None or float(g.abs().max()) == 0.0, f"{cname}: {name} should have no gradient"
            continue
        assert g is not None, f"{cname}: {name} has no gradient but the reference has one"
        if name in ref:
            want = ref[name]
            e = maxnorm_rel(g.detach().cpu().numpy(), want)
            scale = np.abs(want).max()
        else:
            want = ref["gpn." + name[3:]]
            got = recipe.grad_probe(g)
            # compare sum / ramp-dot relative to the L2 norm (they can cancel to ~0), norm and max relatively
            nrm = max(want[1], 1e-30)
            e = max(abs(got[0] - want[0]) / (nrm * np.sqrt(g.numel())) * 10, abs(got[1] - want[1]) / nrm,
                    abs(got[2] - want[2]) / max(want[2], 1e-30), abs(got[3] - want[3]) / (nrm * np.sqrt(g.numel())) * 10,
                    np.abs(got[4:] - want[4:]).max() / max(want[2], 1e-30))
            scale = want[2]
        errs[name] = e
        if scale < 1e-12:   # gradient that is zero up to rounding in the reference (softmax shift invariance)
            continue
        assert e <= grad_tol, f"{cname}: {name} err {e:.3e}"
    return errs
