"""Comparison helpers for the golden fixtures (tests only)."""
import os
import types

import numpy as np
import torch

import recipe

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}



def floor_open(ref_max, case_scale):
    """Is the zero-gradient floor `1e-6 * case scale` open to this tensor?  Only to numerically zero gradients (|fp64
    reference| <= 1e-7 of the case's largest gradient: below fp32's resolution of the case) -- except in the DIAGNOSTIC
    arithmetic mode `f32` (set_bilinear_mode("f32") / CGAT_BILINEAR_MODE=f32: f32-input MFMA / plain fp32 chains, never a default and not what any
    number is quoted in), which keeps round 3's criterion (floor open to every tensor): its accumulators carry the
    un-alternated rounding bias the split modes cancel, and four cancellation-heavy tensors of the
    sin-filled fixtures sit at 1.0-2.0e-4 of their value there."""
    from cgat_amd import ops                            # the arithmetic mode the library is actually in, not the
    return ref_max <= 1e-7 * case_scale or ops.get_bilinear_mode() == "f32"   # environment variable it started from

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


# Tensors the reference itself cannot resolve to `tol` in fp32 are admitted up to NOISE_MULT times the reference's own
# fp32-vs-fp64 deviation nf.  The difference of two independent fp32 evaluations is ~sqrt(2) x one evaluation's error in
# max-norm, and nf is ONE sample of that error, so small multiples are the expected range; 4 (16 until round 3) leaves
# room for that and nothing else.
NOISE_MULT = 4.0


def check_case(fname, cname, case, *, device="cpu", tol=1e-4, grad_tol=None, report=None):
    """Run `case` (module built from the namespace under test) and compare output, input grads,
    parameter grads (full or probe) and None-ness of grads with the reference's fixture.

    Criterion per tensor:  ||got - ref||_inf <= max(tol * ||ref||_inf, NOISE_MULT * nf)
    where nf = ||ref_fp32 - ref_fp64||_inf is the reference's own rounding deviation recorded in the
    fixture.  The second term only matters for tensors the reference itself cannot resolve to
    `tol` in fp32 (gradients that are zero by symmetry, or dominated by cancellation)."""
    grad_tol = tol if grad_tol is None else grad_tol
    ref = case_arrays(fname, cname)
    y, grads, _ = recipe.run_case(case, torch.float32, device=device)
    errs = {"out": maxnorm_rel(y.detach().cpu().numpy(), ref["out"])}
    nf_out = np.abs(ref["out"].astype(np.float64) - ref["out_f64"]).max() / max(np.abs(ref["out_f64"]).max(), 1e-300)
    assert errs["out"] <= max(tol, NOISE_MULT * nf_out), f"{cname}: out err {errs['out']:.3e}"
    # gradients below 1e-6 of the largest gradient of the case are zero at fp32 resolution
    # (fp32 epsilon times the ~sqrt(E) growth of an E-term sum)
    case_scale = max([float(v[1]) for k, v in ref.items() if k.startswith("nf.")] + [0.0])
    for name, g in grads.items():
        if name + ".none" in ref:
            # reference leaves this gradient unset; we may return None or exact zeros
            assert g is None or float(g.abs().max()) == 0.0, f"{cname}: {name} should have no gradient"
            continue
        assert g is not None, f"{cname}: {name} has no gradient but the reference has one"
        nf_abs, ref_max = ref["nf." + name]
        if name in ref:
            want = ref[name]
            abs_err = np.abs(g.detach().cpu().numpy().astype(np.float64) - want).max()
        else:
            want = ref["gpn." + name[3:]]
            got = recipe.grad_probe(g)
            n = g.numel()
            # probe entries: sum, L2, max, ramp-dot, first 8 values -> each converted to a rigorous lower bound
            # of the max element error: |sum e| <= n max|e|, |ramp . e| <= n max|e|, | ||a|| - ||b|| | <= sqrt(n) max|e|
            abs_err = max(abs(got[0] - want[0]) / n, abs(got[1] - want[1]) / np.sqrt(n), abs(got[2] - want[2]),
                          abs(got[3] - want[3]) / n, np.abs(got[4:] - want[4:]).max())
        # the zero floor is only open to numerically zero gradients: |ref| below fp32's resolution of the case
        # (tests/test_hip_golden.py NUM_ZERO)
        floor = 1e-6 * case_scale if floor_open(ref_max, case_scale) else 0.0
        allowed = max(grad_tol * ref_max, NOISE_MULT * nf_abs, floor)
        errs[name] = abs_err / max(ref_max, 1e-300)
        if report is not None:
            report.append((cname, name, abs_err, ref_max, nf_abs))
        assert abs_err <= allowed, (f"{cname}: {name} abs err {abs_err:.3e} > allowed {allowed:.3e} "
                                    f"(|ref| {ref_max:.3e}, reference fp32 noise {nf_abs:.3e})")
    return errs
