"""Dev tool (GPU box): per-tensor error of golden cases with the small-row programs on (default) and off
(CGAT_ROWPROG_MAX_ROWS=0: the generic engine), each in a fresh process.
usage: python tools/rowprog_path_probe.py tiny.npz net_embed [name-substring]"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if len(sys.argv) > 1 and sys.argv[1] == "--child":
    sys.path[:0] = [ROOT, os.path.join(ROOT, "tests"), os.path.join(ROOT, "tests", "golden")]
    import recipe
    from golden_util import check_case
    from test_hip_golden import product_ns
    fname, cname, sub = sys.argv[2], sys.argv[3], (sys.argv[4] if len(sys.argv) > 4 else "")
    cases = recipe.tiny_cases(product_ns()) if fname.startswith("tiny") else recipe.base_cases(product_ns())
    rep = []
    try:
        check_case(fname, cname, cases[cname], device="cuda:0", tol=1e-4, report=rep)
    except AssertionError as ex:
        print("ASSERT", str(ex)[:200])
    for (c, name, err, ref, nf) in rep:
        allowed = max(1e-4 * ref, 4 * nf)
        if sub in name and (err > 0.3 * allowed or sub):
            print(f"  {name:90s} err {err:.3e} ref {ref:.3e} nf {nf:.3e} err/allowed {err / max(allowed, 1e-300):.2f}")
    sys.exit(0)
for rows, pyrows in (("2048", "2048"), ("0", "0"), ("2048", "0"), ("0", "2048")):
    env = dict(os.environ, CGAT_ROWPROG_MAX_ROWS=rows, CGAT_ROWPROG_PY_MAX_ROWS=pyrows)
    print(f"== CGAT_ROWPROG_MAX_ROWS={rows} (library)  CGAT_ROWPROG_PY_MAX_ROWS={pyrows} (modules)")
    sys.stdout.flush()
    subprocess.call([sys.executable, __file__, "--child"] + sys.argv[1:], env=env)
