"""CPU-side checks of the drop-in boundary: the shared library loads without a GPU, exports
every function include/cgat_hip.h declares, and the Python binding covers exactly that set.
No compute call is made here."""
import ctypes
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    src = open(os.path.join(ROOT, "include", "cgat_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(cgat_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from cgat_amd import _lib
    lib = ctypes.CDLL(_lib.LIB_PATH)
    names = declared_functions()
    assert len(names) >= 30
    for n in names:
        assert hasattr(lib, n), f"{n} declared in cgat_hip.h but not exported"


def test_binding_covers_header():
    from cgat_amd import _lib
    assert sorted(_lib.PROTOTYPES) == declared_functions()
    assert _lib.lib.cgat_abi_version() == _lib.ABI_VERSION


def test_no_fallback_on_cpu_tensors():
    import pytest
    import torch
    import cgat_amd as P
    layer = P.GATConvNodes(16, 16, 16, 3, concat=True)
    x = torch.randn(4, 16)
    ei = torch.tensor([[0, 1, 2, 3], [1, 2, 3, 0]])
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        layer(x, ei, torch.randn(4, 16), x)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        P.SimpleNetwork(16, 16, [16])(x)


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "cgat_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            src = open(os.path.join(pkg, f)).read()
            assert not re.search(r"^\s*(from|import)\s+oracle", src, flags=re.M), f
    # bench.py may use the oracle only inside its cpu_baseline legs
    import ast
    tree = ast.parse(open(os.path.join(ROOT, "bench.py")).read())
    for fn in [n for n in ast.walk(tree) if isinstance(n, ast.FunctionDef)]:
        uses = [n for n in ast.walk(fn) if isinstance(n, (ast.Import, ast.ImportFrom)) and
                "oracle" in (getattr(n, "module", None) or " ".join(a.name for a in n.names))]
        assert not uses or fn.name.startswith("cpu_baseline"), fn.name
    top = [n for n in tree.body if isinstance(n, (ast.Import, ast.ImportFrom)) and
           "oracle" in (getattr(n, "module", None) or " ".join(a.name for a in n.names))]
    assert not top


def test_state_dict_layout_matches_reference():
    import torch
    import cgat_amd as P
    from oracle import cgat_oracle as O
    kw = dict(msg_heads=3, update_edges=True)
    p, o = P.CGAtNet(200, 32, 2, **kw), O.CGAtNet(200, 32, 2, **kw)
    sp, so = p.state_dict(), o.state_dict()
    assert list(sp) == list(so)
    assert all(sp[k].shape == so[k].shape and sp[k].dtype == so[k].dtype for k in sp)
    # same construction order => same initial values under the same seed
    torch.manual_seed(1); p = P.CGAtNet(200, 32, 2, **kw)
    torch.manual_seed(1); o = O.CGAtNet(200, 32, 2, **kw)
    assert all(torch.equal(a, b) for a, b in zip(p.state_dict().values(), o.state_dict().values()))
