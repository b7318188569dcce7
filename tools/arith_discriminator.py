"""Dev tool: the kernel-level arithmetic discriminator (tests/arith_cases.py; asserted by
tests/test_hip_kernels.py::test_operand_width_discriminator) in every arithmetic mode."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
from cgat_amd import _lib, ops
import arith_cases as A

dev = torch.device("cuda:0")
if __name__ == "__main__":
    for name, fn in (("bilinear_rows", A.run_rows), ("bilinear_dual", A.run_dual), ("bilinear_wgrad", A.run_wgrad)):
        print(name, {k: (f"{v:.3e}" if not isinstance(v, tuple) else tuple(f"{x:.3e}" for x in v)) for k, v in fn(_lib, ops, dev).items()})
