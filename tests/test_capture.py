"""hipGraph capture of the layer step and of the full stack at the reference's shipped batch size (64 crystals per GPU,
lightning_module.py:468-473), tests/capture_worker.py run as a child process: the replayed graph must give BIT-identical
outputs and gradients to the eager step, for the same inputs and after the static inputs were overwritten; two captures
in one process, one after the other."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_SHA = {}


@pytest.mark.gpu
@pytest.mark.parametrize("one_by_one", [False, True])
def test_graphed_steps_equal_eager_bitwise(one_by_one):
    """one_by_one: the hypernetwork's operands prepared tensor by tensor (CGAT_NO_TPREP_BATCH=1: the fall-back of the
    batched preparation) -- the path on which a captured 4-byte hipMemsetAsync made the second replay return NaN in
    round 4 (csrc/rowops.hip, fill_launch)."""
    env = dict(os.environ)
    if one_by_one:
        env["CGAT_NO_TPREP_BATCH"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capture_worker.py")], capture_output=True, text=True,
                       timeout=900, env=env)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "CAPTURE_OK layer" in r.stdout and "CAPTURE_OK stack" in r.stdout, r.stdout[-2000:]
    _SHA[one_by_one] = [ln.split()[1] for ln in r.stdout.splitlines() if ln.startswith("STACK_SHA1")][0]
    if len(_SHA) == 2:   # the batched and the one-by-one preparation build the same operand images: bit-equal steps
        assert _SHA[False] == _SHA[True], _SHA
