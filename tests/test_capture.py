"""hipGraph capture of the layer step and of the full stack at the reference's shipped batch size (64 crystals per GPU,
lightning_module.py:468-473), tests/capture_worker.py run as a child process: the replayed graph must give BIT-identical
outputs and gradients to the eager step, for the same inputs and after the static inputs were overwritten; two captures
in one process, one after the other."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_graphed_steps_equal_eager_bitwise():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "capture_worker.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    assert "CAPTURE_OK layer" in r.stdout and "CAPTURE_OK stack" in r.stdout, r.stdout[-2000:]
