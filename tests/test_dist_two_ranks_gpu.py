"""BASELINE configs[3] at world size 2 on the hardware a one-GPU box has: two FRESH child processes, both on cuda:0,
backend gloo (tests/dist2_worker.py).  Complements tests/test_rccl_one_rank.py (one rank over real RCCL) and
tests/test_dist_gloo.py (two ranks, oracle network on CPU): here the HIP path itself runs on two ranks with different
batches."""
import json
import os
import socket
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_hip_path():
    port = _free_port()
    procs = []
    for rank in range(2):
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2",
                   HSA_ENABLE_IPC_MODE_LEGACY="0")
        env.pop("CGAT_DIST_FORCE", None)
        # child processes, not an exec: this (pytest) process may already have initialised the GPU
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dist2_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=900) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
    results = []
    for so, _ in outs:
        line = [l for l in so.splitlines() if l.startswith("DIST2_RESULT ")][-1]
        results.append(json.loads(line[len("DIST2_RESULT "):]))
    assert sorted(r["rank"] for r in results) == [0, 1]
    for r in results:
        for key, v in r["layer"].items():
            assert v["mean_bit_equal"], (r["rank"], key, v)
        assert r["layer"]["bucket=1MB"]["launched_in_backward"] > 0, r["layer"]       # overlapped from the second step on
        t = r["trainer"]
        assert t["replicas_bit_identical"] and t["unused_none"] > 0, t
    assert results[0]["trainer"]["losses"] != results[1]["trainer"]["losses"]            # the ranks did see different crystals
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r04_dist_two_ranks_one_gpu.json"), "w") as f:
        json.dump(results, f, indent=1)
