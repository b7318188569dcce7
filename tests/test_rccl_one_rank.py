"""BASELINE configs[3] (data-parallel training, RCCL gradient all-reduce) on the hardware a one-GPU box has: a
one-rank RCCL communicator in a FRESH child process (tests/rccl_worker.py; reference: Lightning strategy='ddp',
CGAT/train.py:53-62).  The mean over one rank is the identity, so gradients and parameters must be bit-equal to the run
without the averager -- with the layer's side stream on and off, with bucket views and with set-to-none gradients, with
two accumulated micro-batches -- and from the second step on buckets must launch from the autograd hooks."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_data_parallel_path_over_one_rank_rccl():
    env = dict(os.environ)
    env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", HSA_ENABLE_IPC_MODE_LEGACY="0")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK"):
        env.pop(k, None)
    # a child process, not an exec: this (pytest) process may already have initialised the GPU
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "rccl_worker.py")], env=env, capture_output=True,
                       text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("RCCL_RESULT ")][-1]
    res = json.loads(line[len("RCCL_RESULT "):])
    assert res["backend"] == "nccl" and res["world"] == 1 and res["probe_ok"]
    for key, v in res["layer"].items():
        assert v["bit_equal"], (key, v)
        if "bucket=1MB grads" in key:
            # 38 MB of layer gradients in 1-MB buckets: after the first step they launch while backward is running
            assert v["launched_in_backward"] > 0, (key, v)
    t = res["trainer"]
    assert t["params_bit_equal"] and t["losses_equal"] and t["unused_stay_none"], t
    assert t["stats"]["launched_in_backward"] > 0 and t["stats"]["cold_skipped"] > 0, t
    # the visible all-reduce (a stub with NCCL's stream contract that doubles the buffer after a 1 ms spin): every
    # gradient exactly 2 x the plain one, i.e. no bucket is read before its Work is waited for and none is written after
    # its collective started
    for key, v in res["visible"].items():
        assert v.get("exactly_doubled", v.get("params_bit_equal")), (key, v)
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "r04_rccl_one_rank.json"), "w") as f:
        json.dump(res, f, indent=1)
