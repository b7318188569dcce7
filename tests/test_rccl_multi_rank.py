"""BASELINE configs[3] on real RCCL at world size > 1.  Enables itself on the first box with two or more visible GPUs
(the authoring pool has one GPU per box: there the tests skip, and the one-rank RCCL and two-ranks-on-one-GPU tests stand
in): N child processes started BEFORE any GPU call in them, backend nccl (= RCCL), one device each.

* world 2 over RCCL vs world 2 over gloo (both ranks on cuda:0, host-memory collectives): the mean gradient is
  BIT-EQUAL (a two-term sum has one order), on every rank;
* world = all visible devices over RCCL: every rank ends with the same mean gradient bytes, replicas bit-identical after
  three AdamW steps, world_size == N, buckets launched from the backward hooks.
Also (one GPU is enough): bench.py under torch.distributed.run with two ranks emits ONE line with two per-rank timings.
Reference: Lightning strategy='ddp', CGAT/train.py:53-62."""
import json
import os
import socket
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
N_DEV = torch.cuda.device_count()          # counting devices does not initialise the GPU in this process


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _run_world(world, backend):
    port = _free_port()
    procs = []
    for rank in range(world):
        env = dict(os.environ)
        env.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world),
                   HSA_ENABLE_IPC_MODE_LEGACY="0", CGAT_TEST_BACKEND=backend)
        env.pop("CGAT_DIST_FORCE", None)
        env.pop("CGAT_DIST_SHARE_GPU", None)
        procs.append(subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "rcclN_worker.py")], env=env,
                                      stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=1200) for p in procs]
    for p, (so, se) in zip(procs, outs):
        assert p.returncode == 0, so[-3000:] + se[-3000:]
    res = []
    for so, _ in outs:
        line = [l for l in so.splitlines() if l.startswith("RCCLN_RESULT ")][-1]
        res.append(json.loads(line[len("RCCLN_RESULT "):]))
    return sorted(res, key=lambda r: r["rank"])


@pytest.mark.gpu
@pytest.mark.skipif(N_DEV < 2, reason="needs two or more visible GPUs (RCCL at world size > 1)")
def test_rccl_world2_bit_equal_to_gloo():
    rccl = _run_world(2, "nccl")
    gloo = _run_world(2, "gloo")
    for r in rccl:
        assert r["backend"] == "nccl" and r["world_size"] == 2 and r["device"] == f"cuda:{r['rank']}", r
        assert r["mean_identical_across_ranks"] and r["replicas_bit_identical"], r
    for a, b in zip(rccl, gloo):
        assert a["mean_grad_sha256"] == b["mean_grad_sha256"], (a, b)      # (g0 + g1) / 2 whatever carried it
        assert a["params_sha256"] == b["params_sha256"], (a, b)            # and the same trained replicas
    _save("r05_rccl_world2.json", {"rccl": rccl, "gloo": gloo})


@pytest.mark.gpu
@pytest.mark.skipif(N_DEV < 2, reason="needs two or more visible GPUs (RCCL at world size > 1)")
def test_rccl_all_devices():
    res = _run_world(N_DEV, "nccl")
    assert [r["rank"] for r in res] == list(range(N_DEV))
    for r in res:
        assert r["backend"] == "nccl" and r["world_size"] == N_DEV and r["device"] == f"cuda:{r['rank']}", r
        assert r["mean_identical_across_ranks"] and r["replicas_bit_identical"], r
        assert r["launched_in_backward"] > 0, r                               # overlapped with backward from the hooks
    assert len({tuple(r["losses"]) for r in res}) > 1                         # the ranks did see different crystals
    _save(f"r05_rccl_world{N_DEV}.json", res)


@pytest.mark.gpu
def test_bench_two_ranks_emits_one_line_with_per_rank_timings():
    """The driver's SCALE command (python -m torch.distributed.run --nproc-per-node N bench.py --gpus N ...) on what a
    one-GPU box has: two ranks sharing cuda:0 over gloo (CGAT_DIST_BACKEND / CGAT_DIST_SHARE_GPU are the test hooks of
    cgat_amd.dist.init_from_env).  Rank 0 prints exactly one JSON line; it carries both ranks' timings, the world size
    and the whole-job value."""
    env = dict(os.environ)
    env.update(CGAT_DIST_BACKEND="gloo", CGAT_DIST_SHARE_GPU="1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1",
           "--graphs", "256", "--no-extra-legs", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, lines
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["scaling"] == "weak"
    assert len(d["ranks"]["ms_per_step_per_rank"]) == 2 and d["ranks"]["world_size"] == 2, d["ranks"]
    assert abs(d["value"] - 2 * d["config"]["edges_per_rank"] / (d["ms_per_step"] * 1e-3)) <= 1e-6 * d["value"]


def _save(name, obj):
    out = os.path.join(ROOT, "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, name), "w") as f:
        json.dump(obj, f, indent=1)
