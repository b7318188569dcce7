"""The N>1 path on CPU: two gloo ranks shard the crystals, run forward/backward on their shard,
average gradients with GradientAverager, and must end with the gradient of the full batch.
The model on CPU is the oracle (the HIP product has no CPU path); the sharding and the
bucketed all-reduce are the product code under test."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _loss_sum(model, batch, x, e, x0, lo, hi, A, K):
    """sum over crystals [lo,hi) of the squared node outputs, through one GATConvNodes layer"""
    n0, n1 = lo * A, hi * A
    e0, e1 = n0 * K, n1 * K
    ei = batch.edge_index[:, e0:e1] - n0
    y = model(x[n0:n1], ei.contiguous(), e[e0:e1], x0[n0:n1])
    return y.square().sum()


def _worker(rank, world, port, bucket_bytes, out, mode="plain"):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cgat_amd.dist import GradientAverager, init_from_env, shard_range
    from cgat_amd.graph import synthetic_batch
    from oracle import cgat_oracle as O
    torch.set_num_threads(2)
    r, w, dev = init_from_env("gloo")
    assert (r, w, dev.type) == (rank, world, "cpu")
    G, A, K, C = 10, 6, 4, 16
    batch, _ = synthetic_batch(G, A, K, seed=5)
    g = torch.Generator().manual_seed(6)
    N, E = batch.num_nodes, batch.edge_index.shape[1]
    x, e, x0 = torch.randn(N, C, generator=g), torch.randn(E, C, generator=g), torch.randn(N, C, generator=g)
    torch.manual_seed(1)
    model = O.GATConvNodes(C, C, C, 3, concat=True)
    unused = torch.nn.Parameter(torch.ones(3))            # a parameter that never receives a gradient
    half = torch.nn.Parameter(torch.ones(3))              # used on rank 0 only: must end with the mean on both ranks
    params = list(model.parameters()) + [unused, half]
    avg = GradientAverager(params, bucket_bytes=bucket_bytes)
    lo, hi = shard_range(G, rank, world)
    opt = torch.optim.AdamW(params, lr=1e-2, weight_decay=1e-2) if mode == "optim" else None
    for it in range(2):                                    # two steps: hooks must re-arm
        if mode == "views" or (mode == "optim" and it == 1):
            avg.zero_grad()                                # gradients accumulate straight into the bucket views
        else:
            for p in params:
                p.grad = None                              # what optimizer.zero_grad(set_to_none=True) leaves
        extra = (half.sum() * 3.0) if rank == 0 else 0.0
        if mode == "accumulate":                           # two micro-batches per rank (train.py:62 accumulate_grad_batches)
            mid = (lo + hi) // 2
            with avg.no_sync():
                (_loss_sum(model, batch, x, e, x0, lo, mid, A, K) * world / G + extra).backward()
            (_loss_sum(model, batch, x, e, x0, mid, hi, A, K) * world / G).backward()
        else:
            (_loss_sum(model, batch, x, e, x0, lo, hi, A, K) * world / G + extra).backward()
        avg.finish()
        if opt is not None:
            opt.step()
    if mode == "optim":
        # replicas must stay bit-identical: same mean gradient, same update, and the globally unused parameter
        # untouched (no weight decay applied to it on either rank)
        flat = torch.cat([p.detach().reshape(-1) for p in params])
        both = [torch.empty_like(flat) for _ in range(world)]
        dist.all_gather(both, flat)
        if rank == 0:
            out.put((float((both[0] - both[1]).abs().max()), bool((unused.detach() == 1).all()), len(avg.buckets)))
    elif rank == 0:
        torch.manual_seed(1)
        ref = O.GATConvNodes(C, C, C, 3, concat=True)
        (_loss_sum(ref, batch, x, e, x0, 0, G, A, K) / G).backward()
        # normalised by the layer's largest gradient: MH_A.fc_out.bias is zero by softmax shift
        # invariance, its value is rounding noise in any summation order
        scale = max(q.grad.abs().max().item() for q in ref.parameters())
        worst = 0.0
        for p, q in zip(model.parameters(), ref.parameters()):
            den = max(q.grad.abs().max().item(), 1e-3 * scale)
            worst = max(worst, (p.grad - q.grad).abs().max().item() / den)
        worst = max(worst, float((half.grad - 1.5).abs().max()))     # (3 + 0) / 2 ranks
        out.put((worst, unused.grad is None, len(avg.buckets)))
    dist.barrier()
    dist.destroy_process_group()


def _run(bucket_bytes, mode):
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, bucket_bytes, out, mode)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    return out.get()


@pytest.mark.parametrize("bucket_bytes", [64 << 20, 4096])
@pytest.mark.parametrize("mode", ["plain", "views", "accumulate"])
def test_two_rank_gradient_mean_equals_full_batch(bucket_bytes, mode):
    worst, unused_none, nb = _run(bucket_bytes, mode)
    assert worst <= 2e-5, worst
    assert unused_none
    assert nb >= (1 if bucket_bytes > 1 << 20 else 2)


def test_two_rank_replicas_identical_after_optimizer_steps():
    diff, unused_untouched, _ = _run(4096, "optim")
    assert diff == 0.0
    assert unused_untouched


def test_shard_range_partitions():
    from cgat_amd.dist import shard_range
    for n in (0, 1, 7, 4167):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
