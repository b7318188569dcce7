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


def _net_worker(rank, world, port, out):
    """The shipped network structure (update_edges=True, no_hyper=True: Edge.MH_A / Edge.MH_M never receive a gradient,
    reference CGAT.py:224-225) over two ranks: after the first step those parameters are cold and must not gate a
    bucket, so buckets launch from the hooks, i.e. while backward is still running."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cgat_amd.dist import GradientAverager, init_from_env, shard_range
    from cgat_amd.graph import synthetic_batch
    from oracle import cgat_oracle as O
    torch.set_num_threads(2)
    init_from_env("gloo")
    kw = dict(msg_heads=2, neighbor_number=4, update_edges=True)
    torch.manual_seed(1)
    net = O.CGAtNet(200, 16, 2, **kw)
    late = torch.nn.Parameter(torch.ones(5))              # unused in steps 0-1 (goes cold), used on rank 1 in step 2
    params = list(net.parameters()) + [late]
    avg = GradientAverager(params, bucket_bytes=64 << 10)
    G = 4
    shards = [synthetic_batch(2, 5, 4, seed=10 + r) for r in range(world)]      # each rank: its own two crystals
    b, roost = shards[rank]
    log = []
    for it in range(5):
        avg.zero_grad()
        loss = net(b, roost)[:, 0].sum() * world / G
        if it == 2 and rank == 1:
            loss = loss + late.sum() * 4.0
        before = dict(avg.stats)
        loss.backward()
        in_bwd = avg.stats["launched_in_backward"] - before["launched_in_backward"]
        n_hot, n_buckets = avg.n_hot, len(avg.buckets)
        avg.finish()
        log.append(dict(in_backward=in_bwd, in_finish=avg.stats["launched_in_finish"] - before["launched_in_finish"],
                        cold_reduced=avg.stats["cold_reduced"] - before["cold_reduced"],
                        cold_skipped=avg.stats["cold_skipped"] - before["cold_skipped"], n_hot=n_hot,
                        n_buckets=n_buckets, late=None if late.grad is None else float(late.grad[0])))
    if rank == 0:
        torch.manual_seed(1)
        ref = O.CGAtNet(200, 16, 2, **kw)
        tot = 0.0
        for bb, rr in shards:
            tot = tot + ref(bb, rr)[:, 0].sum() / G
        tot.backward()
        scale = max(q.grad.abs().max().item() for q in ref.parameters() if q.grad is not None)
        worst, none_mismatch = 0.0, 0
        for (name, p), q in zip(net.named_parameters(), ref.parameters()):
            if (p.grad is None) != (q.grad is None):
                none_mismatch += 1
            elif q.grad is not None:
                worst = max(worst, (p.grad - q.grad).abs().max().item() / max(q.grad.abs().max().item(), 1e-3 * scale))
        n_unused = sum(1 for q in ref.parameters() if q.grad is None)
        out.put((worst, none_mismatch, n_unused, log))
    dist.barrier()
    dist.destroy_process_group()


def test_unused_parameters_do_not_gate_the_buckets():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_net_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=240)
        assert p.exitcode == 0
    worst, none_mismatch, n_unused, log = out.get()
    assert worst <= 2e-5 and none_mismatch == 0, (worst, none_mismatch)
    assert n_unused >= 16                                  # Edge.MH_A / Edge.MH_M of both layers + the last Edge.Pooling_NN
    # step 0 has no history: the never-used parameters sit in every bucket and nothing can launch before finish()
    assert log[0]["in_backward"] == 0 and log[0]["in_finish"] == log[0]["n_buckets"] >= 2
    # from step 1 on they are cold: every hot bucket launches from the hooks, the cold ones are skipped altogether
    for it in (1, 4):
        assert log[it]["in_backward"] == log[it]["n_hot"] >= 2 and log[it]["in_finish"] == 0, log[it]
        assert log[it]["cold_skipped"] >= 1 and log[it]["cold_reduced"] == 0
    # step 2: a cold parameter gets a gradient on ONE rank -> its cold bucket is reduced in finish(), mean on both ranks
    assert log[2]["cold_reduced"] >= 1 and log[2]["late"] == 2.0, log[2]
    assert log[1]["late"] is None and log[3]["late"] is None
    # step 3: that parameter is hot (used somewhere in step 2) but unused again: it holds its bucket back until finish()
    # for this one step and is cold again from step 4 on
    assert log[3]["in_finish"] >= 1 and log[3]["n_hot"] >= log[1]["n_hot"]


def _static_worker(rank, world, port, out):
    """static_graph=True: after two steps with the same global bitmap the cold set freezes -- no bitmap all-reduce, no
    host synchronisation in finish() -- the mean stays right, and a cold parameter that gets a gradient raises."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                      LOCAL_RANK=str(rank))
    from cgat_amd.dist import GradientAverager, init_from_env
    torch.set_num_threads(2)
    init_from_env("gloo")
    torch.manual_seed(1)
    lin = torch.nn.Linear(6, 4)
    dead = torch.nn.Parameter(torch.ones(3))
    params = list(lin.parameters()) + [dead]
    avg = GradientAverager(params, bucket_bytes=64, static_graph=True)
    x = torch.arange(12, dtype=torch.float32).reshape(2, 6) * (rank + 1)
    frozen, calls = [], []
    real = dist.all_reduce

    def counting(t, *a, **k):
        calls.append(t.dtype)
        return real(t, *a, **k)
    dist.all_reduce = counting
    for it in range(4):
        avg.zero_grad()
        n0 = sum(1 for d in calls if d == torch.int32)
        lin(x).sum().backward()
        avg.finish()
        frozen.append((avg._frozen, sum(1 for d in calls if d == torch.int32) - n0))
    want_w = (torch.arange(12, dtype=torch.float32).reshape(2, 6).sum(0) * 1.5).expand(4, 6)   # mean of ranks 1x and 2x
    ok = bool(torch.allclose(lin.weight.grad, want_w)) and dead.grad is None
    # a cold parameter receives a gradient on rank 1 ONLY: the violation must surface on BOTH ranks (a rank raising
    # alone would leave the other one waiting in its next collective) and in the SAME finish() on both: the one
    # VIOL_LAG = 2 steps after the violation (every flag is queued and read at a fixed distance, none is overwritten)
    raised_at = []
    for it in range(4):
        avg.zero_grad()
        try:
            (lin(x).sum() + (dead.sum() if (rank == 1 and it == 0) else 0.0)).backward()
            avg.finish()
        except RuntimeError as ex:
            if "static_graph" in str(ex):
                raised_at.append(it)
            break
    out.put((rank, frozen, ok, raised_at == [GradientAverager.VIOL_LAG]))
    dist.all_reduce = real
    dist.barrier()
    dist.destroy_process_group()


def test_static_graph_freezes_the_cold_set():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    port = _free_port()
    procs = [ctx.Process(target=_static_worker, args=(r, 2, port, out)) for r in range(2)]
    for p in procs:
        p.start()
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    got = sorted(out.get() for _ in range(2))
    assert [g[0] for g in got] == [0, 1]
    for _, frozen, ok, raised in got:
        # step 0 learns the bitmap, step 1 confirms it (both reduce the int32 bitmap), from step 2 on: frozen, no bitmap
        assert [f for f, _ in frozen] == [False, True, True, True], frozen
        assert [n for _, n in frozen] == [1, 1, 0, 0], frozen
        assert ok and raised                             # ... and BOTH ranks raise, at the finish() after the violation


def _two_obs_worker(port, out):
    """No parameter is cold in step 1: the initial (empty) cold set must not count as an observation."""
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK="0",
                      CGAT_DIST_FORCE="1")
    from cgat_amd.dist import GradientAverager, init_from_env
    init_from_env("gloo")
    a, b = torch.nn.Parameter(torch.ones(3)), torch.nn.Parameter(torch.ones(3))
    avg = GradientAverager([a, b], static_graph=True, force=True)
    states = []
    avg.zero_grad(); (a.sum() + b.sum()).backward(); avg.finish(); states.append(avg._frozen)      # step 1: all used
    avg.zero_grad(); a.sum().backward(); avg.finish(); states.append((avg._frozen, b.grad is None))  # step 2: b unused
    out.put(states)
    dist.destroy_process_group()


def test_static_graph_needs_two_observed_steps():
    ctx = mp.get_context("spawn")
    out = ctx.SimpleQueue()
    pr = ctx.Process(target=_two_obs_worker, args=(_free_port(), out))
    pr.start()
    pr.join(timeout=120)
    assert pr.exitcode == 0
    s1, (s2, b_none) = out.get()
    assert s1 is False and s2 is False and b_none      # not frozen after one step; the unused parameter ends with None


def test_force_without_process_group_raises():
    from cgat_amd.dist import GradientAverager
    if dist.is_initialized():
        pytest.skip("a process group is initialised in this process")
    with pytest.raises(RuntimeError):
        GradientAverager([torch.nn.Parameter(torch.ones(2))], force=True)


def test_shard_range_partitions():
    from cgat_amd.dist import shard_range
    for n in (0, 1, 7, 4167):
        for w in (1, 2, 3, 8):
            parts = [shard_range(n, r, w) for r in range(w)]
            assert parts[0][0] == 0 and parts[-1][1] == n
            assert all(a[1] == b[0] for a, b in zip(parts, parts[1:]))
            sizes = [b - a for a, b in parts]
            assert max(sizes) - min(sizes) <= 1
