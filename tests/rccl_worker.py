"""Child process of tests/test_rccl_one_rank.py: the data-parallel path (cgat_amd.dist.GradientAverager, the trainer)
over a ONE-RANK RCCL communicator on the box's single GPU.  It is started as a fresh process -- the process group is
created before anything else touches the GPU -- and prints one JSON object with what it measured; the test asserts.

What a one-rank communicator exercises: backend "nccl" (= RCCL) initialisation, asynchronous all-reduces issued from
post-accumulate-grad hooks while the layer's backward still has work queued on its side stream (ops.NodeLayerFn),
the ordering of RCCL's internal stream against the compute streams, the used-bitmap reduction, the hot / cold bucket
re-layout and no_sync accumulation.  The mean over one rank is the identity, so every result must be BIT-equal to the
run without the averager."""
import copy
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29533")
    os.environ.update(RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", CGAT_DIST_FORCE="1")
    import numpy as np
    import torch
    import torch.distributed as dist
    from cgat_amd.dist import GradientAverager, init_from_env
    rank, world, dev = init_from_env()                     # backend nccl: RCCL, one rank
    assert dist.is_initialized() and dist.get_backend() == "nccl" and world == 1 and dev.type == "cuda"
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.graph import synthetic_dataset_dict
    res = {"backend": dist.get_backend(), "world": dist.get_world_size()}
    probe = torch.arange(8, dtype=torch.float32, device=dev)
    dist.all_reduce(probe)
    res["probe_ok"] = bool((probe.cpu() == torch.arange(8, dtype=torch.float32)).all())

    # ---- A: one GATConvNodes layer at the benchmark widths, gradients with / without the averager, side stream on / off
    b, _ = P.synthetic_batch(96, 20, 12, seed=3)
    g = torch.Generator().manual_seed(4)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)
    params = list(layer.parameters())

    def layer_grads(avg, mode, micro=1):
        for p in params:
            p.grad = None
        if avg is not None and mode == "views":
            avg.zero_grad()
        for k in range(micro):
            y = layer(x, ei, e, x0)
            if avg is not None and k < micro - 1:
                with avg.no_sync():
                    y.backward(cot)
            else:
                y.backward(cot)
        if avg is not None:
            avg.finish()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]

    res["layer"] = {}
    for overlap in (True, False):
        ops.set_overlap_wgrad(overlap)
        plain = layer_grads(None, "none")
        plain2 = layer_grads(None, "none", micro=2)
        for bucket in (64 << 20, 1 << 20):
            avg = GradientAverager(params, bucket_bytes=bucket, force=True)
            for mode in ("none", "views"):
                for rep in range(3):                        # step 0 lays out, later steps launch from the hooks
                    got = layer_grads(avg, mode)
                key = f"overlap={int(overlap)} bucket={bucket >> 20}MB grads={mode}"
                res["layer"][key] = {"bit_equal": all(torch.equal(a, c) for a, c in zip(got, plain)),
                                     "launched_in_backward": avg.stats["launched_in_backward"],
                                     "launched_in_finish": avg.stats["launched_in_finish"], "buckets": len(avg.buckets)}
            got2 = layer_grads(avg, "views", micro=2)
            res["layer"][f"overlap={int(overlap)} bucket={bucket >> 20}MB no_sync x2"] = {
                "bit_equal": all(torch.equal(a, c) for a, c in zip(got2, plain2))}
            avg.close()
    ops.set_overlap_wgrad(None)                            # back to the mode's default

    # ---- B: two steps of the training step (BASELINE configs[3]) with the averager forced on == without it
    data, emb = synthetic_dataset_dict(60, (2, 40), 24, seed=3)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device=dev)
    torch.manual_seed(0)
    net = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to(dev)
    net2 = copy.deepcopy(net)
    tr_plain = P.DataParallelTrainer(net, ds, lr=1e-3, weight_decay=1e-2)
    tr_rccl = P.DataParallelTrainer(net2, ds, lr=1e-3, weight_decay=1e-2, force_averager=True, bucket_bytes=256 << 10)
    rs = np.random.RandomState(1)
    losses = []
    for _ in range(3):
        ids = rs.permutation(60)[:24]
        l1, _ = tr_plain.step(ids)
        l2, _ = tr_rccl.step(ids)
        losses.append((float(l1), float(l2)))
    torch.cuda.synchronize()
    st = tr_rccl.averager.stats
    res["trainer"] = {"params_bit_equal": all(torch.equal(p.detach(), q.detach())
                                               for p, q in zip(net.parameters(), net2.parameters())),
                      "losses_equal": all(a == c for a, c in losses), "stats": st,
                      "unused_stay_none": all((p.grad is None) == (q.grad is None)
                                              for p, q in zip(net.parameters(), net2.parameters()))}
    # ---- C: a VISIBLE all-reduce.  Over one rank the real collective is an identity and cannot show a missing stream
    # dependency.  Here dist.all_reduce is replaced by a stub with NCCL's ordering contract -- the collective runs on its
    # own stream behind an event of the caller's stream, Work.wait() makes the caller's stream wait for it without
    # blocking the host -- that first spins for ~1 ms and then DOUBLES a floating-point buffer (the int32 used-bitmap
    # passes through).  Every gradient must come out as exactly 2 x the plain one: a consumer that reads a bucket before
    # waiting, or a producer (the layer's side stream, a late accumulation) that writes into it after the collective
    # started, changes bits.
    comm = torch.cuda.Stream(device=dev)

    class _Work:
        def __init__(self, ev):
            self.ev = ev

        def wait(self):
            torch.cuda.current_stream(dev).wait_event(self.ev)
            return True

    real_all_reduce = dist.all_reduce

    def visible_all_reduce(t, op=None, group=None, async_op=False):
        ev0 = torch.cuda.Event()
        ev0.record(torch.cuda.current_stream(dev))
        with torch.cuda.stream(comm):
            comm.wait_event(ev0)
            if t.is_floating_point():
                torch.cuda._sleep(2_000_000)
                t.mul_(2.0)
            ev1 = torch.cuda.Event()
            ev1.record(comm)
        w = _Work(ev1)
        if async_op:
            return w
        w.wait()
        return None

    dist.all_reduce = visible_all_reduce
    res["visible"] = {}
    try:
        for overlap in (True, False):
            ops.set_overlap_wgrad(overlap)
            plain = layer_grads(None, "none")
            plain2 = layer_grads(None, "none", micro=2)
            for bucket in (64 << 20, 1 << 20):
                avg = GradientAverager(params, bucket_bytes=bucket, force=True)
                for mode in ("none", "views"):
                    for rep in range(3):
                        got = layer_grads(avg, mode)
                    res["visible"][f"overlap={int(overlap)} bucket={bucket >> 20}MB grads={mode}"] = {
                        "exactly_doubled": all(torch.equal(a, 2.0 * c) for a, c in zip(got, plain)),
                        "launched_in_backward": avg.stats["launched_in_backward"]}
                got2 = layer_grads(avg, "views", micro=2)
                res["visible"][f"overlap={int(overlap)} bucket={bucket >> 20}MB no_sync x2"] = {
                    "exactly_doubled": all(torch.equal(a, 2.0 * c) for a, c in zip(got2, plain2))}
                avg.close()
        # the trainer: p <- AdamW(p, 2 g) must equal a plain trainer fed the doubled loss (gradients 2 g, same bits)
        torch.manual_seed(0)
        net3 = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to(dev)
        net4 = copy.deepcopy(net3)
        tr_vis = P.DataParallelTrainer(net3, ds, lr=1e-3, weight_decay=1e-2, force_averager=True, bucket_bytes=256 << 10)
        tr_ref = P.DataParallelTrainer(net4, ds, lr=1e-3, weight_decay=1e-2)
        ref_loss = tr_ref._loss
        tr_ref._loss = lambda ids: (lambda lb: (2.0 * lb[0], lb[1]))(ref_loss(ids))
        rs = np.random.RandomState(2)
        for _ in range(3):
            ids = rs.permutation(60)[:24]
            tr_vis.step(ids)
            tr_ref.step(ids)
        torch.cuda.synchronize()
        res["visible"]["trainer"] = {"params_bit_equal": all(torch.equal(p.detach(), q.detach())
                                                             for p, q in zip(net3.parameters(), net4.parameters()))}
    finally:
        dist.all_reduce = real_all_reduce
        ops.set_overlap_wgrad(None)
    dist.barrier()
    dist.destroy_process_group()
    print("RCCL_RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
