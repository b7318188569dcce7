"""Child process of tests/test_dist_two_ranks_gpu.py: ONE RANK of a world-size-2 data-parallel run of the HIP path with
both ranks on the box's single GPU (backend gloo, CGAT_DIST_SHARE_GPU=1: the collectives go through host memory, the
layer, the hooks, the bucket views and the optimiser are the product's).  What a one-rank run cannot show: two ranks with
DIFFERENT batches must end with the mean gradient (bit-equal to (g0 + g1) * 0.5 of the two plain runs) and with
bit-identical replicas after optimiser steps; the collectives are issued in the same order on both ranks whatever order
their autograd engines ran in (reference: Lightning strategy='ddp', CGAT/train.py:53-62)."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(os.environ["RANK"])
    os.environ["CGAT_DIST_SHARE_GPU"] = "1"
    import numpy as np
    import torch
    import torch.distributed as dist
    from cgat_amd.dist import GradientAverager, init_from_env
    _, world, dev = init_from_env("gloo")
    assert world == 2 and dev.type == "cuda"
    import cgat_amd as P
    from cgat_amd.graph import synthetic_dataset_dict

    # ---- A: one layer at the benchmark widths, a different batch per rank ----
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)      # identical parameters on both ranks
    params = list(layer.parameters())

    def batch(r):
        b, _ = P.synthetic_batch(48 + 16 * r, 20, 12, seed=10 + r)       # ragged across ranks: 48 and 64 crystals
        g = torch.Generator().manual_seed(20 + r)
        N, E = b.num_nodes, b.edge_index.shape[1]
        x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
        return x, b.edge_index.to(dev), e, x0, cot

    def grads(r, avg):
        x, ei, e, x0, cot = batch(r)
        if avg is not None:
            avg.zero_grad()
        else:
            for p in params:
                p.grad = None
        layer(x, ei, e, x0).backward(cot)
        if avg is not None:
            avg.finish()
        torch.cuda.synchronize()
        return [p.grad.detach().clone() for p in params]

    g0, g1 = grads(0, None), grads(1, None)                              # both plain runs, locally
    want = [(a + b) * 0.5 for a, b in zip(g0, g1)]
    res = {"rank": rank, "layer": {}}
    for bucket in (64 << 20, 1 << 20):
        avg = GradientAverager(params, bucket_bytes=bucket)
        for rep in range(3):
            got = grads(rank, avg)
        res["layer"][f"bucket={bucket >> 20}MB"] = {
            "mean_bit_equal": all(torch.equal(a, b) for a, b in zip(got, want)),
            "launched_in_backward": avg.stats["launched_in_backward"], "buckets": len(avg.buckets)}
        avg.close()

    # ---- B: three training steps, each rank its share of a global batch; replicas must stay bit-identical ----
    data, emb = synthetic_dataset_dict(80, (2, 40), 24, seed=5)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device=dev)
    torch.manual_seed(0)
    net = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to(dev)
    tr = P.DataParallelTrainer(net, ds, lr=1e-3, weight_decay=1e-2, rank=rank, world=world, bucket_bytes=256 << 10)
    rs = np.random.RandomState(3)
    losses = []
    for _ in range(3):
        ids = rs.permutation(80)[:32]
        loss, _ = tr.step(tr.local_ids(ids))
        losses.append(float(loss))
    torch.cuda.synchronize()
    flat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu()
    both = [None, None]
    dist.all_gather_object(both, flat.numpy().tobytes())
    res["trainer"] = {"replicas_bit_identical": both[0] == both[1], "losses": losses,
                      "unused_none": sum(p.grad is None for p in net.parameters()),
                      "stats": tr.averager.stats}
    dist.barrier()
    dist.destroy_process_group()
    print("DIST2_RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
