"""Child process of tests/test_rccl_multi_rank.py: ONE RANK of a world-size-N data-parallel run of the HIP path, one device
per rank (backend from CGAT_TEST_BACKEND: "nccl" = RCCL over xGMI on a multi-GPU box, or "gloo" with every rank on
cuda:0 as the comparison leg).  Each rank runs a DIFFERENT batch through one GATConvNodes layer with the
GradientAverager, reports its mean gradients (raw bytes, for the bit comparison between backends) and then three
DataParallelTrainer steps, after which the replicas must be bit-identical (reference: Lightning strategy='ddp',
CGAT/train.py:53-62).  Started BEFORE anything touches the GPU in this process."""
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    rank = int(os.environ["RANK"])
    backend = os.environ.get("CGAT_TEST_BACKEND", "nccl")
    if backend == "gloo":
        os.environ["CGAT_DIST_SHARE_GPU"] = "1"
    import numpy as np
    import torch
    import torch.distributed as dist
    from cgat_amd.dist import GradientAverager, init_from_env
    _, world, dev = init_from_env(backend)
    assert dev.type == "cuda"
    import cgat_amd as P
    from cgat_amd.graph import synthetic_dataset_dict

    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)      # identical parameters on every rank
    params = list(layer.parameters())
    b, _ = P.synthetic_batch(40 + 8 * rank, 20, 12, seed=10 + rank)      # ragged across ranks
    g = torch.Generator().manual_seed(20 + rank)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    avg = GradientAverager(params, bucket_bytes=1 << 20)
    for rep in range(3):
        avg.zero_grad()
        layer(x, ei, e, x0).backward(cot)
        avg.finish()
    torch.cuda.synchronize()
    flat = torch.cat([p.grad.detach().reshape(-1) for p in params]).cpu().numpy()
    res = {"rank": rank, "world_size": dist.get_world_size(), "backend": dist.get_backend(), "device": str(dev),
           "devices_visible": torch.cuda.device_count(),
           "mean_grad_sha256": hashlib.sha256(flat.tobytes()).hexdigest(), "mean_grad_absmax": float(np.abs(flat).max()),
           "launched_in_backward": avg.stats["launched_in_backward"]}
    # every rank holds the SAME mean: compare byte-wise across ranks
    mine = [None] * world
    dist.all_gather_object(mine, res["mean_grad_sha256"])
    res["mean_identical_across_ranks"] = len(set(mine)) == 1
    avg.close()

    data, emb = synthetic_dataset_dict(80, (2, 40), 24, seed=5)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device=dev)
    torch.manual_seed(0)
    net = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to(dev)
    tr = P.DataParallelTrainer(net, ds, lr=1e-3, weight_decay=1e-2, rank=rank, world=world, bucket_bytes=256 << 10)
    rs = np.random.RandomState(3)
    losses = []
    for _ in range(3):
        ids = rs.permutation(80)[:8 * world]
        loss, _ = tr.step(tr.local_ids(ids))
        losses.append(float(loss))
    torch.cuda.synchronize()
    pflat = torch.cat([p.detach().reshape(-1) for p in net.parameters()]).cpu().numpy()
    hashes = [None] * world
    dist.all_gather_object(hashes, hashlib.sha256(pflat.tobytes()).hexdigest())
    res["replicas_bit_identical"] = len(set(hashes)) == 1
    res["params_sha256"] = hashes[rank]
    res["losses"] = losses
    dist.barrier()
    dist.destroy_process_group()
    print("RCCLN_RESULT " + json.dumps(res), flush=True)


if __name__ == "__main__":
    main()
