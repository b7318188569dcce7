"""Layer step (GATConvNodes forward + backward, 1M edges) at feature widths other than the tuned C = Ce = 128:
what a user of --atom-fea-len 64 / 256 gets from the generic kernels.  GPU only."""
import json
import sys
import time

import torch

sys.path.insert(0, __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
import cgat_amd as P  # noqa: E402


def run(C, graphs=4167, steps=5, warm=2, heads=3):
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    layer = P.GATConvNodes(C, C, C, heads, concat=True).to(dev)
    b, _ = P.synthetic_batch(graphs, 20, 12, seed=0)
    N, E = b.num_nodes, b.edge_index.shape[1]
    g = torch.Generator().manual_seed(5)
    x, e, x0, cot = (torch.randn(n, C, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    x.requires_grad_(True); e.requires_grad_(True); x0.requires_grad_(True)

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = e.grad = x0.grad = None
        layer(x, ei, e, x0).backward(cot)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    Hd = int(3 * C / 1.5)
    # multiply-adds that scale with the width: per-edge first layer (3C x 2 H Hd, x3 for fwd + two backward products)
    # and the hypernetwork contractions (4 layers x 3 x N C^3)
    flop = 2.0 * (3 * E * 3 * C * 2 * heads * Hd + 12 * N * C ** 3)
    return {"C": C, "hidden": Hd, "N": N, "E": E, "ms_per_step": round(ms, 2), "edges_per_s": round(E / ms * 1e3),
            "algorithmic_TFLOPs": round(flop / ms / 1e9, 1), "peak_mem_GB": round(torch.cuda.max_memory_allocated() / 2**30, 1)}


if __name__ == "__main__":
    out = []
    for C in [int(a) for a in sys.argv[1:]] or [128, 64, 256, 96]:
        torch.cuda.reset_peak_memory_stats()
        r = run(C)
        print(json.dumps(r), flush=True)
        out.append(r)
        torch.cuda.empty_cache()
    with open("gpurun_out/r03_width_sweep.json", "w") as f:
        json.dump(out, f, indent=1)
