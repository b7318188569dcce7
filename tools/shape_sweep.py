"""Layer step (GATConvNodes forward + backward, 1M edges, C = 128) across the constructor's other axes: number of heads,
vector attention, first layer (H_Net_0), final layer (no hypernetwork), neighbours per atom.  Looks for cliffs off the
benchmark configuration.  GPU only."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import cgat_amd as P  # noqa: E402


def run(name, C=128, heads=3, graphs=4167, nbrs=12, steps=5, warm=2, **kw):
    dev = torch.device("cuda:0")
    torch.manual_seed(1)
    layer = P.GATConvNodes(C, C, C, heads, concat=True, **kw).to(dev)
    b, _ = P.synthetic_batch(graphs, 20, nbrs, seed=0)
    N, E = b.num_nodes, b.edge_index.shape[1]
    g = torch.Generator().manual_seed(5)
    x, e, x0, cot = (torch.randn(n, C, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    x.requires_grad_(True); e.requires_grad_(True); x0.requires_grad_(True)

    def step():
        for p in layer.parameters():
            p.grad = None
        x.grad = e.grad = x0.grad = None
        y = layer(x, ei, e, x0)
        y.backward(torch.ones_like(y) if y.shape != cot.shape else cot)
    for _ in range(warm):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    return {"case": name, "heads": heads, "N": N, "E": E, "ms_per_step": round(ms, 2), "edges_per_s": round(E / ms * 1e3),
            "ms_per_head": round(ms / heads, 2)}


if __name__ == "__main__":
    cases = [("bench config (H=3)", {}), ("H=1", {"heads": 1}), ("H=2", {"heads": 2}), ("H=4", {"heads": 4}),
             ("H=5", {"heads": 5}), ("H=8", {"heads": 8}), ("first layer (H_Net_0)", {"first": True}),
             ("final layer", {"final": True}), ("vector attention H=3", {"vector_attention": True}),
             ("vector attention H=5", {"vector_attention": True, "heads": 5}),
             ("24 neighbours, 2084 crystals", {"nbrs": 24, "graphs": 2084}),
             ("6 neighbours, 8334 crystals", {"nbrs": 6, "graphs": 8334})]
    out = []
    for name, kw in cases:
        try:
            r = run(name, **kw)
        except Exception as ex:  # noqa: BLE001 -- a sweep: report and go on
            r = {"case": name, "error": repr(ex)[:300]}
        print(json.dumps(r), flush=True)
        out.append(r)
        torch.cuda.empty_cache()
    os.makedirs("gpurun_out", exist_ok=True)
    with open("gpurun_out/r03_shape_sweep.json", "w") as f:
        json.dump(out, f, indent=1)
