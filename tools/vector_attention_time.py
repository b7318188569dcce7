"""Dev tool: fwd+bwd time of GATConvNodes with vector attention vs scalar attention at the benchmark shape."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
dev = "cuda:0"
G = int(sys.argv[1]) if len(sys.argv) > 1 else 4167
b, _ = P.synthetic_batch(G, 20, 12, seed=0)
g = torch.Generator().manual_seed(1)
N, E = b.num_nodes, b.edge_index.shape[1]
x, e, x0, cot = (torch.randn(s, 128, generator=g).to(dev) for s in (N, E, N, N))
ei = b.edge_index.to(dev)
for va in (False, True):
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True, vector_attention=va).to(dev)
    xs, es = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
    def step():
        for p in layer.parameters(): p.grad = None
        xs.grad = es.grad = None
        layer(xs, ei, es, x0).backward(cot)
    step(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(3): step()
    torch.cuda.synchronize()
    print(f"vector_attention={va}: {(time.perf_counter() - t0) / 3 * 1e3:.1f} ms/step, peak mem {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
