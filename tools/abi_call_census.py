"""Dev tool: which C-ABI entry points one training-shaped step calls, how often and with which sizes
(bench shapes: --graphs crystals x 20 atoms x 12 neighbours, the 4-layer network fwd+bwd).  Wraps the ctypes functions."""
import collections, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import cgat_amd as P
from cgat_amd import _lib, ops
graphs = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = "cuda:0"
b, roost = P.synthetic_batch(graphs, 20, 12, seed=4)
b = b.to(dev); roost = tuple(t.to(dev) for t in roost)
torch.manual_seed(1)
net = P.CGAtNet(200, 128, 4, msg_heads=3, neighbor_number=12, update_edges=True).to(dev)
params = list(net.parameters())
def step():
    for p in params: p.grad = None
    out = net(b, roost)
    (out[:, 0] - b.y).abs().mean().backward()
step(); torch.cuda.synchronize()
calls = collections.Counter()
launches = collections.Counter()
lib = _lib.lib
names = [n for n in dir(lib) if n.startswith("cgat_")]
orig = {}
for n in names:
    f = getattr(lib, n)
    if "workspace" in n or "saved" in n or n in ("cgat_last_error", "cgat_abi_version", "cgat_prof_launches", "cgat_prof_get",
                                                  "cgat_prof_reset", "cgat_prof_enable", "cgat_get_bilinear_mode",
                                                  "cgat_get_edge_storage", "cgat_mt_chunk_elems"):
        continue
    orig[n] = f
    def mk(n, f):
        def w(*a):
            ints = tuple(int(x) for x in a if isinstance(x, int) and not isinstance(x, bool) and 0 < x < (1 << 24))
            n0 = ops.prof_launches()
            r = f(*a)
            launches[n] += ops.prof_launches() - n0
            calls[(n, ints[:6])] += 1
            return r
        return w
    setattr(lib, n, mk(n, f))
step(); torch.cuda.synchronize()
for n, f in orig.items():
    setattr(lib, n, f)
tot = collections.Counter()
for (n, ints), c in calls.items():
    tot[n] += c
print("entry point: calls per step, library kernel launches per step")
for n, c in tot.most_common():
    print(f"  {n:45s} {c:4d} {launches[n]:5d}")
print("by sizes (first small integer arguments):")
for (n, ints), c in sorted(calls.items(), key=lambda kv: -kv[1])[:40]:
    print(f"  {c:4d} x {n} {ints}")
