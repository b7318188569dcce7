"""Edge-chunked execution of one GATConvNodes layer (BASELINE configs[4]: 50 000 crystals x 64 neighbours = 64 M edges).

A layer step keeps per-edge pre-activations for its backward (6 KB per edge: 393 GB at 64 M edges), so beyond
`max_edges_per_pass` the layer runs over CLOSED chunks of the graph -- node ranges [n0, n1) that no edge enters or
leaves (crystals never share an edge, reference lightning_module.py:200, so every crystal boundary is such a cut) --
one chunk at a time through the same kernels:

  forward   per chunk, nothing saved but the layer inputs;
  backward  per chunk: the chunk's forward again (now saving), then its backward; input gradients land in their slices,
            parameter gradients are summed over the chunks in chunk order (deterministic).

The workspace is bounded by the chunk size, the result is the unchunked layer's up to the summation order of the
parameter gradients.  Chunks are found from edge_index alone (no `batch` vector is passed to the layer API): a boundary
n is closed when no edge has one endpoint below n and the other at or above it.
"""
import os

import torch

_max_edges_per_pass = int(os.environ.get("CGAT_MAX_EDGES_PER_PASS", str(8 << 20)))


def set_max_edges_per_pass(n):
    """Edges one pass of a GATConvNodes layer may cover before it is split into closed chunks (default 8 M, env
    CGAT_MAX_EDGES_PER_PASS: ~70 GB of per-chunk workspace at C = 128, H = 3; measured 4 M / 8 M / 16 M: 1.24 / 1.01 /
    1.00 s per 64 M-edge step)."""
    global _max_edges_per_pass
    _max_edges_per_pass = int(n)


def max_edges_per_pass():
    return _max_edges_per_pass


class Chunk:
    __slots__ = ("n0", "n1", "e0", "e1", "edge_index")

    def __init__(self, n0, n1, e0, e1, edge_index):
        self.n0, self.n1, self.e0, self.e1, self.edge_index = n0, n1, e0, e1, edge_index


_chunk_cache = {}


def closed_chunks(edge_index, num_nodes, max_edges):
    """Closed node ranges covering the graph, each with at most `max_edges` edges where the graph allows it.
    Requires edge_index[0] ascending (the reference's layout, data.py:116-120: every atom's K edges are consecutive), so
    that a node range owns a contiguous range of edges.  Cached per edge_index storage."""
    key = (edge_index.data_ptr(), tuple(edge_index.shape), int(num_nodes), edge_index._version, int(max_edges))
    hit = _chunk_cache.get(key)
    if hit is not None:
        return hit[0]
    src, dst = edge_index[0], edge_index[1]
    N, E = int(num_nodes), int(edge_index.shape[1])
    if E > 1 and not bool((src[1:] >= src[:-1]).all()):
        raise NotImplementedError("edge-chunked execution needs edge_index[0] in ascending order (the reference's batches "
                                  "are); sort the edges by source or raise cgat_amd.set_max_edges_per_pass")
    lo, hi = torch.minimum(src, dst), torch.maximum(src, dst)
    # crossings[n] = number of edges with lo < n <= hi  (difference array over the boundaries 0 .. N)
    diff = torch.zeros(N + 2, dtype=torch.int32, device=edge_index.device)
    one = torch.ones(E, dtype=torch.int32, device=edge_index.device)
    diff.index_add_(0, lo + 1, one)
    diff.index_add_(0, hi + 1, -one)
    closed = (torch.cumsum(diff[:N + 1], 0) == 0)
    closed[0] = True
    closed[N] = True
    bounds = closed.nonzero().flatten()                                   # node boundaries, ascending, incl. 0 and N
    eptr = torch.searchsorted(src.contiguous(), bounds).cpu().tolist()    # first edge of each boundary's node
    bounds = bounds.cpu().tolist()
    chunks, i = [], 0
    while i < len(bounds) - 1:
        j = i + 1
        while j + 1 < len(bounds) and eptr[j + 1] - eptr[i] <= max_edges:
            j += 1
        n0, n1, e0, e1 = bounds[i], bounds[j], eptr[i], eptr[j]
        ei_c = (edge_index[:, e0:e1] - n0).contiguous()
        ei_c._cgat_plan = None                    # ops.get_plan: this tensor keeps its own CSR plan (built + validated once)
        chunks.append(Chunk(n0, n1, e0, e1, ei_c))
        i = j
    if len(_chunk_cache) >= 4:
        _chunk_cache.pop(next(iter(_chunk_cache)))
    _chunk_cache[key] = (chunks, edge_index)                              # keep the keyed storage alive
    return chunks


class ChunkedLayerFn(torch.autograd.Function):
    """y = run(x, edge_index, edge_attr, x_0) evaluated chunk by chunk with per-chunk recomputation in backward.
    `run(x, ei, e, x0)` is the layer's own single-pass propagate (it reads the layer's parameters, passed here as
    `params` so that autograd routes their gradients)."""

    @staticmethod
    def forward(ctx, run, chunks, x, edge_attr, x_0, *params):
        ctx.run, ctx.chunks = run, chunks
        ctx.save_for_backward(x, edge_attr, x_0, *params)
        y = None
        with torch.no_grad():
            for c in chunks:
                yc = run(x[c.n0:c.n1], c.edge_index, edge_attr[c.e0:c.e1], x_0[c.n0:c.n1])
                if y is None:
                    y = torch.empty(x.shape[0], yc.shape[1], dtype=yc.dtype, device=yc.device)
                y[c.n0:c.n1] = yc
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, edge_attr, x_0, *params = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_x = torch.zeros_like(x) if need[2] else None
        g_e = torch.empty_like(edge_attr) if need[3] else None
        g_x0 = torch.zeros_like(x_0) if need[4] else None
        g_p = [None] * len(params)
        for c in ctx.chunks:
            with torch.enable_grad():
                xs = x[c.n0:c.n1].detach().requires_grad_(True)
                es = edge_attr[c.e0:c.e1].detach().requires_grad_(True)
                x0s = x_0[c.n0:c.n1].detach().requires_grad_(True)
                yc = ctx.run(xs, c.edge_index, es, x0s)
                grads = torch.autograd.grad(yc, [xs, es, x0s] + list(params), g_y[c.n0:c.n1], allow_unused=True)
            if g_x is not None and grads[0] is not None:
                g_x[c.n0:c.n1] = grads[0]
            if g_e is not None:
                if grads[1] is not None:
                    g_e[c.e0:c.e1] = grads[1]
                else:
                    g_e[c.e0:c.e1].zero_()
            if g_x0 is not None and grads[2] is not None:
                g_x0[c.n0:c.n1] = grads[2]
            for k, g in enumerate(grads[3:]):
                if g is not None:
                    g_p[k] = g if g_p[k] is None else g_p[k].add_(g)
            del grads, yc
        return (None, None, g_x, g_e, g_x0, *g_p)


class ChunkedSplitLayerFn(torch.autograd.Function):
    """The same, for a layer whose single pass is aggregate (per-edge attention: 6 KB of saved pre-activations per edge)
    followed by update (the hypernetwork: a few [rows, C] tensors per predicted layer).  Only the AGGREGATE half is
    recomputed in backward; the update half runs once, in forward, with its (small) autograd state kept per chunk --
    ~1.5 GB per 8 M-edge chunk at K = 64 -- so a step costs 2 aggregate passes + 1 update pass instead of 2 + 2 (the
    hypernetwork's contractions were 200 ms of the 830 ms 64 M-edge step, a quarter of them the recomputation).
    `agg(x, ei, e)` -> aggr [n, C]; `upd(aggr, x_0, x)` -> y [n, C]."""

    @staticmethod
    def forward(ctx, agg, upd, chunks, x, edge_attr, x_0, *params):
        ctx.agg, ctx.chunks = agg, chunks
        ctx.save_for_backward(x, edge_attr, x_0, *params)
        ctx.upd_state = []
        y = None
        keep = any(ctx.needs_input_grad)              # inference: nothing to keep, the update runs without a graph too
        for c in chunks:
            with torch.no_grad():
                aggr = agg(x[c.n0:c.n1], c.edge_index, edge_attr[c.e0:c.e1])
                if not keep:
                    yc = upd(aggr, x_0[c.n0:c.n1], x[c.n0:c.n1])
                    if y is None:
                        y = torch.empty(x.shape[0], yc.shape[1], dtype=yc.dtype, device=yc.device)
                    y[c.n0:c.n1] = yc
                    continue
            with torch.enable_grad():
                a = aggr.detach().requires_grad_(True)
                xs = x[c.n0:c.n1].detach().requires_grad_(True)
                x0s = x_0[c.n0:c.n1].detach().requires_grad_(True)
                yc = upd(a, x0s, xs)
            ctx.upd_state.append((a, xs, x0s, yc))
            if y is None:
                y = torch.empty(x.shape[0], yc.shape[1], dtype=yc.dtype, device=yc.device)
            y[c.n0:c.n1] = yc.detach()
        return y

    @staticmethod
    def backward(ctx, g_y):
        x, edge_attr, x_0, *params = ctx.saved_tensors
        need = ctx.needs_input_grad
        g_x = torch.zeros_like(x) if need[3] else None
        g_e = torch.empty_like(edge_attr) if need[4] else None
        g_x0 = torch.zeros_like(x_0) if need[5] else None
        g_p = [None] * len(params)

        def add_params(grads):
            for k, g in enumerate(grads):
                if g is not None:
                    g_p[k] = g if g_p[k] is None else g_p[k].add_(g)
        for ci, c in enumerate(ctx.chunks):
            a, xs_u, x0s_u, yc = ctx.upd_state[ci]
            ctx.upd_state[ci] = None                                     # the chunk's update state is released here
            gu = torch.autograd.grad(yc, [a, xs_u, x0s_u] + list(params), g_y[c.n0:c.n1], allow_unused=True)
            g_aggr = gu[0]
            add_params(gu[3:])
            if g_x is not None and gu[1] is not None:
                g_x[c.n0:c.n1] = gu[1]
            if g_x0 is not None and gu[2] is not None:
                g_x0[c.n0:c.n1] = gu[2]
            del a, xs_u, x0s_u, yc
            if g_aggr is None:                                            # (an update that ignores its aggregate)
                if g_e is not None:
                    g_e[c.e0:c.e1].zero_()
                continue
            with torch.enable_grad():
                xs = x[c.n0:c.n1].detach().requires_grad_(True)
                es = edge_attr[c.e0:c.e1].detach().requires_grad_(True)
                aggr = ctx.agg(xs, c.edge_index, es)
                ga = torch.autograd.grad(aggr, [xs, es] + list(params), g_aggr, allow_unused=True)
            if g_x is not None and ga[0] is not None:
                g_x[c.n0:c.n1] += ga[0]
            if g_e is not None:
                if ga[1] is not None:
                    g_e[c.e0:c.e1] = ga[1]
                else:
                    g_e[c.e0:c.e1].zero_()
            add_params(ga[2:])
            del ga, gu, aggr
        return (None, None, None, g_x, g_e, g_x0, *g_p)
