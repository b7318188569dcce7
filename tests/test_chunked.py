"""Edge-chunked layer execution (BASELINE configs[4] / config 5: 64 neighbours per atom, 64 M edges): chunk discovery
and the chunked autograd function on CPU (host logic, the oracle as the layer); on the GPU the K = 64 layer against the
oracle, chunked == unchunked at 1 M edges, and crystal locality at 8 M edges."""
import os
import sys

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def test_closed_chunks_partition_the_graph():
    from cgat_amd import chunked
    from cgat_amd.graph import synthetic_batch
    b, _ = synthetic_batch(10, 6, 4, seed=1)
    ch = chunked.closed_chunks(b.edge_index, b.num_nodes, 60)
    assert ch[0].n0 == 0 and ch[-1].n1 == b.num_nodes and ch[0].e0 == 0 and ch[-1].e1 == b.edge_index.shape[1]
    for a, c in zip(ch, ch[1:]):
        assert a.n1 == c.n0 and a.e1 == c.e0
    for c in ch:
        assert c.e1 - c.e0 <= 60 and c.n0 % 6 == 0                       # whole crystals only
        ei = b.edge_index[:, c.e0:c.e1]
        assert int(ei.min()) >= c.n0 and int(ei.max()) < c.n1            # closed: no edge leaves the node range
        assert torch.equal(c.edge_index, ei - c.n0)
    # an edge between two "crystals" fuses them into one closed component (larger than the budget: kept whole)
    ei = torch.tensor([[0, 0, 1, 2, 3, 3, 4, 5], [1, 3, 0, 3, 2, 2, 5, 4]])
    ch = chunked.closed_chunks(ei, 6, 2)
    assert [(c.n0, c.n1, c.e0, c.e1) for c in ch] == [(0, 4, 0, 6), (4, 6, 6, 8)]
    with pytest.raises(NotImplementedError):
        chunked.closed_chunks(torch.tensor([[1, 0], [0, 1]]), 2, 1)      # sources not ascending


def test_chunked_function_equals_single_pass_on_cpu():
    """ChunkedLayerFn with the oracle layer as `run`: outputs and every gradient equal the single pass."""
    from cgat_amd import chunked
    from cgat_amd.graph import synthetic_batch
    from oracle import cgat_oracle as O
    G, A, K, C = 6, 5, 7, 16
    b, _ = synthetic_batch(G, A, K, seed=2)
    g = torch.Generator().manual_seed(3)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0 = (torch.randn(n, C, generator=g).double() for n in (N, E, N))
    torch.manual_seed(1)
    layer = O.GATConvNodes(C, C, C, 3, concat=True).double()
    params = list(layer.parameters())
    cot = torch.randn(N, C, generator=g).double()
    ins = [t.clone().requires_grad_(True) for t in (x, e, x0)]
    y1 = layer(ins[0], b.edge_index, ins[1], ins[2])
    g1 = torch.autograd.grad((y1 * cot).sum(), ins + params, allow_unused=True)
    chunks = chunked.closed_chunks(b.edge_index, N, 2 * A * K)
    assert len(chunks) == 3
    ins2 = [t.clone().requires_grad_(True) for t in (x, e, x0)]
    run = lambda xs, ei, es, x0s: layer(xs, ei, es, x0s)
    y2 = chunked.ChunkedLayerFn.apply(run, chunks, ins2[0], ins2[1], ins2[2], *params)
    g2 = torch.autograd.grad((y2 * cot).sum(), ins2 + params, allow_unused=True)
    assert float((y1 - y2).abs().max()) <= 1e-12
    for a, c in zip(g1, g2):
        assert (a is None) == (c is None)
        if a is not None:
            assert float((a - c).abs().max()) <= 1e-10 * max(1.0, float(a.abs().max()))


@pytest.mark.gpu
def test_config5_k64_layer_vs_oracle_chunked():
    """Config 5 shape at a size the oracle finishes in seconds: 64 neighbours per atom (in-degree ~64: segments far
    longer than at K = 12), C = Ce = 128, H = 3, run in 3 closed chunks, against the oracle (fp32 tolerance 1e-4)."""
    import cgat_amd as P
    from cgat_amd import chunked
    from oracle import cgat_oracle as O
    from test_hip_golden import _compare_with_oracle
    b, _ = P.synthetic_batch(6, 20, 64, seed=41)
    g = torch.Generator().manual_seed(42)
    N, E = b.num_nodes, b.edge_index.shape[1]
    inputs = {"x": torch.randn(N, 128, generator=g), "edge_index": b.edge_index,
              "edge_attr": torch.randn(E, 128, generator=g), "x_0": torch.randn(N, 128, generator=g)}
    call = lambda m, i: m(i["x"], i["edge_index"], i["edge_attr"], i["x_0"])
    old = chunked.max_edges_per_pass()
    chunked.set_max_edges_per_pass(2 * 20 * 64)
    try:
        _compare_with_oracle(lambda: P.GATConvNodes(128, 128, 128, 3, concat=True),
                             lambda: O.GATConvNodes(128, 128, 128, 3, concat=True), inputs, call)
    finally:
        chunked.set_max_edges_per_pass(old)


@pytest.mark.gpu
def test_config5_chunked_equals_unchunked_at_1m_edges():
    """814 crystals x 20 atoms x 64 neighbours = 1 041 920 edges: the layer in closed chunks of <= 262 144 edges against
    the single pass -- outputs and all gradients to 1e-5 (only the summation order of the parameter gradients and the
    tile boundaries differ)."""
    import cgat_amd as P
    from cgat_amd import chunked
    dev = "cuda:0"
    b, _ = P.synthetic_batch(814, 20, 64, seed=43)
    g = torch.Generator().manual_seed(44)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)
    params = list(layer.parameters())

    def run():
        ins = [t.clone().requires_grad_(True) for t in (x, e, x0)]
        y = layer(ins[0], ei, ins[1], ins[2])
        return y.detach(), torch.autograd.grad((y * cot).sum(), ins + params)
    old = chunked.max_edges_per_pass()
    try:
        chunked.set_max_edges_per_pass(1 << 30)
        y1, g1 = run()
        chunked.set_max_edges_per_pass(262144)
        assert len(chunked.closed_chunks(ei, N, 262144)) >= 4
        y2, g2 = run()
    finally:
        chunked.set_max_edges_per_pass(old)
    assert float((y1 - y2).abs().max()) <= 1e-5 * float(y1.abs().max())
    scale = max(float(a.abs().max()) for a in g1)
    for k, (a, c) in enumerate(zip(g1, g2)):
        # 1e-5 of the tensor, or fp32 resolution of the layer's largest gradient (MH_A.fc_out.bias is zero by softmax
        # shift invariance: its value is rounding noise at that scale in any summation order)
        assert float((a - c).abs().max()) <= max(1e-5 * float(a.abs().max()), 1e-6 * scale), k


@pytest.mark.gpu
def test_config5_locality_at_8m_edges():
    """6250 crystals x 20 atoms x 64 neighbours = 8 M edges in chunks of 2 M: perturbing one crystal's inputs changes
    that crystal's outputs and input gradients and nobody else's (crystals never share an edge; a chunking bug --
    a chunk that cuts a crystal, an off-by-one slice -- would leak)."""
    import cgat_amd as P
    from cgat_amd import chunked
    dev = "cuda:0"
    G, A, K = 6250, 20, 64
    b, _ = P.synthetic_batch(G, A, K, seed=45)
    g = torch.Generator(device=dev).manual_seed(46)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g, device=dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)

    def run(xx, ee):
        xx, ee = xx.clone().requires_grad_(True), ee.clone().requires_grad_(True)
        y = layer(xx, ei, ee, x0)
        gx, ge = torch.autograd.grad((y * cot).sum(), [xx, ee])
        return y.detach(), gx, ge
    old = chunked.max_edges_per_pass()
    chunked.set_max_edges_per_pass(2 << 20)
    try:
        y1, gx1, ge1 = run(x, e)
        c = 3333                                                          # a crystal in the middle of a chunk
        x2, e2 = x.clone(), e.clone()
        x2[c * A:(c + 1) * A] += 1.0
        e2[c * A * K:(c + 1) * A * K] *= 1.5
        y2, gx2, ge2 = run(x2, e2)
    finally:
        chunked.set_max_edges_per_pass(old)
    assert torch.isfinite(y1).all() and torch.isfinite(gx1).all() and torch.isfinite(ge1).all()
    # "unchanged" to 1e-5 of the tensor's max-norm, not bitwise: the fp16-split kernels scale some operands by one
    # power of two per tensor, which the perturbed crystal may move for its whole chunk
    def same(a, c):
        return float((a - c).abs().max()) <= 1e-5 * float(a.abs().max())
    mask = torch.ones(N, dtype=torch.bool, device=dev)
    mask[c * A:(c + 1) * A] = False
    assert same(y1[mask], y2[mask]) and same(gx1[mask], gx2[mask])
    assert not same(y1[~mask], y2[~mask])
    emask = torch.ones(E, dtype=torch.bool, device=dev)
    emask[c * A * K:(c + 1) * A * K] = False
    assert same(ge1[emask], ge2[emask])
    # the other chunks do not see the perturbed crystal at all: bit-identical there
    far = torch.zeros(N, dtype=torch.bool, device=dev)
    far[:1000 * A] = True
    assert torch.equal(y1[far], y2[far]) and torch.equal(gx1[far], gx2[far])


@pytest.mark.gpu
def test_config5_bf16_edge_storage_vs_oracle():
    """The "bf16 activations" of BASELINE configs[4]: Z / gZ of the per-edge phase stored as bf16 (logits, softmax
    statistics, sums and products unchanged).  K = 64 layer on 6 crystals against the oracle's fp64 run: stated
    tolerance of the mode 1e-2 max-norm relative for outputs and all gradients (fp32 storage: 1e-4); and the mode is
    really on (results differ from the fp32-storage run by more than 1e-6)."""
    import cgat_amd as P
    from oracle import cgat_oracle as O
    dev = "cuda:0"
    b, _ = P.synthetic_batch(6, 20, 64, seed=47)
    g = torch.Generator().manual_seed(48)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g) for n in (N, E, N, N))
    torch.manual_seed(1)
    om = O.GATConvNodes(128, 128, 128, 3, concat=True).double()
    pm = P.GATConvNodes(128, 128, 128, 3, concat=True)
    pm.load_state_dict({k: v.float() for k, v in om.state_dict().items()})
    pm = pm.to(dev)
    xo, eo = x.double().requires_grad_(True), e.double().requires_grad_(True)
    yo = om(xo, b.edge_index, eo, x0.double())
    go = torch.autograd.grad((yo * cot.double()).sum(), [xo, eo] + list(om.parameters()))

    def run():
        xp, ep = x.to(dev).requires_grad_(True), e.to(dev).requires_grad_(True)
        y = pm(xp, b.edge_index.to(dev), ep, x0.to(dev))
        return y.detach(), torch.autograd.grad((y * cot.to(dev)).sum(), [xp, ep] + list(pm.parameters()))
    assert P.get_edge_storage() == "f32"
    y32, g32 = run()
    P.set_edge_storage("bf16")
    try:
        if P.get_bilinear_mode() == "f32":              # the diagnostic f32-MFMA mode has no bf16 form: it must refuse
            with pytest.raises(RuntimeError, match="bf16"):
                run()
            return
        y16, g16 = run()
    finally:
        P.set_edge_storage("f32")
    rel = lambda a, r: float((a.double().cpu() - r).abs().max() / r.abs().max())
    assert rel(y16, yo.detach()) <= 1e-2
    scale = max(float(r.abs().max()) for r in go)
    worst = 0.0
    for a, r in zip(g16, go):
        err = float((a.double().cpu() - r).abs().max())
        worst = max(worst, err / max(float(r.abs().max()), 1e-3 * scale))
    assert worst <= 1e-2, worst
    assert rel(y16, y32.double().cpu()) > 1e-6          # the mode is on, in EVERY split arithmetic mode (round 5)
    assert rel(y32, yo.detach()) <= 1e-4
    # "bf16-mma" (round 6, the 24-bit modes): bf16 OPERANDS in the two per-edge backward products as well (one matrix pass,
    # fp32 accumulation) -- the same stated tolerance, the same forward, gradients that differ from the storage-only
    # mode's (the mode is on)
    if P.get_bilinear_mode() in ("f16x3c", "bf16x6"):
        P.set_edge_storage("bf16-mma")
        try:
            ym, gm = run()
        finally:
            P.set_edge_storage("f32")
        assert torch.equal(ym, y16)
        worst_m, moved = 0.0, 0.0
        for a, a16, r in zip(gm, g16, go):
            den = max(float(r.abs().max()), 1e-3 * scale)
            worst_m = max(worst_m, float((a.double().cpu() - r).abs().max()) / den)
            moved = max(moved, float((a.double().cpu() - a16.double().cpu()).abs().max()) / den)
        assert worst_m <= 1e-2, worst_m
        assert moved > 1e-5, moved


@pytest.mark.gpu
def test_bf16_edge_storage_is_never_silently_ignored():
    """Round 4's library ignored the storage switch outside the f16x3 mode and ran fp32 storage under the bf16 label.  Now
    the bf16 form exists in every split mode, and a scalar-attention layer WITHOUT a bf16 form -- any layer in the f32
    arithmetic mode, or one at other widths -- raises instead (CGAT_ERR_UNSUPPORTED from cgat_nodes_attention_forward)."""
    import cgat_amd as P
    dev = "cuda:0"
    b, _ = P.synthetic_batch(4, 10, 6, seed=5)
    g = torch.Generator().manual_seed(6)
    N, E = b.num_nodes, b.edge_index.shape[1]
    ei = b.edge_index.to(dev)

    def layer_out(C, backward=False):
        torch.manual_seed(3)
        layer = P.GATConvNodes(C, C, C, 3, concat=True).to(dev)
        x, e, x0 = (torch.randn(n, C, generator=g).to(dev) for n in (N, E, N))
        if backward:                                    # the backward has its own refusals: run it too
            x.requires_grad_(True); e.requires_grad_(True)
            y = layer(x, ei, e, x0)
            y.square().sum().backward()
            assert torch.isfinite(x.grad).all() and torch.isfinite(e.grad).all()
            return y.detach()
        return layer(x, ei, e, x0)

    def raw_attention(Hd):
        """The fused scalar-attention op through the C ABI with a hidden width the module never builds (Hd != 256)."""
        from cgat_amd import ops
        H, C = 3, 128
        D = 3 * C
        plan = ops.get_plan(ei, N)
        x, e = torch.randn(N, C, generator=g).to(dev).requires_grad_(True), torch.randn(E, C, generator=g).to(dev)
        ws = [torch.randn(H * Hd, D, generator=g).to(dev) * 0.05, torch.zeros(H * Hd, device=dev),
              torch.randn(H, Hd, generator=g).to(dev) * 0.05, torch.zeros(H, device=dev),
              torch.randn(H * Hd, D, generator=g).to(dev) * 0.05, torch.zeros(H * Hd, device=dev),
              torch.randn(H * C, Hd, generator=g).to(dev) * 0.05, torch.zeros(H * C, device=dev)]
        y = ops.NodesAttentionFn.apply(x, e, plan, H, *ws)
        y.sum().backward()
        return y
    mode0 = P.get_bilinear_mode()
    P.set_edge_storage("bf16")
    try:
        for mode in ("f16x3c", "bf16x6", "f16x3"):
            P.set_bilinear_mode(mode)
            assert torch.isfinite(layer_out(128, backward=True)).all()
        for mode in ("f16x3c", "bf16x6"):
            # Hd = 128 / 384: the 24-bit modes' bf16 backward exists at Hd == 256 only -- the FORWARD must refuse
            P.set_bilinear_mode(mode)
            for Hd in (128, 384):
                with pytest.raises(RuntimeError, match="bf16"):
                    raw_attention(Hd)
        P.set_bilinear_mode("f32")
        with pytest.raises(RuntimeError, match="bf16"):
            layer_out(128)
        P.set_bilinear_mode(mode0)
        with pytest.raises(RuntimeError, match="bf16"):
            layer_out(64)                               # no bf16 form at this width
    finally:
        P.set_edge_storage("f32")
        P.set_bilinear_mode(mode0)
    assert torch.isfinite(layer_out(64)).all()


@pytest.mark.gpu
def test_bf16_edge_storage_at_1m_edges_matches_fp32_storage():
    """E = 1 000 080 (the benchmark batch): outputs and gradients in the bf16 edge-storage mode stay within 2e-2 of the
    fp32-storage run, per tensor in max-norm."""
    import cgat_amd as P
    dev = "cuda:0"
    b, _ = P.synthetic_batch(4167, 20, 12, seed=0)
    g = torch.Generator().manual_seed(50)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)

    def run():
        xx, ee = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
        y = layer(xx, ei, ee, x0)
        return [y.detach()] + list(torch.autograd.grad((y * cot).sum(), [xx, ee] + list(layer.parameters())))
    a = run()
    P.set_edge_storage("bf16")
    try:
        if P.get_bilinear_mode() == "f32":              # (no bf16 form in the diagnostic f32-MFMA mode: refused)
            with pytest.raises(RuntimeError, match="bf16"):
                run()
            return
        c = run()
    finally:
        P.set_edge_storage("f32")
    # per tensor, relative to its own largest entry (gradients: or 1 % of the layer's largest gradient, whichever is
    # larger -- `damping`'s single-element gradient is a cancellation-dominated sum over all atoms and moves by 1.7 %)
    scale = max(float(t.abs().max()) for t in a[1:])
    for k, (u, v) in enumerate(zip(a, c)):
        den = float(u.abs().max()) if k == 0 else max(float(u.abs().max()), 1e-2 * scale)
        assert float((u - v).abs().max()) <= 2e-2 * den, (k, float((u - v).abs().max()), den)
    # ... and the two runs are NOT the same computation (in round 4's default mode this test compared a run with itself)
    assert float((a[0] - c[0]).abs().max()) > 1e-6 * float(a[0].abs().max())


@pytest.mark.gpu
@pytest.mark.parametrize("K", [12, 64])
def test_rebuilt_gz_equals_stored_gz(K):
    """The backward at the benchmark widths never stores the pre-activation gradient gZ: its consumers rebuild it from one
    sign bit per element and per-node rows.  Against the stored-gZ path of round 1 (edge storage mode
    "f32+gz"): the gradient wrt edge_attr -- whose kernel rebuilds each value with the operations that produced the
    stored one -- agrees to 2e-6 of its largest entry (fp32 rounding: the two paths run different code for the logit
    gradients g_a the values are built from), everything else (the source-side sum also runs in a different summation
    order) to 1e-5."""
    import cgat_amd as P
    dev = "cuda:0"
    b, _ = P.synthetic_batch(500, 20, K, seed=3)
    g = torch.Generator().manual_seed(51)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x, e, x0, cot = (torch.randn(n, 128, generator=g).to(dev) for n in (N, E, N, N))
    ei = b.edge_index.to(dev)
    torch.manual_seed(2)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)

    def run():
        xx, ee = x.clone().requires_grad_(True), e.clone().requires_grad_(True)
        y = layer(xx, ei, ee, x0)
        return [y.detach()] + list(torch.autograd.grad((y * cot).sum(), [xx, ee] + list(layer.parameters())))
    a = run()
    P.set_edge_storage("f32+gz")
    try:
        c = run()
    finally:
        P.set_edge_storage("f32")
    assert torch.equal(a[0], c[0])                      # forward is the same code
    assert float((a[2] - c[2]).abs().max()) <= 2e-6 * float(a[2].abs().max())
    scale = max(float(t.abs().max()) for t in a[1:])    # (a gradient that is zero by symmetry -- the logit bias under
    for k, (u, v) in enumerate(zip(a, c)):              # the softmax -- is rounding noise of the layer's scale)
        assert float((u - v).abs().max()) <= 1e-5 * max(float(u.abs().max()), 1e-2 * scale), k   # floor: 1e-7 of the scale


@pytest.mark.gpu
def test_config5_full_size_chunked_step():
    """BASELINE configs[4] at FULL size: 50 000 crystals x 20 atoms x 64 neighbours = 64 000 000 edges through one
    GATConvNodes layer, forward + backward, in closed chunks (<= 8 M edges each) with the config's bf16 edge storage and
    bf16 per-edge backward operands ("bf16-mma").  No oracle can run this size; the properties checked: everything finite;
    the LAST 3 crystals -- the tail of the last chunk -- give the same output rows (1e-3) and, to the mode's stated
    tolerance (1e-2), the same input gradients as the 3 crystals evaluated alone; the layer really ran in more than one chunk."""
    import cgat_amd as P
    from cgat_amd import chunked, ops
    if P.get_bilinear_mode() not in ("f16x3c", "bf16x6"):
        pytest.skip("bf16-mma exists in the 24-bit modes")
    dev = "cuda:0"
    G, A, K, GS = 50000, 20, 64, 3
    b, _ = P.synthetic_batch(G, A, K, seed=11)
    N, E = b.num_nodes, b.edge_index.shape[1]
    assert E == 64000000
    ei = b.edge_index.to(dev)
    gen = torch.Generator(device=dev).manual_seed(12)
    x, e, x0 = (torch.randn(n, 128, generator=gen, device=dev) for n in (N, E, N))
    torch.manual_seed(1)
    layer = P.GATConvNodes(128, 128, 128, 3, concat=True).to(dev)
    n0, e0 = (G - GS) * A, (G - GS) * A * K
    cot = torch.randn(GS * A, 128, generator=gen, device=dev)
    validate = ops._validate_indices
    ops.set_validate_indices(False)              # (two host round trips per chunk plan; the indices are ours)
    P.set_edge_storage("bf16-mma")
    try:
        assert len(chunked.closed_chunks(ei, N, chunked.max_edges_per_pass())) >= 8
        x.requires_grad_(True); e.requires_grad_(True)
        y = layer(x, ei, e, x0)
        gx, ge = torch.autograd.grad((y[n0:] * cot).sum(), [x, e])
        assert torch.isfinite(y).all() and torch.isfinite(gx).all() and torch.isfinite(ge[e0:]).all()
        assert float(gx[:n0].abs().max()) == 0.0 and float(ge[:e0].abs().max()) == 0.0     # nothing leaks out of the 3 crystals
        xs, es = x[n0:].detach().clone().requires_grad_(True), e[e0:].detach().clone().requires_grad_(True)
        ys = layer(xs, (ei[:, e0:] - n0).contiguous(), es, x0[n0:].contiguous())
        gxs, ges = torch.autograd.grad((ys * cot).sum(), [xs, es])
    finally:
        P.set_edge_storage("f32")
        ops.set_validate_indices(validate)
    rel = lambda a, r: float((a - r).abs().max() / r.abs().max())
    # (bf16 storage: a pre-activation within an fp32 ulp of a bf16 rounding boundary may round the other way in the
    # other kernels the 60-row batch runs on -- the mode's own tolerance class, not 1e-5)
    assert rel(y[n0:].detach(), ys.detach()) <= 1e-3
    assert rel(gx[n0:], gxs) <= 1e-2 and rel(ge[e0:], ges) <= 1e-2
