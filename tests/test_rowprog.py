"""Small-row dense-layer programs (csrc/rowprog.hip, cgat_rowprog_run) on the GPU, against torch fp64: single ops in
every addressing form, whole networks (SimpleNetwork / ResidualNetwork of the reference, message_changed.py:31-138)
forward + backward through the autograd wrapper, several networks on shared rows, concurrent launches on two streams
(the grid barrier's counters), and hipGraph replay.  Tolerance: max-norm relative 2e-5 -- exact fp32 products, only the
summation order differs from torch's."""
import pytest
import torch

pytestmark = pytest.mark.gpu

TOL = 2e-5


def rel(a, b):
    a, b = a.double().cpu(), b.double().cpu()
    d = b.abs().max().item()
    return (a - b).abs().max().item() / (d if d > 0 else 1.0)


@pytest.fixture(scope="module")
def env():
    import cgat_amd
    from cgat_amd import _lib, rowprog
    yield cgat_amd, _lib, rowprog, torch.device("cuda:0")
    assert rowprog.barrier_timeouts() == 0          # no grid barrier of this module's launches ever gave up


def _act(v, a):
    return {0: lambda t: t, 1: torch.tanh, 2: lambda t: torch.where(t > 0, t, 0.01 * t), 3: torch.relu}[a](v)


def _dact(g, y, a):
    if a == 1:
        return g * (1 - y * y)
    if a == 2:
        return torch.where(y > 0, g, 0.01 * g)
    if a == 3:
        return torch.where(y > 0, g, torch.zeros_like(g))
    return g


@pytest.mark.parametrize("M,N,K", [(64, 1024, 128), (64, 2, 128), (1, 1, 1), (17, 33, 5), (437, 256, 256), (194, 127, 200),
                                   (64, 128, 2), (1280, 384, 128), (2048, 128, 64), (3, 700, 1030)])
@pytest.mark.parametrize("form", ["plain", "full", "transposed"])
def test_single_op(env, M, N, K, form):
    _, _lib, rp, dev = env
    g = torch.Generator().manual_seed(M * 7 + N * 3 + K)
    rnd = lambda *s: torch.randn(*s, generator=g).to(dev)
    out = torch.full((M, N + 3), float("nan"), device=dev)[:, :N]        # a padded leading dimension
    if form == "plain":
        A, B = rnd(M, K), rnd(N, K)
        rp.run([rp.op(0, M, N, K, A, B, out)], dev)
        ref = A.double() @ B.double().t()
        assert rel(out, ref) <= TOL
        return
    if form == "transposed":
        # the weight-gradient form: every operand addressed through its transpose, a derivative on the row operand,
        # row sums out
        At, Dt, Bt = rnd(K, M), rnd(K, M), rnd(K, N)
        rs = torch.full((M,), float("nan"), device=dev)
        rp.run([rp.op(0, M, N, K, At, Bt, out, a_t=True, b0_t=True, dact=Dt, dact_type=3, rowsum=rs)], dev)
        Ad = _dact(At.double(), Dt.double(), 3).t()
        assert rel(out, Ad @ Bt.double()) <= TOL
        assert rel(rs, Ad.sum(1)) <= TOL
        return
    A, D, B0, B1, bias, resid = rnd(M, K), rnd(M, K), rnd(N, K), rnd(N, K), rnd(N), rnd(M, N)
    h = torch.full((M, N), float("nan"), device=dev)
    prev = rnd(M, N)
    out.copy_(prev)
    rp.run([rp.op(0, M, N, K, A, B0, out, dact=D, dact_type=2, B1=B1, bias=bias, act=2, resid=resid, accumulate=True,
                  h_out=h)], dev)
    Ad = _dact(A.double(), D.double(), 2)
    hr = _act(Ad @ B0.double().t() + bias.double(), 2)
    ref = hr + A.double() @ B1.double().t() + resid.double() + prev.double()
    assert rel(h, hr) <= TOL and rel(out, ref) <= TOL


def _torch_nets(x, final_resid, spec, params):
    """fp64 torch evaluation of RowNetsFn's definition."""
    outs, it = [], iter(params)
    for ni, net in enumerate(spec):
        cur = x
        for l, (act, skip, has_b) in enumerate(net):
            W, b, R = next(it), next(it), next(it)
            hcur = _act(cur @ W.reshape(W.shape[0], -1).t() + (b if b is not None else 0), act)
            if skip == 1:
                hcur = hcur + cur
            elif skip == 2:
                hcur = hcur + cur @ R.t()
            cur = hcur
        if ni == 0 and final_resid is not None:
            cur = cur + final_resid
        outs.append(cur)
    return outs


def _check_nets(env, M, in_dim, spec, widths, final_resid=False, seed=0):
    _, _lib, rp, dev = env
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(M, in_dim, generator=g)
    params = []
    for net, ws in zip(spec, widths):
        k = in_dim
        for (act, skip, has_b), n in zip(net, ws):
            params.append(torch.randn(n, k, generator=g) / k ** 0.5)
            params.append(torch.randn(n, generator=g) if has_b else None)
            params.append(torch.randn(n, k, generator=g) / k ** 0.5 if skip == 2 else None)
            k = n
    fr = torch.randn(M, widths[0][-1], generator=g) if final_resid else None
    cots = [torch.randn(M, ws[-1], generator=g) for ws in widths]

    def leaves(dt, device):
        mk = lambda t: None if t is None else t.to(device=device, dtype=dt).requires_grad_(True)
        return mk(x), mk(fr), [mk(p) for p in params]
    xr, frr, pr = leaves(torch.float64, "cpu")
    outs_r = _torch_nets(xr, frr, spec, pr)
    sum((o * c.double()).sum() for o, c in zip(outs_r, cots)).backward()
    xg, frg, pg = leaves(torch.float32, dev)
    outs_g = rp.RowNetsFn.apply(xg, frg, spec, *pg)
    sum((o * c.to(dev)).sum() for o, c in zip(outs_g, cots)).backward()
    for o, r in zip(outs_g, outs_r):
        assert rel(o, r) <= TOL
    assert rel(xg.grad, xr.grad) <= TOL
    if fr is not None:
        assert rel(frg.grad, frr.grad) <= TOL
    for a, b in zip(pg, pr):
        if a is not None:
            assert rel(a.grad, b.grad) <= 5 * TOL, (a.shape,)


def test_output_head_shapes(env):
    """ResidualNetwork(128 -> 1024,1024,512,512,256,256,128 -> 2) at 64 crystals (CGAT.py:526-537)."""
    _, _lib, rp, dev = env
    dims = [128, 1024, 1024, 512, 512, 256, 256, 128]
    spec = tuple((_lib.ACT_RELU, 1 if dims[i] == dims[i + 1] else 2, True) for i in range(7)) + ((0, 0, True),)
    _check_nets(env, 64, 128, (spec,), (dims[1:] + [2],))
    _check_nets(env, 1, 128, (spec,), (dims[1:] + [2],), seed=1)            # a single crystal
    _check_nets(env, 333, 128, (spec[:-1],), (dims[1:],), seed=2)           # last_layer=False


def test_simple_networks_on_shared_rows(env):
    """Roost's gate (2C -> 256 -> 1) and message (2C -> 256 -> C) networks read the same rows (roost_message.py:137-153)."""
    _, _lib, rp, dev = env
    gate, msg = rp.mlp_spec(1, _lib.ACT_LEAKY), rp.mlp_spec(1, _lib.ACT_LEAKY)
    _check_nets(env, 436, 256, (gate, msg), ([256, 1], [256, 128]))
    _check_nets(env, 7, 256, (msg, gate, msg), ([256, 128], [256, 1], [64, 40]), final_resid=True, seed=3)
    _check_nets(env, 194, 200, (rp.mlp_spec(0, 0),), ([127],), seed=4)      # Roost's embedding Linear(200, C - 1)
    _check_nets(env, 50, 24, (rp.mlp_spec(3, _lib.ACT_TANH, has_bias=False),), ([24, 8, 24, 5],), seed=5)


def test_modules_route_through_the_program(env):
    """SimpleNetwork / ResidualNetwork at a few hundred rows issue ONE library launch per direction."""
    P, _lib, rp, dev = env
    from cgat_amd import ops
    torch.manual_seed(0)
    net = P.ResidualNetwork(128, 2, [1024, 1024, 512, 512, 256, 256, 128]).to(dev)
    x = torch.randn(64, 128, device=dev, requires_grad=True)
    net(x).sum().backward()                          # warm-up (allocations, the barrier counters)
    n0 = ops.prof_launches()
    y = net(x)
    n1 = ops.prof_launches()
    y.sum().backward()
    n2 = ops.prof_launches()
    assert (n1 - n0, n2 - n1) == (1, 1)
    sn = P.SimpleNetwork(256, 1, [256]).to(dev)
    z = torch.randn(436, 256, device=dev, requires_grad=True)
    n0 = ops.prof_launches()
    sn(z).sum().backward()
    assert ops.prof_launches() - n0 == 2


def test_concurrent_programs_and_determinism(env):
    """Two streams run multi-phase programs at the same time (each launch has its own barrier counter); every
    repetition is bit-identical to the first."""
    _, _lib, rp, dev = env
    dims = [128, 1024, 1024, 512, 512, 256, 256, 128]
    spec = (tuple((_lib.ACT_RELU, 1 if dims[i] == dims[i + 1] else 2, True) for i in range(7)) + ((0, 0, True),),)
    g = torch.Generator().manual_seed(11)
    params, k = [], 128
    for n in dims[1:] + [2]:
        params += [(torch.randn(n, k, generator=g) / k ** 0.5).to(dev), torch.randn(n, generator=g).to(dev),
                   (torch.randn(n, k, generator=g) / k ** 0.5).to(dev) if n != k else None]
        k = n
    xs = [torch.randn(64, 128, generator=g).to(dev) for _ in range(2)]
    first = [rp.RowNetsFn.apply(x, None, spec, *params)[0].clone() for x in xs]
    streams = [torch.cuda.Stream(dev), torch.cuda.Stream(dev)]
    torch.cuda.synchronize()
    outs = [[], []]
    for rep in range(20):
        for s, st in enumerate(streams):
            with torch.cuda.stream(st):
                outs[s].append(rp.RowNetsFn.apply(xs[s], None, spec, *params)[0])
    torch.cuda.synchronize()
    for s in range(2):
        for o in outs[s]:
            assert torch.equal(o, first[s])


def test_program_replays_in_a_hipgraph(env):
    P, _lib, rp, dev = env
    torch.manual_seed(0)
    net = P.ResidualNetwork(128, 2, [1024, 1024, 512, 512, 256, 256, 128]).to(dev)
    x = torch.randn(64, 128, device=dev)
    xg = x.clone().requires_grad_(True)
    net(xg).square().sum().backward()
    ref_y, ref_g = net(xg).detach().clone(), xg.grad.clone()
    ref_w = net.fcs[0].weight.grad.clone()

    def step():
        xg.grad = None
        for p in net.parameters():
            p.grad = None
        y = net(xg)
        y.square().sum().backward()
        return y
    gs = P.GraphedStep(step, warmup=2)
    for _ in range(3):
        gs.replay()
    torch.cuda.synchronize()
    assert torch.equal(xg.grad, ref_g) and torch.equal(net.fcs[0].weight.grad, ref_w)
