#!/usr/bin/env python3
"""bench.py -- BASELINE.json's metric on MI355X:
    "edges/sec through one CGAT attention layer (fwd+bwd), 1M-edge batch".

One step = one GATConvNodes.forward (message MLPs, softmax over incoming edges, scatter-add,
head mean, H_Net hypernetwork update; reference CGAT.py:307-335) plus its full backward
(gradients wrt x, edge_attr, x_0 and all 9.48 M layer parameters) over one synthetic batch of
4167 crystals x 20 atoms x 12 neighbours = 83 340 atoms, 1 000 080 edges, C = Ce = 128, H = 3,
fp32.  Inputs are resident in HBM before the timed region.  With N > 1 every rank runs its own
batch of that size (graphs are independent; weak scaling) and the step ends with the gradient
mean across ranks (RCCL all-reduce), as Lightning DDP does for the reference.

    python bench.py --gpus 1 --steps 5 --warmup 2
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W
Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

C_FEA, HEADS, K_NBR, ATOMS = 128, 3, 12, 20
GRAPHS = 4167                       # -> E = 1 000 080
MFMA_F32_PEAK_TFLOPS = 157.3        # MI355X_MICROARCH.md: dense f32-input matrix peak (= vector peak)
MFMA_BF16_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense bf16 matrix peak
# CGAT_DIST_FORCE=1: the gradient all-reduce path (GradientAverager over a ONE-RANK RCCL communicator) also at N = 1, e.g.
#   CGAT_DIST_FORCE=1 python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 bench.py --gpus 1
# -- the step then contains what every rank of an N > 1 run does besides its own layer step (bucket views, hooks,
# asynchronous all-reduces, used-bitmap reduction); reported as "allreduce" in the line.
FORCE_ALLREDUCE = os.environ.get("CGAT_DIST_FORCE") == "1"


def make_inputs(graphs, seed, device, K=K_NBR):
    import cgat_amd as P
    b, _ = P.synthetic_batch(graphs, ATOMS, K, seed=seed)
    N, E = b.num_nodes, b.edge_index.shape[1]
    if E > (8 << 20):                                      # the 64M-edge batch: features drawn on the device
        g = torch.Generator(device=device).manual_seed(1000 + seed)
        x, e, x0, cot = (torch.randn(n, C_FEA, generator=g, device=device) for n in (N, E, N, N))
        return b.edge_index.to(device), x, e, x0, cot
    g = torch.Generator().manual_seed(1000 + seed)
    x = torch.randn(N, C_FEA, generator=g).to(device)
    e = torch.randn(E, C_FEA, generator=g).to(device)
    x0 = torch.randn(N, C_FEA, generator=g).to(device)
    cot = torch.randn(N, C_FEA, generator=g).to(device)
    return b.edge_index.to(device), x, e, x0, cot


def _cpu_layer_time(layer, n_graphs, backward, reps, warm):
    """Seconds of one pass of the oracle layer over n_graphs synthetic crystals on the host cores (lower median of
    `reps` repetitions: the faster one of two)."""
    ei, x, e, x0, cot = make_inputs(n_graphs, 0, "cpu")
    times = []
    for r in range(reps + warm):
        t0 = time.perf_counter()
        if backward:
            xx, ee, xx0 = (t.clone().requires_grad_(True) for t in (x, e, x0))
            y = layer(xx, ei, ee, xx0)
            torch.autograd.grad((y * cot).sum(), [xx, ee, xx0] + list(layer.parameters()))
        else:
            with torch.no_grad():
                layer(x, ei, e, x0)
        dt = time.perf_counter() - t0
        if r >= warm:
            times.append(dt)
    times.sort()
    return times[(len(times) - 1) // 2], int(ei.shape[1])


def cpu_baseline():
    """The oracle (op-for-op restatement of the reference's CPU path: cat -> head repeat -> grouped
    Conv1d -> LeakyReLU -> conv -> segment softmax -> scatter-add -> Linear(C -> C*C+C) hypernet ->
    bmm -> LayerNorm -> tanh) timed on this box's host cores as BASELINE.md §3 specifies: BASELINE config 1 run
    whole (1000 crystals, E = 240 000, one layer forward, no_grad) and the metric's fwd+bwd at E = 60 000, 240 000 and
    480 000 of the same synthetic workload (the 1M-edge batch needs ~240 GB of host memory for the reference's
    materialised hypernetwork weights; 480 000 edges need ~30 GB), 2 repetitions each (the faster one), with the
    linearity of the per-edge cost judged on the three sizes: `value` is the rate MEASURED at the largest size,
    `asymptotic` the marginal rate between the two largest sizes (the slope of time over edges: what a fixed per-call
    overhead cannot distort) -- the number to extrapolate to 1M edges with."""
    from oracle import cgat_oracle as O
    torch.manual_seed(1)
    layer = O.GATConvNodes(C_FEA, C_FEA, C_FEA, HEADS, concat=True)
    t60, e60 = _cpu_layer_time(layer, 250, True, reps=2, warm=1)
    t240, e240 = _cpu_layer_time(layer, 1000, True, reps=2, warm=0)
    t480, e480 = _cpu_layer_time(layer, 2000, True, reps=2, warm=0)
    tf, ef = _cpu_layer_time(layer, 1000, False, reps=1, warm=0)
    r60, r240, r480 = e60 / t60, e240 / t240, e480 / t480
    marginal = (e480 - e240) / (t480 - t240) if t480 > t240 else r480
    return {"value": r480, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "config1_fwd": {"edges": ef, "seconds": round(tf, 3), "edges_per_s": round(ef / tf, 1),
                            "what": "BASELINE configs[0]: 1000 crystals, one layer forward, no_grad, run whole"},
            "fwdbwd_60k": {"edges": e60, "seconds": round(t60, 3), "edges_per_s": round(r60, 1)},
            "fwdbwd_240k": {"edges": e240, "seconds": round(t240, 3), "edges_per_s": round(r240, 1)},
            "fwdbwd_480k": {"edges": e480, "seconds": round(t480, 3), "edges_per_s": round(r480, 1)},
            "asymptotic": {"edges_per_s": round(marginal, 1),
                           "what": "(E480k - E240k) / (t480k - t240k): the marginal rate between the two largest sizes; "
                                   "value / asymptotic -> 1 as the fixed per-call cost stops mattering"},
            "linearity": {"per_edge_cost_ratio_240k_over_60k": round((t240 / e240) / (t60 / e60), 3),
                          "per_edge_cost_ratio_480k_over_240k": round((t480 / e480) / (t240 / e240), 3),
                          "what": "1.0 = the per-edge cost does not depend on the batch size; the second ratio says how far "
                                  "the rate at 480 000 edges is from extrapolating to the 1M-edge batch"},
            "sample": f"oracle layer fwd+bwd on {e60} edges (best of 2 after 1 warm-up), on {e240} and on {e480} edges "
                      f"(best of 2 each), forward on {ef} edges (1 pass), fp32, all host threads; run BEFORE the GPU leg"}


def cpu_baseline_collate(data, emb, n_graphs, reps=5):
    """The oracle's restatement of the reference's host-side collation (CompositionData.__getitem__ for every crystal
    + Batch.from_data_list + collate_batch) on the same batch."""
    import numpy as np
    from oracle import collate_oracle as O
    from cgat_amd.graph import ELEMENT_SYMBOLS
    elem_id = {el: k for k, el in enumerate(ELEMENT_SYMBOLS)}
    table = np.asarray([emb[el] for el in ELEMENT_SYMBOLS], dtype=np.float32)
    t0 = time.perf_counter()
    for _ in range(reps):
        ref = O.collate(data, list(range(n_graphs)), table, elem_id, K_NBR, "e_above_hull")
    dt = (time.perf_counter() - t0) / reps
    return {"value": ref["edge_attr"].shape[0] / dt, "unit": "edges/s", "cores": 1, "kind": "port",
            "sample": f"the same {n_graphs}-crystal batch through the oracle's restatement of CompositionData.__getitem__ "
                      f"+ collate_batch (numpy, mean of {reps}, {dt:.2f} s per batch; without the host-to-device copy the "
                      "reference adds)"}


def bench_collate(args, rank, world, device):
    """Informational: device-side collation of the BASELINE batch (4167 crystals x 20 atoms x 12
    neighbours -> the tensors the layer consumes) from a packed HBM-resident dataset; one step = one batch.  The
    CPU baseline is the oracle's restatement of the reference's per-crystal Python loops on a bounded sample."""
    import numpy as np
    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.graph import synthetic_dataset_dict
    data, emb = synthetic_dataset_dict(args.graphs, ATOMS, 24, seed=rank)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=K_NBR, device=device)
    rs = np.random.RandomState(rank)
    batches = [rs.permutation(args.graphs) for _ in range(args.warmup + args.steps)]
    for i in range(args.warmup):
        ds.collate(batches[i])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for i in range(args.steps):
        gb, _ = ds.collate(batches[args.warmup + i])
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        N, E = gb.num_nodes, gb.edge_index.shape[1]
        n_l, ms_l = ops.prof_get("collate")
        # algorithmic HBM bytes per batch: embedding rows written (800 B per atom; the 82-KB table itself stays in
        # L2; the composition rows are ~1/6 of the atom rows), three int32 tables read and three int64 arrays
        # written per edge, batch vector
        alg = N * 800 * (1 + 1 / 6) + E * (12 + 24) + N * 8
        roof = None
        if n_l:
            avg_ms = ms_l / n_l
            roof = {"bound": "hbm", "kernel": "collate_kernel", "achieved": round(alg / (avg_ms * 1e-3) / 1e9, 1),
                    "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (avg_ms * 1e-3) / 1e9 / 8000.0, 4), "traffic": None,
                    "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg)}
        out = {"metric": "batch-edges/sec collated on the device (dataset -> layer inputs) [informational, SURVEY 8 f1]",
               "value": world * E * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "int32/int64 indices, f32 rows", "data": "synthetic",
               "config": {"workload": f"collate {args.graphs} crystals x {ATOMS} atoms x {K_NBR} of 24 stored neighbours "
                                      f"per rank from a packed HBM-resident dataset: N={N}, E={E}",
                          "edges_per_rank": E, "parallelism": f"dp{world} (every rank collates its own crystals)"},
               "roofline": roof}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_collate(data, emb, args.graphs)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def cpu_baseline_optim(shapes, lr, wd):
    """The oracle's AdamW update (torch fp32 on the host cores) over the same tensors."""
    from oracle import optim_oracle as O
    g = torch.Generator().manual_seed(0)
    ps = [torch.randn(sh, generator=g) for sh in shapes]
    gs = [torch.randn(sh, generator=g) for sh in shapes]
    ms, vs = [torch.zeros(sh) for sh in shapes], [torch.zeros(sh) for sh in shapes]
    times = []
    for step in range(1, 5):
        t0 = time.perf_counter()
        for i in range(len(ps)):
            ps[i], ms[i], vs[i] = O.adamw_step(ps[i], gs[i], ms[i], vs[i], step, lr=lr, weight_decay=wd)
        times.append(time.perf_counter() - t0)
    dt = sorted(times[1:])[1]
    n = sum(p.numel() for p in ps)
    return {"value": n / dt, "unit": "parameters/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"all {len(ps)} tensors ({n} parameters), median of 3 steps after 1 warm-up ({dt:.3f} s per step)"}


def bench_optim(args, rank, world, device):
    """Informational: one fused AdamW step over all 44.6 M parameters of CGAtNet(200,128,4,msg_heads=3)."""
    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.optim import FusedAdamW
    torch.manual_seed(1)
    net = P.CGAtNet(200, C_FEA, 4, msg_heads=HEADS, neighbor_number=K_NBR, update_edges=True).to(device)
    params = list(net.parameters())
    g = torch.Generator().manual_seed(2)
    for p in params:
        p.grad = torch.randn(p.shape, generator=g).to(device)
    opt = FusedAdamW(params, lr=1e-3, weight_decay=1e-2)
    for _ in range(args.warmup):
        opt.step()
    torch.cuda.synchronize()
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        opt.step()
    torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        n = sum(p.numel() for p in params)
        n_l, ms_l = ops.prof_get("adamw")
        alg = 28.0 * n                                     # read p, g, m, v; write p, m, v
        avg_ms = ms_l / max(n_l, 1)
        out = {"metric": "parameters/sec through one fused AdamW step [informational, SURVEY 8 f4]",
               "value": world * n * args.steps / elapsed, "unit": "parameters/s", "n_gpus": world, "steps": args.steps,
               "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True,
               "scaling": "weak", "vs_baseline": None, "dtype": "f32", "data": "synthetic",
               "config": {"workload": f"FusedAdamW over the {len(params)} tensors / {n} parameters of "
                                      "CGAtNet(200,128,4,msg_heads=3,update_edges=True), one launch per step",
                          "parallelism": f"dp{world} (replicated optimiser state)"},
               "roofline": {"bound": "hbm", "kernel": "adamw_mt_kernel", "achieved": round(alg / (avg_ms * 1e-3) / 1e9, 1),
                            "peak": 8000.0, "unit": "GB/s", "frac": round(alg / (avg_ms * 1e-3) / 1e9 / 8000.0, 4),
                            "traffic": None, "avg_launch_ms": round(avg_ms, 4), "algorithmic_bytes_per_launch": int(alg)}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline_optim([tuple(p.shape) for p in params], 1e-3, 1e-2)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_edge_hyper(args, rank, world, device):
    """Informational: the edge update WITH its per-edge hypernetwork, GATConvEdges(no_hyper=False)
    (reference CGAT.py:187-229; not the shipped configuration, which discards this branch): forward + full backward
    on the 1M-edge batch.  Per edge: the attention / message networks over [x_i; e; x_j], head softmax, and an H_Net
    update whose contractions run over E rows instead of N."""
    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    torch.manual_seed(1)
    layer = P.GATConvEdges(C_FEA, C_FEA, C_FEA, HEADS, concat=True, no_hyper=False).to(device)
    params = list(layer.parameters())
    ei, x, e, x0, cot = make_inputs(args.graphs, rank, device, K_NBR)
    g = torch.Generator().manual_seed(7 + rank)
    E = ei.shape[1]
    e0 = torch.randn(E, C_FEA, generator=g).to(device)     # the edge features the damping mixes in (x_0 of the edge net)
    cot_e = torch.randn(E, C_FEA, generator=g).to(device)
    x.requires_grad_(True); e.requires_grad_(True)

    def step():
        for p in params:
            p.grad = None
        x.grad = e.grad = None
        y = layer(x, ei, e, e0)
        y.backward(cot_e)
    for _ in range(args.warmup):
        step()
    _fence(world)
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _fence(world)
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    if rank == 0:
        tags = ("bilinear_rows", "bilinear_dual", "bilinear_wgrad", "mlp_chain", "rows_dw", "linear128", "gemm_f32", "edge_z")
        shares = {}
        for t_ in tags:
            n_t, ms_t = ops.prof_get(t_)
            if n_t:
                shares[t_] = {"launches_per_step": n_t / args.steps, "ms_per_step": round(ms_t / args.steps, 3)}
        fl = 2.0 * E * C_FEA ** 3                           # one contraction launch over E rows
        n_r, ms_r = ops.prof_get("bilinear_rows")
        roof = None
        if n_r:
            ach = fl / (ms_r / n_r * 1e-3) / 1e12
            # the forward contraction kernel of the arithmetic mode this process runs in and its matrix pass-equivalents
            # per product (the headline's table, bench_layer): VERDICT r5 found the f16x3 kernel named in f16x3c mode
            kname, passes, peak0 = {"f16x3c": ("bilinear_rows128_ring16c_kernel", 3.75, MFMA_BF16_PEAK_TFLOPS),
                                    "f16x3": ("bilinear_rows128_ring16_kernel", 3, MFMA_BF16_PEAK_TFLOPS),
                                    "bf16x6": ("bilinear_rows128_ring16_kernel", 6, MFMA_BF16_PEAK_TFLOPS),
                                    "f32": ("bilinear_rows128_kernel", 1, MFMA_F32_PEAK_TFLOPS)}[P.get_bilinear_mode()]
            peak = peak0 / passes
            roof = {"bound": "mfma", "kernel": kname, "matrix_passes_per_product": passes, "achieved": round(ach, 2),
                    "peak": round(peak, 1), "unit": "TFLOP/s", "frac": round(ach / peak, 4), "traffic": None,
                    "avg_launch_ms": round(ms_r / n_r, 4), "flops_per_launch": fl}
        print(json.dumps({
            "metric": "edges/sec through one GATConvEdges(no_hyper=False) layer (fwd+bwd), 1M-edge batch [informational, SURVEY 8 f2]",
            "value": world * E * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 storage / " + P.get_bilinear_mode(), "data": "synthetic",
            "config": {"workload": f"GATConvEdges({C_FEA},{C_FEA},{C_FEA},heads={HEADS},no_hyper=False) fwd+bwd, "
                                   f"{args.graphs} crystals x {ATOMS} atoms x {K_NBR} nbrs: E={E} rows through the per-edge H_Net",
                       "parallelism": f"dp{world}"},
            "roofline": roof, "kernel_ms_per_step": shares}), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def bench_lightning(args, rank, world, device):
    """The network the reference harness builds with its argparse defaults (lightning_module.py:427-593 -> 165-176):
    CGAtNet(200, 128, n_graph=5, msg_heads=5, neighbor_number=24, vector_attention=True, global_vector_attention=True,
    mean_pooling=False, rezero=True, n_graph_roost=3), forward + backward of the L1 loss on a ~1M-edge batch
    (2083 crystals x 20 atoms x 24 neighbours = 999 840 edges).  The roofline object is for the kernel tag with the most
    GPU time in the timed region."""
    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    H, K, L = 5, 24, 5
    graphs = args.graphs if args.graphs != GRAPHS else 2083
    torch.manual_seed(1)
    net = P.CGAtNet(200, C_FEA, L, msg_heads=H, neighbor_number=K, update_edges=True, vector_attention=True,
                    global_vector_attention=True, mean_pooling=False, rezero=True, n_graph_roost=3).to(device)
    params = list(net.parameters())
    b, roost = P.synthetic_batch(graphs, ATOMS, K, seed=rank)
    b = b.to(device)
    roost = tuple(t.to(device) for t in roost)
    N, E = b.num_nodes, b.edge_index.shape[1]

    def step():
        for p in params:
            p.grad = None
        out = net(b, roost)
        (out[:, 0] - b.y).abs().mean().backward()
    for _ in range(args.warmup):
        step()
    _fence(world)
    ops.prof_reset(); ops.prof_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _fence(world)
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    launch_rep = None
    if graphs < 512 or args.hipgraph:
        # the harness' default network at the harness' default batch (--batch-size 64): per-tag event timing off, eager
        # step re-timed, then launches / host syncs per step and the step replayed as one hipGraph
        _fence(world)
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        _fence(world)
        elapsed = time.perf_counter() - t0
        launch_rep = _launch_report(step, args.steps, world, try_graph=True)
    if rank == 0:
        Hd, W2 = 256, 2 * H * 256
        tags = ("edge_z", "linear128", "seg_attnpool_fwd", "seg_attnpool_bwd", "bilinear_rows", "bilinear_dual",
                "bilinear_wgrad", "edge_ge", "edge_gw", "rows_ge", "rows_gw", "edge_gj", "mlp_chain", "rows_dw", "gemm_f32")
        shares = {}
        for t_ in tags:
            n_t, ms_t = ops.prof_get(t_)
            if n_t:
                shares[t_] = {"launches_per_step": n_t / args.steps, "ms_per_step": round(ms_t / args.steps, 3),
                              "avg_launch_ms": round(ms_t / n_t, 4)}
        # algorithmic work per launch of the tags that can dominate (per Node layer; C = 128, H = 5, Hd = 256):
        #   edge_z (first layers of both networks, operand split): writes hidden [E, W2] fp32, reads e and gathers Pj
        #   linear128 (per-head second layers, 128-wide blocks): 2 * rows * 128 * 128 flop per launch
        #   seg_attnpool_*: reads logits + messages [E, 2 * H * C] (+ the gradient pair in backward)
        alg = {"edge_z": ("hbm", E * (W2 * 4.0 + W2 * 4.0 + C_FEA * 4.0) + N * W2 * 4.0),
               "seg_attnpool_fwd": ("hbm", E * 2.0 * H * C_FEA * 4 + N * H * C_FEA * 4.0),
               "seg_attnpool_bwd": ("hbm", E * 4.0 * H * C_FEA * 4 + 2.0 * N * H * C_FEA * 4),
               "linear128": ("mfma", 2.0 * E * 128 * 128)}
        roof = None
        if shares:
            dom = max(shares, key=lambda k: shares[k]["ms_per_step"])
            roof = {"kernel_tag": dom, **shares[dom]}
            if dom in alg:
                kind, amount = alg[dom]
                avg_s = shares[dom]["avg_launch_ms"] * 1e-3
                if kind == "hbm":
                    roof.update({"bound": "hbm", "achieved": round(amount / avg_s / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                                 "frac": round(amount / avg_s / 8e12, 4), "traffic": None,
                                 "algorithmic_bytes_per_launch": int(amount)})
                else:
                    peak = MFMA_BF16_PEAK_TFLOPS / 3
                    roof.update({"bound": "mfma", "achieved": round(amount / avg_s / 1e12, 2), "peak": round(peak, 1),
                                 "unit": "TFLOP/s", "frac": round(amount / avg_s / 1e12 / peak, 4), "traffic": None,
                                 "flops_per_launch": amount,
                                 "note": "per-launch figure at E rows; launches over N rows (trunks) share the tag"})
        n_par = sum(p.numel() for p in params)
        print(json.dumps({
            "metric": "batch-edges/sec through the reference harness' DEFAULT network (5 layers, 5 heads, 24 neighbours, vector "
                      "attention, concat pooling, rezero), fwd+bwd, ~1M-edge batch [informational, SURVEY 8 f2]",
            "value": world * E * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32 storage / " + P.get_bilinear_mode(), "data": "synthetic",
            "config": {"workload": f"CGAtNet(200,128,5,msg_heads=5,neighbor_number=24,vector_attention=True,"
                                   f"global_vector_attention=True,mean_pooling=False,rezero=True) fwd+bwd of the L1 loss, "
                                   f"{graphs} crystals x {ATOMS} atoms x {K} nbrs: N={N}, E={E}, {n_par} parameters",
                       "edge_layer_passes_per_s": round(L * E * args.steps / elapsed, 1), "parallelism": f"dp{world}"},
            "roofline": roof, "kernel_ms_per_step": shares, "launch_bound": launch_rep,
            "peak_memory_GB": round(torch.cuda.max_memory_allocated() / 1e9, 1)}), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def bench_train(args, rank, world, device):
    """BASELINE configs[3]: the end-to-end data-parallel training step on a DCGAT-shaped synthetic dataset (ragged
    crystals of 2..40 atoms, 24 stored / 12 used neighbours, y = e_above_hull * n_atoms): device collation of the
    rank's crystals -> CGAtNet(200,128,4,msg_heads=3) forward -> RobustL1 -> backward with the bucketed gradient
    all-reduce (RCCL over xGMI) overlapped -> one fused AdamW launch.  weak: every rank trains on --graphs crystals
    per step; strong: the --graphs crystals of a step are split across the ranks."""
    import numpy as np
    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.graph import synthetic_dataset_dict
    from cgat_amd.trainer import DataParallelTrainer
    per_rank = args.graphs if args.scaling == "weak" else max(1, args.graphs // world)
    n_data = 2 * per_rank                                  # this rank's shard of the dataset, resident in HBM
    data, emb = synthetic_dataset_dict(n_data, (2, 40), 24, seed=100 + rank)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=K_NBR, device=device)
    torch.manual_seed(1)                                   # identical replicas
    net = P.CGAtNet(200, C_FEA, 4, msg_heads=HEADS, neighbor_number=K_NBR, update_edges=True).to(device)
    tr = DataParallelTrainer(net, ds, lr=1e-4, weight_decay=1e-6, rank=rank, world=world, force_averager=FORCE_ALLREDUCE)
    ops.set_validate_indices(False)                        # the collation kernel is the only producer of the indices
    rs = np.random.RandomState(rank)
    batches = [rs.permutation(n_data)[:per_rank] for _ in range(args.warmup + args.steps)]
    edges = 0
    for i in range(args.warmup):
        tr.step(batches[i])
    _fence(world)
    marks = [torch.cuda.Event(enable_timing=True) for _ in range(args.steps + 1)]
    t0 = time.perf_counter()
    marks[0].record()
    for i in range(args.steps):
        _, e = tr.step(batches[args.warmup + i])
        marks[i + 1].record()
        edges += e
    _fence(world)
    elapsed = time.perf_counter() - t0
    step_ms = [round(marks[i].elapsed_time(marks[i + 1]), 2) for i in range(args.steps)]
    n0 = ops.prof_launches()
    tr.step(batches[0])
    train_launches = ops.prof_launches() - n0
    train_syncs = _count_host_syncs(lambda: tr.step(batches[0]))
    tot = torch.tensor([elapsed, float(edges)], device=device, dtype=torch.float64)
    if world > 1:
        mx = tot.clone()
        dist.all_reduce(mx, op=dist.ReduceOp.MAX)
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        elapsed, edges = float(mx[0]), float(tot[1])
    if rank == 0:
        n_par = sum(p.numel() for p in net.parameters())
        out = {"metric": "batch-edges/sec through the end-to-end data-parallel training step (collate -> CGAtNet fwd+bwd -> "
                         "gradient all-reduce -> AdamW), DCGAT-shaped synthetic dataset [BASELINE configs[3]]",
               "value": edges / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": 1e3 * elapsed / args.steps, "step_ms_gpu_events": step_ms,
               "launch_bound": {"library_kernel_launches_per_step": train_launches, "host_syncs_per_step": train_syncs},
               "higher_is_better": True, "scaling": args.scaling,
               "vs_baseline": None, "dtype": "f32 storage / " + P.get_bilinear_mode(),
               "data": "synthetic", "bilinear_mode": P.get_bilinear_mode(),
               "config": {"workload": f"train step: {per_rank} ragged crystals (2..40 atoms, 24 stored / {K_NBR} used nbrs) per "
                                      f"rank and step, CGAtNet(200,128,4,msg_heads=3,update_edges=True), RobustL1, FusedAdamW, "
                                      f"{n_par} parameters = {4 * n_par / 1e6:.0f} MB of gradients all-reduced per step",
                          "edges_per_step_all_ranks": edges / args.steps,
                          "parallelism": f"dp{world} (graphs sharded, bucketed gradient all-reduce overlapped with backward)"},
               "roofline": None}
        if tr.averager is not None:
            out["allreduce"] = _allreduce_report(tr.averager, args.warmup + args.steps)
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


def _allreduce_report(avg, n_steps):
    import torch.distributed as dist
    st = avg.stats
    return {"backend": dist.get_backend(), "world": avg.world, "buckets": len(avg.buckets), "hot_buckets": avg.n_hot,
            "launched_from_hooks_per_step": round(st["launched_in_backward"] / n_steps, 2),
            "launched_in_finish_per_step": round(st["launched_in_finish"] / n_steps, 2),
            "cold_buckets_skipped_per_step": round(st["cold_skipped"] / n_steps, 2),
            "MB_reduced_per_step": round(st["bytes_reduced"] / n_steps / 1e6, 1),
            "what": "cgat_amd.dist.GradientAverager: gradients live in the buckets, hot buckets are all-reduced from the "
                    "autograd hooks while backward is running, parameters unused on every rank are cold (not reduced)"}


def _count_host_syncs(step):
    """Host synchronisations of ONE step, counted by torch's sync-debug mode (every .item() / .tolist() / blocking copy
    of a device tensor warns); the library's own calls never synchronise (include/cgat_hip.h)."""
    import warnings
    torch.cuda.synchronize()
    torch.cuda.set_sync_debug_mode("warn")
    try:
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            step()
    finally:
        torch.cuda.set_sync_debug_mode("default")
    torch.cuda.synchronize()
    return sum(1 for x in w if "synchroniz" in str(x.message).lower())


def _count_all_kernels(step):
    """Every GPU kernel of ONE step (the library's and torch's own elementwise / copy / fill kernels), counted by
    torch.profiler's device activity; None when another tracer (rocprofv3) owns the process or the profiler is missing."""
    if any(k in os.environ for k in ("ROCP_TOOL_LIBRARIES", "ROCPROFILER_LIBRARY_CTOR")) or \
            "rocprofiler" in os.environ.get("LD_PRELOAD", ""):
        return None
    try:
        from torch.profiler import ProfilerActivity, profile
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            step()
            torch.cuda.synchronize()
        n = 0
        for ev in prof.events():
            if getattr(ev, "device_type", None) is not None and "cuda" in str(ev.device_type).lower():
                name = ev.name.lower()
                if not (name.startswith("memcpy") or name.startswith("memset")):
                    n += 1
        return n or None
    except Exception:
        return None


def _launch_report(step, steps, world, try_graph):
    """Launch-bound regime (the reference's shipped --batch-size 64, lightning_module.py:468-473): library kernel launches
    and host synchronisations per step, and the same step captured into ONE hipGraph (cgat_amd.GraphedStep) and
    replayed."""
    import cgat_amd as P
    from cgat_amd import ops
    n0 = ops.prof_launches()
    step()
    launches = ops.prof_launches() - n0
    rep = {"library_kernel_launches_per_step": launches, "all_kernel_launches_per_step": _count_all_kernels(step),
           "host_syncs_per_step": _count_host_syncs(step),
           "what": "library launches = kernels issued by libcgat_hip in one step; all = every GPU kernel of the step "
                   "incl. torch's own elementwise / copy / fill kernels (torch.profiler device activity; null under "
                   "rocprofv3); host syncs by torch.cuda.set_sync_debug_mode"}
    if try_graph:
        try:
            gs = P.GraphedStep(step, warmup=2)
            for _ in range(3):
                gs.replay()
            _fence(world)
            t0 = time.perf_counter()
            for _ in range(steps):
                gs.replay()
            _fence(world)
            rep["hipgraph"] = {"ms_per_step": round(1e3 * (time.perf_counter() - t0) / steps, 4),
                               "library_kernel_launches_in_graph": gs.kernel_launches,
                               "what": "the same step (forward + backward) captured once into a hipGraph and replayed: one "
                                       "host-side launch per step; bit-identical results (tests/test_capture.py)"}
        except Exception as ex:                      # never let the informational leg take the bench line down
            rep["hipgraph"] = {"error": repr(ex)[:300]}
    return rep


def _counters_of_this_build(path, mode, key=None):
    """(data, source label, stale note): a builder-collected counter summary under profiles/ -- replayed in the line only
    if its stamp names THIS source tree (tools/tree_id.py) and this arithmetic mode; ({}, None, why) otherwise."""
    if not os.path.exists(path):
        return {}, None, None
    data = json.load(open(path))
    if key is not None:
        data = data.get(key, {})
        if not data:
            return {}, None, None
    name = "profiles/" + os.path.basename(path) + (f"[{key}]" if key else "")
    stamp = str(data.get("collected_at_commit", "unknown commit"))
    try:
        sys.path.insert(0, os.path.join(ROOT, "tools"))
        from tree_id import tree_id
        mine = tree_id()
    except Exception:
        mine = None
    import re
    m = re.search(r"source tree ([0-9a-f]{16})", stamp)
    if mine is None or m is None or m.group(1) != mine or data.get("mode", mode) != mode:
        return {}, None, (f"{name} @ {stamp}, mode {data.get('mode', '?')}: not this build (source tree {mine}, mode {mode}) "
                          "-- not replayed")
    return data, (f"{name} @ {stamp} (builder-collected rocprofv3 --pmc passes of this command on this source tree, "
                  "replayed here; not measured in this run)"), None


def _fence(world):
    import torch.distributed as dist
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()


def _time_steps(step, warmup, steps, world):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; seconds."""
    for _ in range(warmup):
        step()
    _fence(world)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    _fence(world)
    return time.perf_counter() - t0


def make_layer_workload(graphs, rank, world, device, K=K_NBR):
    """One GATConvNodes layer (non-first: H_Net with damping), forward + full backward."""
    import cgat_amd as P
    from cgat_amd.dist import GradientAverager
    torch.manual_seed(1)                                   # identical parameters on every rank
    layer = P.GATConvNodes(C_FEA, C_FEA, C_FEA, HEADS, concat=True).to(device)
    params = list(layer.parameters())
    ei, x, e, x0, cot = make_inputs(graphs, rank, device, K)  # each rank: its own crystals
    x.requires_grad_(True); e.requires_grad_(True); x0.requires_grad_(True)
    averager = GradientAverager(params, force=FORCE_ALLREDUCE, static_graph=True) if (world > 1 or FORCE_ALLREDUCE) else None

    def step():
        if averager is not None:
            averager.zero_grad()                           # gradients accumulate straight into the all-reduce buckets
        else:
            for p in params:
                p.grad = None
        x.grad = e.grad = x0.grad = None
        y = layer(x, ei, e, x0)
        y.backward(cot)
        if averager is not None:
            averager.finish()
    step.averager = averager
    return step, x.shape[0], ei.shape[1]


def make_stack_workload(graphs, rank, world, device):
    """BASELINE configs[2]: CGAtNet(200,128,4,msg_heads=3) forward + backward of the L1 loss on the same batch."""
    import cgat_amd as P
    from cgat_amd.dist import GradientAverager
    torch.manual_seed(1)
    net = P.CGAtNet(200, C_FEA, 4, msg_heads=HEADS, neighbor_number=K_NBR, update_edges=True).to(device)
    params = list(net.parameters())
    b, roost = P.synthetic_batch(graphs, ATOMS, K_NBR, seed=rank)
    b = b.to(device)
    roost = tuple(t.to(device) for t in roost)
    averager = GradientAverager(params, force=FORCE_ALLREDUCE, static_graph=True) if (world > 1 or FORCE_ALLREDUCE) else None

    def step():
        if averager is not None:
            averager.zero_grad()
        else:
            for p in params:
                p.grad = None
        out = net(b, roost)
        loss = (out[:, 0] - b.y).abs().mean()              # L1 on the prediction column, as the harness' default
        loss.backward()
        if averager is not None:
            averager.finish()
    return step, b.num_nodes, b.edge_index.shape[1]


# Executed-algorithmic work of ONE layer step (forward + backward) at N atoms, E edges, C = Ce = 128, H = 3, Hd = 256:
# what the kernels compute after the algebra of DESIGN.md §3 (one product counted once, whatever number of matrix-core
# passes the arithmetic mode spends on it).
def layer_step_flops(N, E):
    C, W2 = C_FEA, 2 * HEADS * 256
    hyper = 12 * 2.0 * N * C ** 3                          # 4 forward + 4 fused-backward + 4 weight-gradient contractions
    dense = 4 * 18 * 2.0 * N * C * C                       # per predicted layer: 6 dense 128x128 layers, x3 (fwd, dX, dW)
    edge = 2.0 * E * (2 * C) * W2 + 2 * 2.0 * E * W2 * C   # per-edge pre-activations (K = Ce + C), g_e and dW_e products
    node = 2.0 * N * C * W2 * (1 + 4) + 3 * 2 * 2.0 * N * (W2 // 2) * C   # Pi; g_x / dW_i / dW_j; fc_out of MH_M x3
    return hyper + dense + edge + node


def layer_step_bytes(N, E):
    """(compulsory, executed, by family) HBM bytes of one layer step.  compulsory = SURVEY 8(d)'s 4.0 KB per edge.
    executed = what the implementation's kernels move when every operand is read / written once per kernel that touches
    it (itemised below per kernel family, DESIGN.md §4 compares each line with the rocprofv3 FETCH_SIZE / WRITE_SIZE
    counters: 41 GB here against 52.8 GB by counters at the benchmark shape -- the difference is the prepared T that
    every contraction workgroup streams from L2 / Infinity Cache, counted at the fabric, and the gathered reads of
    edge_gw, for which the gfx950 x2 correction of FETCH_SIZE is not calibrated)."""
    Z, NC, EC, NW = E * 1536 * 4.0, N * 128 * 4.0, E * 128 * 4.0, N * 1536 * 4.0
    S, T = N * 768 * 4.0, 3 * 8.4e6                     # per-node message sums; prepared T per contraction launch
    fam = {
        "edge_zx (per-edge forward: Z written)": (EC + EC + NW, Z + E * 12),
        "seg_softmax + seg_wsum (message half of Z read)": (Z / 2 + E * 24, S + E * 12),
        "edge_seg_bwd (Z read once; sign bits, Gi written)": (Z + S + E * 24, E * 192 + NW + E * 24),
        "edge_ge x3 (g_e, g_x)": (E * 192 + S + 2 * NW, EC + NC),
        "edge_gw x3 (dW_e, dW_i, dW_j)": (E * 128 * 6 + E * 192 + S + 2 * NW + N * 1024, 24e6),
        "edge_gj": (E * 192 + S, NW),
        "hypernet forward: 4 contractions + slabs + LayerNorm": (4 * (3 * NC + T) + 12 * NC, 12 * NC + 7 * NC),
        "hypernet backward: 4 fused contractions + finish": (4 * (4 * NC + T) + 24 * NC, 20 * NC + 8 * NC),
        "weight-gradient contraction + operand preparation": (24 * NC + 8 * NC + 4 * NC + 12 * NC, 4 * 8.4e6 + 8 * NC + 16 * NC),
        "trunk chains (8 launches)": (24 * NC, 40 * NC),
        "width-128 dense layers (22 launches)": (22 * NC, 22 * NC),
        "dense-layer weight gradients (2 batched launches)": (48 * NC + 12 * NC, 0.03e9),
        "LayerNorm backward, mix, column sums, maxima, T planes": (25 * NC, 5 * NC + 0.8e9),
    }
    compulsory = 4.0e3 * E
    executed = sum(r + w for r, w in fam.values())
    return compulsory, executed, {k: {"read_GB": round(r / 1e9, 2), "write_GB": round(w / 1e9, 2)} for k, (r, w) in fam.items()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)      # 20 x 25 ms: the clocks settle within the first steps
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--graphs", type=int, default=GRAPHS, help="crystals per rank (default: the 1M-edge batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exclusive-pass", action="store_true",
                    help="skip the 2-step serial pass after the timed region (exclusive durations of the kernels that "
                         "run concurrently); used for the rocprofv3 runs so that their averages cover the timed steps only")
    ap.add_argument("--no-extra-legs", action="store_true",
                    help="skip the untimed-for-`value` legs after the timed region (other arithmetic modes, full stack)")
    ap.add_argument("--scaling", choices=["weak", "strong"], default="weak", help="--workload train: per-rank or total batch fixed")
    ap.add_argument("--hipgraph", action="store_true",
                    help="layer / stack workloads: also capture the step into one hipGraph and time its replay (reported "
                         "as launch_bound.hipgraph; `value` stays the eager step); on by default below 512 crystals")
    ap.add_argument("--nbrs", type=int, default=None, help="neighbours per atom (default 12; --workload stress: 64)")
    ap.add_argument("--edge-storage", choices=["f32", "bf16", "bf16-mma"], default="f32",
                    help="storage of the per-edge intermediates Z / gZ (bf16 = the 'bf16 activations' of configs[4]; "
                         "tolerance 1e-2 instead of 1e-4: never the default, reported in the line)")
    ap.add_argument("--mode", choices=["f16x3c", "bf16x6", "f16x3", "f32"], default=None,
                    help="arithmetic of the matrix-core kernels (cgat_amd.set_bilinear_mode); default: the library's default, "
                         "f16x3c = 24-bit operands.  `value` is measured in this mode; the other modes are reported under "
                         "`modes`, each from a fresh process")
    ap.add_argument("--workload", choices=["layer", "stack", "collate", "optim", "train", "stress", "edge_hyper", "lightning"], default="layer",
                    help="layer: BASELINE metric (one GATConvNodes layer).  stack: informational, the full "
                         "CGAtNet(200,128,4,msg_heads=3) fwd+bwd of config 3 on the same 1M-edge batch")
    args = ap.parse_args()

    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.dist import init_from_env

    rank, world, device = init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (cuda device); there is no CPU path to measure")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")
    if args.mode is not None:
        P.set_bilinear_mode(args.mode)

    if args.workload == "collate":
        return bench_collate(args, rank, world, device)
    if args.workload == "optim":
        return bench_optim(args, rank, world, device)
    if args.workload == "edge_hyper":
        return bench_edge_hyper(args, rank, world, device)
    if args.workload == "train":
        return bench_train(args, rank, world, device)
    if args.workload == "lightning":
        return bench_lightning(args, rank, world, device)

    # CPU leg FIRST (rank 0, N = 1 only): the GPU leg that follows is then one contiguous block of device work
    cpu = None
    if world == 1 and not args.no_cpu_baseline and args.workload == "layer":
        cpu = cpu_baseline()

    P.set_edge_storage(args.edge_storage)
    stress = args.workload == "stress"
    K_used = args.nbrs or (64 if stress else K_NBR)
    if stress:
        # BASELINE configs[4]: 50 000 crystals x 20 atoms x 64 neighbours = 64 M edges through one layer, fwd + bwd, in
        # closed chunks of <= 8 M edges (cgat_amd/chunked.py) so that the per-edge workspace stays bounded
        if args.graphs == GRAPHS:
            args.graphs = 50000
        ops.set_validate_indices(False)
        step, N, E = make_layer_workload(args.graphs, rank, world, device, K=K_used)
    elif args.workload == "layer":
        step, N, E = make_layer_workload(args.graphs, rank, world, device, K=K_used)
    else:
        step, N, E = make_stack_workload(args.graphs, rank, world, device)
    layer_like = args.workload in ("layer", "stress")
    step_averager = getattr(step, "averager", None)

    for _ in range(args.warmup):
        step()
    _fence(world)
    # The CSR plan of the batch is built once per edge_index and cached (every layer of a stack, forward and backward,
    # shares it), so the timed steps below do not contain it; what one build costs -- kernels plus the two host
    # synchronisations of the index validation -- is reported beside ms_per_step.
    plan_build_ms = None
    if layer_like and not stress:
        ei_probe = make_inputs(args.graphs, rank, device, K_used)[0]
        reps = []
        for _ in range(4):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            ops.EdgePlan(ei_probe, N)
            torch.cuda.synchronize()
            reps.append(1e3 * (time.perf_counter() - t0))
        plan_build_ms = round(sorted(reps[1:])[1], 3)
        del ei_probe
    ops.prof_reset()
    ops.prof_enable(True)                                  # HIP events around the kernel launches, on their stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    _fence(world)
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    ar_report = _allreduce_report(step_averager, args.warmup + args.steps) if step_averager is not None else None
    launch_rep = None
    if not stress and (args.hipgraph or args.graphs < 512):
        launch_rep = _launch_report(step, args.steps, world, try_graph=step_averager is None)
    ALL_TAGS = ("bilinear_rows", "bilinear_dual", "bilinear_wgrad", "edge_z", "edge_proj", "edge_seg_bwd", "edge_ge",
                "edge_gw", "edge_gj", "rows_ge", "rows_gw", "linear128", "mlp_chain", "rows_dw", "gemm_f32")
    prof = {t: ops.prof_get(t) for t in ALL_TAGS}          # (launches, total ms) inside the timed region
    # With the weight-gradient contractions on the side stream they and the attention-backward kernels they run
    # beside share the chip, so their timed-region durations are not exclusive.  A short serial pass AFTER the timed
    # region (not part of `value`) gives those kernels' exclusive durations; both are reported.
    concurrent = ("bilinear_wgrad", "edge_seg_bwd", "edge_ge", "edge_gw", "edge_gj", "rows_ge", "rows_gw", "rows_dw") \
        if (ops.overlap_enabled() and not args.no_exclusive_pass) else ()
    prof_x = {}
    if concurrent:
        raw_overlap = ops.get_overlap_wgrad()
        ops.set_overlap_wgrad(False)
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(2):
            step()
        _fence(world)
        ops.prof_enable(False)
        prof_x = {t: ops.prof_get(t) for t in concurrent}
        ops.set_overlap_wgrad(raw_overlap)
    per_rank = None
    if world > 1:
        # every rank's own timed-region duration (gathered: the SCALE line is self-checking), `value` uses the maximum
        mine = torch.tensor([elapsed], device=device, dtype=torch.float64)
        every = [torch.zeros_like(mine) for _ in range(world)]
        dist.all_gather(every, mine)
        per_rank = [1e3 * float(x.item()) / args.steps for x in every]
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    # ---- extra legs (N = 1, after the timed region, never part of `value`): the same step in the two arithmetic modes
    # whose operands carry >= 24 significand bits, and the full 4-layer stack of BASELINE configs[2] ----
    mode = P.get_bilinear_mode()
    modes_ms, stack_ms, side_ms = {mode: 1e3 * elapsed / args.steps}, None, None
    if world == 1 and not args.no_extra_legs and args.workload == "layer" and K_used == K_NBR:
        del step
        torch.cuda.empty_cache()
        sstep, _, _ = make_stack_workload(args.graphs, rank, world, device)
        stack_ms = 1e3 * _time_steps(sstep, 1, 3, world) / 3
        del sstep
        torch.cuda.empty_cache()
        # the same timed region in the other arithmetic modes: a FRESH child process per mode (no process-global switch
        # behind this process' back), the same --steps / --warmup, no CPU leg, no further legs
        import subprocess
        # every child runs THIS line's configuration: the same edge storage and neighbour count
        same_cfg = ["--graphs", str(args.graphs), "--edge-storage", args.edge_storage] + \
                   (["--nbrs", str(args.nbrs)] if args.nbrs else []) + ["--workload", args.workload]
        for m in ("f16x3c", "bf16x6", "f16x3", "f32"):
            if m == mode or (m == "f32" and args.edge_storage != "f32"):      # (the f32 mode has no bf16 storage)
                continue
            k, w = (args.steps, args.warmup) if m != "f32" else (max(3, args.steps // 4), 2)
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", m, "--steps", str(k), "--warmup", str(w)] + same_cfg + \
                  ["--no-cpu-baseline", "--no-extra-legs", "--no-exclusive-pass"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=300)
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                modes_ms[m] = float(json.loads(line)["ms_per_step"])
            except Exception as ex:                        # a failed leg must not take the headline down with it
                modes_ms[m] = None
                sys.stderr.write(f"bench.py: mode leg {m} failed: {ex}\n")
        # the same timed region in THIS mode with the weight-gradient launch on the side stream (opt-in in the 24-bit
        # modes, cgat_amd/ops.py: it halves the chip for the dominant kernel's launch) -- informational, never `value`
        if not ops.overlap_enabled():
            cmd = [sys.executable, os.path.abspath(__file__), "--mode", mode, "--steps", str(args.steps), "--warmup",
                   str(args.warmup)] + same_cfg + ["--no-cpu-baseline", "--no-extra-legs", "--no-exclusive-pass"]
            try:
                r = subprocess.run(cmd, capture_output=True, text=True, timeout=180,
                                   env=dict(os.environ, CGAT_OVERLAP_WGRAD="1"))
                line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")][-1]
                side_ms = float(json.loads(line)["ms_per_step"])
            except Exception as ex:
                sys.stderr.write(f"bench.py: side-stream leg failed: {ex}\n")

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        # The three hypernetwork contraction kernels execute 2*N*C^3 flop per launch and predicted layer each (SURVEY 8a
        # a8.2: C*(C*C)*2 per row; the fused backward kernel produces two gradients from ONE such contraction, so it is
        # priced at what it executes, not at the two contractions the reference's autograd performs).
        flops_per_layer = 2.0 * N * C_FEA ** 3
        kernels = {"bilinear_wgrad": "bilinear_wgrad128_f16p_kernel", "bilinear_dual": "bilinear_rows128_dual_kernel",
                   "bilinear_rows": "bilinear_rows128_ring16_kernel"}
        # contractions of each kind per STEP (4 predicted layers; in closed chunks the forward runs twice: once in the
        # forward pass, once as the backward's per-chunk recomputation).  Flop per LAUNCH = flop per step / launches per
        # step, whatever the rows of a launch are (a chunk's share, one layer or the four batched ones): round 4 multiplied
        # a per-chunk launch by the whole batch's rows and printed a fraction of 4.3
        # (closed chunks since round 6: only the aggregate half is recomputed in backward -- cgat_amd/chunked.py
        # ChunkedSplitLayerFn -- so the forward contraction runs once per step there too; CGAT_CHUNK_SPLIT=0: twice)
        contractions_per_step = {"bilinear_wgrad": 4, "bilinear_dual": 4,
                                 "bilinear_rows": 8 if (stress and os.environ.get("CGAT_CHUNK_SPLIT", "1") == "0") else 4}
        kpasses = {}                                       # matrix-core pass-equivalents per product, per kernel
        if mode == "f32":
            kernels = {"bilinear_wgrad": "bilinear_wgrad128_kernel", "bilinear_rows": "bilinear_rows128_kernel"}
            peak, passes = MFMA_F32_PEAK_TFLOPS, 1
            note = "f32-input MFMA (v_mfma_f32_32x32x2_f32), exact fp32"
        elif mode == "f16x3":
            passes = 3
            peak = MFMA_BF16_PEAK_TFLOPS / passes          # dense fp16 matrix peak = the bf16 one (2500)
            note = ("fp32 operands scaled by a power of two (per row / per tensor) and split into 2 fp16 pieces (22 bits), "
                    "3 v_mfma_f32_*_f16 passes per product, fp32 accumulate (measured at the error of an fp32 "
                    f"product chain): executed MFMA flop = 3 x algorithmic, so the roof for ALGORITHMIC flop is the dense "
                    f"fp16 peak {MFMA_BF16_PEAK_TFLOPS:.0f} / 3; the f32-input MFMA roof would be {MFMA_F32_PEAK_TFLOPS}")
        elif mode == "f16x3c":
            # 24-bit operands: three fp16 passes + three 6-bit passes that run K = 128 in the cycles of one fp16 pass of
            # K = 32 (4 x the rate): 3 + 3/4 = 3.75 pass-equivalents of matrix time per product in the kernels that have
            # the form; the others run their six-pass bf16 form
            kernels["bilinear_rows"] = "bilinear_rows128_ring16c_kernel"
            kernels["bilinear_dual"] = "bilinear_rows128_dualc_kernel"
            kernels["bilinear_wgrad"] = "bilinear_wgrad128_f16c_kernel"      # round 5: one launch for the four layers
            F16C = getattr(P.ops, "F16C_KERNELS", ("bilinear_rows", "bilinear_dual"))
            kpasses = {t: (3.75 if t in F16C else 6) for t in kernels}
            if "bilinear_wgrad" not in F16C:
                kernels["bilinear_wgrad"] = "bilinear_wgrad128_bf16_kernel"
            passes = 3.75
            peak = MFMA_BF16_PEAK_TFLOPS / passes
            note = ("24-bit operands: every fp32 operand scaled by a power of two and split EXACTLY into h + l + t (two fp16 "
                    "pieces + the 24th bit); a product = hh + hl + lh as 3 v_mfma_f32_16x16x32_f16 passes (exact) + ll + ht + "
                    "th (weight <= 2^-22) as 3 v_mfma_f32_16x16x128_f8f6f4 passes on 6-bit images (4 x the fp16 rate per "
                    "flop), fp32 accumulate: executed matrix time = 3.75 fp16-pass-equivalents per product in the kernels "
                    f"with this form (roof for ALGORITHMIC flop = {MFMA_BF16_PEAK_TFLOPS:.0f} / 3.75), 6 bf16 passes "
                    f"({MFMA_BF16_PEAK_TFLOPS:.0f} / 6) in the others -- each kernel's `peak` says which; measured error vs "
                    "fp64 at or below the six-pass bf16 split's and the f32-input MFMA's (tools/f16x3c_probe.hip)")
        else:
            kernels["bilinear_wgrad"] = "bilinear_wgrad128_bf16_kernel"
            passes = 6 if mode == "bf16x6" else 3
            peak = MFMA_BF16_PEAK_TFLOPS / passes
            note = (f"fp32 operands split into 3 bf16 pieces, {passes} v_mfma_f32_*_bf16 passes per product, fp32 "
                    f"accumulate (measured fp32-equivalent accuracy): executed MFMA flop = {passes} x algorithmic, so the "
                    f"roof for ALGORITHMIC flop is the dense bf16 peak {MFMA_BF16_PEAK_TFLOPS:.0f} / {passes}; the "
                    f"f32-input MFMA roof would be {MFMA_F32_PEAK_TFLOPS}")
        # NOT measured in this run: HBM bytes per launch and matrix-core busy cycles come from separate rocprofv3 --pmc
        # passes of this same command, collected by the builder and committed under profiles/; the line says so
        # They are only replayed when they were collected on THIS source tree in THIS arithmetic mode (tools/tree_id.py: the
        # stamp carries the sha of the sources the profiled build was made from); otherwise the fields are null and
        # `stale_counters` says what was found
        traffic, traffic_src, stale = _counters_of_this_build(os.path.join(ROOT, "profiles", "pmc_contraction_kernels.json"), mode)
        per_kernel, roof = {}, None
        for tag, kname in kernels.items():
            n_t, ms_t = prof[tag]
            if not n_t:
                continue
            avg_ms = ms_t / n_t
            fl = flops_per_layer * contractions_per_step[tag] / (n_t / args.steps)
            ach = fl / (avg_ms * 1e-3) / 1e12
            kpeak = MFMA_BF16_PEAK_TFLOPS / kpasses[tag] if tag in kpasses else peak
            per_kernel[tag] = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": round(kpeak, 1),
                               "unit": "TFLOP/s", "frac": round(ach / kpeak, 4),
                               "matrix_passes_per_product": kpasses.get(tag, passes),
                               # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled
                               # per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE); see profiles/
                               "traffic": traffic.get(kname, {}).get("hbm_bytes_per_launch"),
                               "traffic_source": traffic_src,
                               "launches_per_step": n_t / args.steps, "avg_launch_ms": round(avg_ms, 4),
                               "ms_per_step": round(ms_t / args.steps, 3), "flops_per_launch": fl}
        for tag in per_kernel:
            if tag in concurrent and prof_x.get(tag, (0, 0))[0]:
                x_ms = prof_x[tag][1] / prof_x[tag][0]
                x_ach = per_kernel[tag]["flops_per_launch"] / (x_ms * 1e-3) / 1e12
                peak_t = per_kernel[tag]["peak"]
                per_kernel[tag]["concurrent"] = ("runs on the side stream on half of the CUs beside the HBM-bound attention "
                                                 "backward: the timed-region duration is shared; `exclusive` = the same "
                                                 "kernel alone on the chip, from a serial pass after the timed region")
                per_kernel[tag]["exclusive"] = {"avg_launch_ms": round(x_ms, 4), "achieved": round(x_ach, 2),
                                                "frac": round(x_ach / peak_t, 4)}
        if per_kernel:
            # the roofline object is for the kernel with the largest GPU time per step INSIDE the timed region, whatever
            # stream it ran on; its `frac` is the in-region one (side-stream kernels also carry their exclusive numbers)
            dom = max(per_kernel, key=lambda t: per_kernel[t]["ms_per_step"])
            roof = dict(per_kernel[dom])
            roof["arithmetic"] = note
            roof["other_contraction_kernels"] = {t: {k: v[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms",
                                                                        "ms_per_step", "traffic", "concurrent", "exclusive")
                                                     if k in v}
                                                 for t, v in per_kernel.items() if t != dom}
            # SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x 256 CUs x clock x time)
            cnt, cnt_src, cnt_stale = _counters_of_this_build(os.path.join(ROOT, "profiles", "mfma_counters.json"), mode)
            if cnt:
                cnt["source"] = cnt_src
                roof["mfma_utilisation_from_counters"] = cnt
            if stale or cnt_stale:
                roof["stale_counters"] = "; ".join(x for x in (stale, cnt_stale) if x)
        # HBM side (the north_star's "fraction of the HBM roofline"): the four per-edge kernels are bound by the
        # Z-sized passes.  Algorithmic bytes per launch with W2 = 2*H*Hd = 1536 fp32 columns per edge:
        W2b = 2 * HEADS * 256 * (2 if args.edge_storage in ("bf16", "bf16-mma") else 4)
        # f16x3: the per-edge forward kernel computes the x_j projection itself (edge_zx_kernel): Z written, e and
        # x[src] rows read, Pi rows once; other modes: Z written, Pj gathered (W2b per edge), e read, Pi rows once
        W2f = 2 * HEADS * 256 * 4                          # per-node rows (Pi, Gi, gS) are fp32 in either storage mode
        ez_bytes = E * (W2b + 2 * C_FEA * 4) + N * W2f if mode == "f16x3" else E * (W2b + W2f + C_FEA * 4) + N * W2f
        # Backward (split-arithmetic modes at these widths): gZ [E, W2] is never stored -- edge_seg_bwd leaves one bit per
        # element (W2 / 8 bytes per edge) and its consumers rebuild the rows from the per-node matrix gS (csrc/kernels.h
        # struct EdgeRC), so only edge_seg_bwd is still bound by a Z-sized pass; edge_ge / edge_gw are priced against the
        # matrix roof below.  f32 mode: the generic GEMM route with gZ stored (tags edge_ge / edge_gw do not occur).
        W2cols = 2 * HEADS * 256
        hbm_alg = {"edge_z": ez_bytes,
                   "edge_seg_bwd": E * (W2b + W2cols // 8) + N * (W2f + W2f // 2)}   # Z read, sign bits written, Gi written, gS read
        if stress:
            hbm_alg["edge_z"] *= 2                          # forward + the backward's per-chunk recomputation
        hbm = {}
        for tag, nbytes in hbm_alg.items():
            n_t, ms_t = prof[tag]
            if n_t:
                gbs = nbytes / (ms_t / args.steps * 1e-3) / 1e9      # bytes of the step / the tag's time in the step
                # counter bytes of one step for the tag (builder-collected rocprofv3 --pmc passes, profiles/): the largest
                # launch of each kernel name = the per-edge launch at the headline batch; only valid for that batch
                kn = {"edge_z": ("edge_zx_kernel",) if mode == "f16x3" else
                                ("edge_z6w_kernel",) if mode in ("f16x3c", "bf16x6") else ("edge_z_kernel",),
                      "edge_seg_bwd": ("edge_seg_bwd_kernel",) if mode == "f16x3" else
                                      ("seg_bwd_msg_kernel", "seg_bwd_soft_kernel", "seg_bwd_att_kernel")}[tag]
                tb = traffic.get("hbm_bound_kernels", {}) if (args.workload == "layer" and args.graphs == GRAPHS and
                                                                args.edge_storage == "f32") else {}
                cbytes = sum(tb[k]["hbm_bytes_largest_launch"] for k in kn) if all(k in tb for k in kn) else None
                csrc = traffic_src
                per_launch = False
                if stress and args.graphs == 50000 and K_used == 64:
                    # the 64 M-edge step runs closed chunks of equal size: counter bytes of ONE chunk's launch of each
                    # kernel (profiles/pmc_stress_kernels.json, keyed by edge storage) against that launch's mean duration
                    st_all, st_src, st_stale = _counters_of_this_build(
                        os.path.join(ROOT, "profiles", "pmc_stress_kernels.json"), mode,
                        key={"f32": "stress", "bf16": "stress_bf16", "bf16-mma": "stress_bf16mma"}[args.edge_storage])
                    sb = st_all.get("hbm_bytes_largest_launch", {})
                    if all(k in sb for k in kn):
                        cbytes, csrc, per_launch = sum(sb[k] for k in kn), st_src, True
                hbm[tag] = {"bound": "hbm", "kernel": " + ".join(kn),
                            "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(gbs / 8000.0, 4),
                            "what": "achieved / frac: ALGORITHMIC bytes over the tag's time; traffic: HBM bytes from the PMC "
                                    "counters (2 x FETCH_SIZE + WRITE_SIZE); frac_from_counters: those over the same time -- "
                                    "gathers of per-node rows that hit in L2 are algorithmic bytes, not HBM bytes",
                            "traffic": cbytes, "traffic_source": csrc if cbytes else None,
                            # headline batch: one launch of each kernel per step -> bytes of the step over the tag's time in
                            # the step; stress: one chunk's launches over the mean duration of one launch of each kernel
                            # (a tag counts timed SCOPES -- one per group of per-edge launches, e.g. the three segment-backward
                            # kernels are one scope -- and the counter bytes are the sum over the group's kernels)
                            "frac_from_counters": (round(cbytes / ((ms_t / n_t if per_launch else ms_t / args.steps)
                                                                   * 1e-3) / 8e12, 4)
                                                   if cbytes and (per_launch or n_t / args.steps == 1) else None),
                            "launches_per_step": n_t / args.steps,
                            "avg_launch_ms": round(ms_t / n_t, 4), "ms_per_step": round(ms_t / args.steps, 3),
                            "algorithmic_bytes_per_launch": int(nbytes / (n_t / args.steps))}
                if tag in concurrent and prof_x.get(tag, (0, 0))[0]:
                    x_ms = prof_x[tag][1] / prof_x[tag][0]
                    x_gbs = nbytes / (n_t / args.steps) / (x_ms * 1e-3) / 1e9
                    hbm[tag]["concurrent"] = "shares the chip with the side-stream contractions in the timed region"
                    hbm[tag]["exclusive"] = {"avg_launch_ms": round(x_ms, 4), "achieved": round(x_gbs, 1),
                                             "frac": round(x_gbs / 8000.0, 4)}
        # the two per-edge products over the rebuilt gZ rows: 2 * E * W2 * 128 flop each, against the same matrix roof
        edge_mfma = {}
        if mode == "f16x3c":
            peak_e = MFMA_BF16_PEAK_TFLOPS / 6             # the per-edge products run their six-pass bf16 form in this mode
        else:
            peak_e = peak
        if mode != "f32":
            for tag, kname in (("edge_ge", "edge_ge_kernel"), ("edge_gw", "edge_gw_kernel")):
                n_t, ms_t = prof[tag]
                if n_t:
                    fl = 2.0 * E * W2cols * C_FEA / (n_t / args.steps)          # one launch per chunk of the step
                    ach = fl / (ms_t / n_t * 1e-3) / 1e12
                    edge_mfma[tag] = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": round(peak_e, 1),
                                      "unit": "TFLOP/s", "frac": round(ach / peak_e, 4), "launches_per_step": n_t / args.steps,
                                      "avg_launch_ms": round(ms_t / n_t, 4), "ms_per_step": round(ms_t / args.steps, 3),
                                      "flops_per_launch": fl}
                    if tag in concurrent and prof_x.get(tag, (0, 0))[0]:
                        x_ms = prof_x[tag][1] / prof_x[tag][0]
                        edge_mfma[tag]["exclusive"] = {"avg_launch_ms": round(x_ms, 4),
                                                       "achieved": round(fl / (x_ms * 1e-3) / 1e12, 2),
                                                       "frac": round(fl / (x_ms * 1e-3) / 1e12 / peak_e, 4)}
        shares = {}
        for tag in ALL_TAGS:
            n_t, ms_t = prof[tag]
            if n_t:
                shares[tag] = {"launches_per_step": n_t / args.steps, "ms_per_step": round(ms_t / args.steps, 3)}
                if tag in concurrent:
                    shares[tag]["concurrent"] = True
        if stress and hbm:
            # K = 64: the hypernetwork (per atom) is 1/64 of an edge's share, the step is the HBM-bound edge phase
            top = max(hbm, key=lambda t: hbm[t]["ms_per_step"])
            if roof is None or hbm[top]["ms_per_step"] > roof["ms_per_step"]:
                contr = roof
                roof = dict(hbm[top])
                roof["other_kernels"] = {t: v for t, v in hbm.items() if t != top}
                if contr is not None:
                    roof["contraction_kernels"] = {k: contr[k] for k in ("kernel", "achieved", "frac", "ms_per_step")}
        metric = ("edges/sec through one CGAT attention layer (fwd+bwd), 1M-edge batch" if args.workload == "layer"
                  else "edges/sec through one CGAT attention layer (fwd+bwd), 64M-edge large-neighbour batch in closed "
                       "chunks [BASELINE configs[4]]" if stress
                  else "batch-edges/sec through the full CGAT stack (4 layers, fwd+bwd), 1M-edge batch [informational]")
        dtype = {"f16x3c": "f32 storage / f16x3c (24-bit operands: exact h + l + t split, 3 fp16 + 3 six-bit matrix passes "
                           "where built, six bf16 passes elsewhere; fp32 accumulate)",
                 "f16x3": "f32 storage / f16x3 split (22-bit operands, fp32 accumulate)",
                 "bf16x6": "f32 storage / bf16x6 split (24-bit operands, fp32 accumulate)",
                 "bf16x3": "f32 storage / bf16x3 split (16-bit products; diagnostic mode)",
                 "f32": "f32 (f32-input MFMA)"}[mode]
        out = {
            "metric": metric,
            "value": world * E * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "plan_build_ms": plan_build_ms,
            "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": dtype + ("; Z / gZ of the edge phase stored as bf16 (tolerance 1e-2)"
                                                   if args.edge_storage == "bf16" else
                                                   "; Z of the edge phase stored as bf16 and the two per-edge backward products "
                                                   "on bf16 operands (one matrix pass, fp32 accumulate; tolerance 1e-2)"
                                                   if args.edge_storage == "bf16-mma" else ""),
            "data": "synthetic", "bilinear_mode": mode, "edge_storage": args.edge_storage,
            "config": {"workload": (f"one GATConvNodes layer (H_Net update) fwd+bwd, {args.graphs} crystals x {ATOMS} atoms x "
                                    f"{K_used} nbrs per rank: N={N}, E={E}, C=Ce={C_FEA}, H={HEADS}, scalar attention" +
                                    (f", closed chunks of <= {P.chunked.max_edges_per_pass()} edges with per-chunk "
                                     "recomputation of the per-edge attention in backward (2 attention forward + 1 hypernetwork forward + 1 "
                                     "backward pass per step)" if stress else ""))
                       if layer_like else
                       (f"CGAtNet(200,128,4,msg_heads=3,update_edges=True) fwd+bwd of L1 loss, {args.graphs} crystals: "
                        f"N={N}, E={E}"),
                       "edges_per_rank": E, "parallelism": f"dp{world} (graphs sharded, gradient all-reduce)"},
            "roofline": roof, "hbm_bound_kernels": hbm if layer_like else None,
            "edge_product_kernels": edge_mfma if layer_like else None,
            "kernel_ms_per_step": shares,
        }
        if args.workload == "layer":
            fl = layer_step_flops(N, E)
            comp, execd, by_family = layer_step_bytes(N, E)
            out["step_roofline"] = {
                "what": "the whole step against both roofs: executed-algorithmic flop (each product once) over the step time "
                        "vs the matrix roof of the arithmetic mode, and HBM bytes over the step time vs 8 TB/s",
                "flop_per_step": fl, "tflops": round(fl / (ms * 1e-3) / 1e12, 1), "tflops_peak": round(peak, 1),
                "tflops_frac": round(fl / (ms * 1e-3) / 1e12 / peak, 4),
                "hbm_bytes_compulsory": int(comp), "hbm_frac_compulsory": round(comp / (ms * 1e-3) / 8e12, 4),
                "hbm_bytes_executed": int(execd), "hbm_frac": round(execd / (ms * 1e-3) / 8e12, 4),
                "hbm_bytes_executed_by_kernel_family": by_family}
            out["modes"] = {"what": "ms per step of the same timed region in each arithmetic mode: `value` is the mode named in "
                                    "bilinear_mode (this process); every other mode is a fresh child process running "
                                    "`bench.py --mode M` with the same --steps / --warmup (f32: a quarter of the steps). "
                                    "f16x3c and bf16x6 carry 24-bit operands, f16x3 22-bit, f32 is the f32-input MFMA",
                            **{m: (round(v, 3) if v is not None else None) for m, v in modes_ms.items()},
                            "edges_per_s": {m: (round(E / (v * 1e-3), 1) if v else None) for m, v in modes_ms.items()}}
            if side_ms is not None:
                out["side_stream"] = {"ms_per_step": round(side_ms, 3), "edges_per_s": round(E / (side_ms * 1e-3), 1),
                                      "what": "the same timed region, same arithmetic, fresh child process with "
                                              "CGAT_OVERLAP_WGRAD=1: the batched weight-gradient launch on a side stream on "
                                              "half of the chip beside the HBM-bound attention backward.  Not `value`: in "
                                              "that order the dominant kernel's launch duration is shared, and `roofline` "
                                              "would not be a statement about the kernel"}
            if stack_ms is not None:
                out["stack_fwd_bwd_ms"] = {"ms_per_step": round(stack_ms, 2), "batch_edges_per_s": round(E / (stack_ms * 1e-3), 1),
                                           "edge_layer_passes_per_s": round(4 * E / (stack_ms * 1e-3), 1),
                                           "what": "BASELINE configs[2]: CGAtNet(200,128,4,msg_heads=3) fwd+bwd of the L1 loss "
                                                   "on the same 1M-edge batch, default arithmetic mode (3 steps after 1 warm-up)"}
        if cpu is not None:
            out["cpu_baseline"] = cpu
        if dist.is_initialized():
            out["ranks"] = {"what": "one process per GPU; ms_per_step of every rank's own timed region (value = edges of all "
                                    "ranks / the maximum), the communicator the gradient all-reduce ran over",
                            "world_size": dist.get_world_size(), "backend": dist.get_backend(),
                            "collective_library": "RCCL (torch.distributed backend 'nccl' on ROCm)"
                            if dist.get_backend() == "nccl" else dist.get_backend(),
                            "ms_per_step_per_rank": [round(x, 3) for x in per_rank] if per_rank else [round(ms, 3)],
                            "devices_visible_to_rank0": torch.cuda.device_count()}
        if ar_report is not None:
            out["allreduce"] = ar_report
        if launch_rep is not None:
            out["launch_bound"] = launch_rep
        print(json.dumps(out), flush=True)
    if dist.is_initialized():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
