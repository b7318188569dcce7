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


def make_inputs(graphs, seed, device):
    import cgat_amd as P
    b, _ = P.synthetic_batch(graphs, ATOMS, K_NBR, seed=seed)
    g = torch.Generator().manual_seed(1000 + seed)
    N, E = b.num_nodes, b.edge_index.shape[1]
    x = torch.randn(N, C_FEA, generator=g).to(device)
    e = torch.randn(E, C_FEA, generator=g).to(device)
    x0 = torch.randn(N, C_FEA, generator=g).to(device)
    cot = torch.randn(N, C_FEA, generator=g).to(device)
    return b.edge_index.to(device), x, e, x0, cot


def cpu_baseline(n_graphs=100, reps=3):
    """The oracle (op-for-op restatement of the reference's CPU path: cat -> head repeat -> grouped
    Conv1d -> LeakyReLU -> conv -> segment softmax -> scatter-add -> Linear(C -> C*C+C) hypernet ->
    bmm -> LayerNorm -> tanh) timed on this box's host cores, same layer, fwd+bwd, on a bounded
    sample of the same synthetic workload."""
    import cgat_amd as P
    from oracle import cgat_oracle as O
    torch.manual_seed(1)
    layer = O.GATConvNodes(C_FEA, C_FEA, C_FEA, HEADS, concat=True)
    ei, x, e, x0, cot = make_inputs(n_graphs, 0, "cpu")
    times = []
    for r in range(reps + 1):
        xx, ee, xx0 = (t.clone().requires_grad_(True) for t in (x, e, x0))
        t0 = time.perf_counter()
        y = layer(xx, ei, ee, xx0)
        torch.autograd.grad((y * cot).sum(), [xx, ee, xx0] + list(layer.parameters()))
        dt = time.perf_counter() - t0
        if r > 0:
            times.append(dt)
    times.sort()
    med = times[len(times) // 2]
    E = ei.shape[1]
    return {"value": E / med, "unit": "edges/s", "cores": torch.get_num_threads(), "kind": "port",
            "sample": f"{n_graphs} crystals ({E} edges) of the same synthetic workload, same layer fwd+bwd, "
                      f"fp32, median of {reps} after 1 warm-up ({med:.2f} s per pass); per-edge cost is "
                      "batch-size independent (SURVEY §8d)"}


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


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--graphs", type=int, default=GRAPHS, help="crystals per rank (default: the 1M-edge batch)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-exclusive-pass", action="store_true",
                    help="skip the 2-step serial pass after the timed region (exclusive durations of the kernels that "
                         "run concurrently); used for the rocprofv3 runs so that their averages cover the timed steps only")
    ap.add_argument("--workload", choices=["layer", "stack", "collate", "optim"], default="layer",
                    help="layer: BASELINE metric (one GATConvNodes layer).  stack: informational, the full "
                         "CGAtNet(200,128,4,msg_heads=3) fwd+bwd of config 3 on the same 1M-edge batch")
    args = ap.parse_args()

    import torch.distributed as dist
    import cgat_amd as P
    from cgat_amd import ops
    from cgat_amd.dist import GradientAverager, init_from_env

    rank, world, device = init_from_env()
    if device.type != "cuda":
        raise SystemExit("bench.py needs an MI355X (cuda device); there is no CPU path to measure")
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}: launch with torch.distributed.run")

    if args.workload == "collate":
        return bench_collate(args, rank, world, device)
    if args.workload == "optim":
        return bench_optim(args, rank, world, device)
    torch.manual_seed(1)                                   # identical parameters on every rank
    if args.workload == "layer":
        layer = P.GATConvNodes(C_FEA, C_FEA, C_FEA, HEADS, concat=True).to(device)
        params = list(layer.parameters())
        ei, x, e, x0, cot = make_inputs(args.graphs, rank, device)     # each rank: its own crystals
        N, E = x.shape[0], ei.shape[1]
        x.requires_grad_(True); e.requires_grad_(True); x0.requires_grad_(True)
        averager = GradientAverager(params) if world > 1 else None

        def step():
            for p in params:
                p.grad = None
            x.grad = e.grad = x0.grad = None
            y = layer(x, ei, e, x0)
            y.backward(cot)
            if averager is not None:
                averager.finish()
    else:
        net = P.CGAtNet(200, C_FEA, 4, msg_heads=HEADS, neighbor_number=K_NBR, update_edges=True).to(device)
        params = list(net.parameters())
        b, roost = P.synthetic_batch(args.graphs, ATOMS, K_NBR, seed=rank)
        b = b.to(device)
        roost = tuple(t.to(device) for t in roost)
        N, E = b.num_nodes, b.edge_index.shape[1]
        averager = GradientAverager(params) if world > 1 else None

        def step():
            for p in params:
                p.grad = None
            out = net(b, roost)
            loss = (out[:, 0] - b.y).abs().mean()          # L1 on the prediction column, as the harness' default
            loss.backward()
            if averager is not None:
                averager.finish()

    def fence():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    fence()
    ops.prof_reset()
    ops.prof_enable(True)                                  # HIP events around the kernel launches, on their stream
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    fence()
    elapsed = time.perf_counter() - t0
    ops.prof_enable(False)
    ALL_TAGS = ("bilinear_rows", "bilinear_dual", "bilinear_wgrad", "edge_z", "edge_proj", "edge_seg_bwd", "edge_ge",
                "edge_gw", "rows_ge", "rows_gw", "linear128", "rows_dw", "gemm_f32")
    prof = {t: ops.prof_get(t) for t in ALL_TAGS}          # (launches, total ms) inside the timed region
    # With the weight-gradient contractions on the side stream (the default), they and the attention-backward kernels
    # they run beside share the chip, so their timed-region durations are not exclusive.  A short serial pass AFTER the
    # timed region (not part of `value`) gives those kernels' exclusive durations; both are reported.
    concurrent = ("bilinear_wgrad", "edge_seg_bwd", "edge_ge", "edge_gw", "rows_ge", "rows_gw") \
        if (ops.overlap_enabled() and not args.no_exclusive_pass) else ()
    prof_x = {}
    if concurrent:
        ops.set_overlap_wgrad(False)
        ops.prof_reset()
        ops.prof_enable(True)
        for _ in range(2):
            step()
        fence()
        ops.prof_enable(False)
        prof_x = {t: ops.prof_get(t) for t in concurrent}
        ops.set_overlap_wgrad(True)
    if world > 1:
        t = torch.tensor([elapsed], device=device, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    if rank == 0:
        ms = 1e3 * elapsed / args.steps
        # The three hypernetwork contraction kernels execute 2*N*C^3 flop per launch each (C*(C*C)*2
        # per row; the fused backward kernel produces two gradients from ONE such contraction, so it is priced at
        # what it executes, not at the two contractions the reference's autograd performs).  The roofline object is
        # for the one with the largest share of the step.
        flops_per_launch = 2.0 * N * C_FEA ** 3
        mode = P.get_bilinear_mode()
        kernels = {"bilinear_wgrad": "bilinear_wgrad128_bf16_kernel", "bilinear_dual": "bilinear_rows128_dual_kernel",
                   "bilinear_rows": "bilinear_rows128_ring16_kernel"}
        if mode == "f32":
            kernels = {"bilinear_wgrad": "bilinear_wgrad128_kernel", "bilinear_rows": "bilinear_rows128_kernel"}
            peak = MFMA_F32_PEAK_TFLOPS
            note = "f32-input MFMA (v_mfma_f32_32x32x2_f32), exact fp32"
        elif mode == "f16x3":
            passes = 3
            peak = MFMA_BF16_PEAK_TFLOPS / passes          # dense fp16 matrix peak = the bf16 one (2500)
            note = ("fp32 operands scaled by a power of two (per row / per tensor) and split into 2 fp16 pieces (22 bits), "
                    "3 v_mfma_f32_16x16x32_f16 passes per product, fp32 accumulate (measured at the error of an fp32 "
                    f"product chain): executed MFMA flop = 3 x algorithmic, so the roof for ALGORITHMIC flop is the dense "
                    f"fp16 peak {MFMA_BF16_PEAK_TFLOPS:.0f} / 3; the f32-input MFMA roof would be {MFMA_F32_PEAK_TFLOPS}")
        else:
            passes = 6 if mode == "bf16x6" else 3
            peak = MFMA_BF16_PEAK_TFLOPS / passes
            note = (f"fp32 operands split into 3 bf16 pieces, {passes} v_mfma_f32_16x16x32_bf16 passes per product, fp32 "
                    f"accumulate (measured fp32-equivalent accuracy): executed MFMA flop = {passes} x algorithmic, so the "
                    f"roof for ALGORITHMIC flop is the dense bf16 peak {MFMA_BF16_PEAK_TFLOPS:.0f} / {passes}; the "
                    f"f32-input MFMA roof would be {MFMA_F32_PEAK_TFLOPS}")
        traffic_file = os.path.join(ROOT, "profiles", "pmc_contraction_kernels.json")
        traffic = json.load(open(traffic_file)) if os.path.exists(traffic_file) else {}
        per_kernel, roof = {}, None
        for tag, kname in kernels.items():
            n_t, ms_t = prof[tag]
            if not n_t:
                continue
            avg_ms = ms_t / n_t
            ach = flops_per_launch / (avg_ms * 1e-3) / 1e12
            per_kernel[tag] = {"bound": "mfma", "kernel": kname, "achieved": round(ach, 2), "peak": round(peak, 1),
                               "unit": "TFLOP/s", "frac": round(ach / peak, 4),
                               # HBM bytes per launch from rocprofv3 PMC passes of this same command (FETCH_SIZE doubled
                               # per the gfx950 correction of MI355X_MICROARCH.md, + WRITE_SIZE); see profiles/
                               "traffic": traffic.get(kname, {}).get("hbm_bytes_per_launch"),
                               "launches_per_step": n_t / args.steps, "avg_launch_ms": round(avg_ms, 4),
                               "ms_per_step": round(ms_t / args.steps, 3), "flops_per_launch": flops_per_launch}
        for tag in per_kernel:
            if tag in concurrent and prof_x.get(tag, (0, 0))[0]:
                x_ms = prof_x[tag][1] / prof_x[tag][0]
                x_ach = flops_per_launch / (x_ms * 1e-3) / 1e12
                per_kernel[tag]["concurrent"] = ("runs on the side stream on half of the CUs beside the HBM-bound attention "
                                                 "backward: the timed-region duration is shared; `exclusive` = the same "
                                                 "kernel alone on the chip, from a serial pass after the timed region")
                per_kernel[tag]["exclusive"] = {"avg_launch_ms": round(x_ms, 4), "achieved": round(x_ach, 2),
                                                "frac": round(x_ach / peak, 4)}
        if per_kernel:
            # the roofline object is for the contraction kernel with the largest share of the step's critical path
            # (kernels that run concurrently on the side stream are listed beside it with both durations)
            cands = [t for t in per_kernel if "concurrent" not in per_kernel[t]] or list(per_kernel)
            dom = max(cands, key=lambda t: per_kernel[t]["ms_per_step"])
            roof = dict(per_kernel[dom])
            roof["arithmetic"] = note
            roof["other_contraction_kernels"] = {t: {k: v[k] for k in ("kernel", "achieved", "frac", "avg_launch_ms",
                                                                        "ms_per_step", "traffic", "concurrent", "exclusive")
                                                     if k in v}
                                                 for t, v in per_kernel.items() if t != dom}
        # HBM side (the north_star's "fraction of the HBM roofline"): the four per-edge kernels are bound by the
        # Z-sized passes.  Algorithmic bytes per launch with W2 = 2*H*Hd = 1536 fp32 columns per edge:
        W2b = 2 * HEADS * 256 * 4
        # f16x3: the per-edge forward kernel computes the x_j projection itself (edge_zx_kernel): Z written, e and
        # x[src] rows read, Pi rows once; other modes: Z written, Pj gathered (W2b per edge), e read, Pi rows once
        ez_bytes = E * (W2b + 2 * C_FEA * 4) + N * W2b if mode == "f16x3" else E * (2 * W2b + C_FEA * 4) + N * W2b
        hbm_alg = {"edge_z": ez_bytes,
                   "edge_seg_bwd": E * 2 * W2b + N * (W2b + W2b // 2),       # Z read, gZ written, Gi written, gS read
                   "edge_ge": E * (W2b + C_FEA * 4),                         # gZ read, g_e written
                   "edge_gw": E * (W2b + C_FEA * (4 if mode == "f16x3" else 6))}   # gZ read, fp16x2 / bf16x3 planes of e read
        hbm = {}
        for tag, nbytes in hbm_alg.items():
            n_t, ms_t = prof[tag]
            if n_t:
                gbs = nbytes / (ms_t / n_t * 1e-3) / 1e9
                hbm[tag] = {"bound": "hbm", "achieved": round(gbs, 1), "peak": 8000.0, "unit": "GB/s",
                            "frac": round(gbs / 8000.0, 4), "avg_launch_ms": round(ms_t / n_t, 4),
                            "algorithmic_bytes_per_launch": int(nbytes)}
                if tag in concurrent and prof_x.get(tag, (0, 0))[0]:
                    x_ms = prof_x[tag][1] / prof_x[tag][0]
                    x_gbs = nbytes / (x_ms * 1e-3) / 1e9
                    hbm[tag]["concurrent"] = "shares the chip with the side-stream contractions in the timed region"
                    hbm[tag]["exclusive"] = {"avg_launch_ms": round(x_ms, 4), "achieved": round(x_gbs, 1),
                                             "frac": round(x_gbs / 8000.0, 4)}
        shares = {}
        for tag in ALL_TAGS:
            n_t, ms_t = prof[tag]
            if n_t:
                shares[tag] = {"launches_per_step": n_t / args.steps, "ms_per_step": round(ms_t / args.steps, 3)}
                if tag in concurrent:
                    shares[tag]["concurrent"] = True
        metric = ("edges/sec through one CGAT attention layer (fwd+bwd), 1M-edge batch" if args.workload == "layer"
                  else "batch-edges/sec through the full CGAT stack (4 layers, fwd+bwd), 1M-edge batch [informational]")
        out = {
            "metric": metric,
            "value": world * E * args.steps / elapsed, "unit": "edges/s", "n_gpus": world, "steps": args.steps,
            "warmup": args.warmup, "ms_per_step": ms, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f32", "data": "synthetic", "bilinear_mode": P.get_bilinear_mode(),
            "config": {"workload": (f"one GATConvNodes layer (H_Net update) fwd+bwd, {args.graphs} crystals x {ATOMS} atoms x "
                                    f"{K_NBR} nbrs per rank: N={N}, E={E}, C=Ce={C_FEA}, H={HEADS}, scalar attention")
                       if args.workload == "layer" else
                       (f"CGAtNet(200,128,4,msg_heads=3,update_edges=True) fwd+bwd of L1 loss, {args.graphs} crystals: "
                        f"N={N}, E={E}"),
                       "edges_per_rank": E, "parallelism": f"dp{world} (graphs sharded, gradient all-reduce)"},
            "roofline": roof, "hbm_bound_kernels": hbm if args.workload == "layer" else None,
            "kernel_ms_per_step": shares,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "layer":
            out["cpu_baseline"] = cpu_baseline()
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
