"""Device-side batch collation through the C ABI: bit-exact against the vectors recorded from the
unmodified reference and against the oracle; plus the host-side packing rules on CPU."""
import os
import sys

import numpy as np
import pytest
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.join(HERE, "golden"))
import collate_recipe as R

GOLD = np.load(os.path.join(HERE, "golden", "collate.npz"))
EMB = {el: R.embedding_table()[k].astype(np.float64).tolist() for k, el in enumerate(R.ELEMENTS)}
KEYS = ("x", "edge_index", "edge_attr", "y", "batch", "comp0", "comp1", "comp2", "comp3", "comp4")


def _tensors(gb, roost):
    return dict(zip(KEYS, (gb.x, gb.edge_index, gb.edge_attr, gb.y, gb.batch) + tuple(roost)))


@pytest.mark.gpu
@pytest.mark.parametrize("case", R.CASES, ids=[c[0] for c in R.CASES])
def test_collate_matches_reference_bit_exact(case):
    import cgat_amd as P
    name, n_graphs, seed, fmt, comps_as, max_nbr, target, batches = case
    data = R.make_dataset(n_graphs, seed, fmt, comps_as)
    ds = P.PackedDataset.from_dict(data, EMB, max_neighbor_number=max_nbr, target=target, device="cuda:0")
    for bi, ids in enumerate(batches):
        gb, roost = ds.collate(ids)
        torch.cuda.synchronize()
        for k, t in _tensors(gb, roost).items():
            ref = GOLD[f"{name}.b{bi}.{k}"]
            got = t.cpu().numpy()
            assert got.shape == ref.shape and got.dtype == ref.dtype, (k, got.shape, got.dtype, ref.shape, ref.dtype)
            assert np.array_equal(got, ref), k
        assert gb.num_graphs == len(ids) and gb.num_nodes == gb.x.shape[0]


@pytest.mark.gpu
def test_collate_large_batch_vs_oracle_and_feeds_the_model():
    """4167 ragged crystals in one batch (the BASELINE batch size in crystals) against the oracle, an empty batch,
    and the collated tensors driven through the HIP layer stack."""
    import cgat_amd as P
    from oracle import collate_oracle as O
    data = R.make_dataset(500, 9, 0, "list")
    ds = P.PackedDataset.from_dict(data, EMB, max_neighbor_number=12, device="cuda:0")
    rs = np.random.RandomState(0)
    ids = rs.randint(0, 500, size=4167)
    gb, roost = ds.collate(ids)
    want = O.collate(data, ids.tolist(), R.embedding_table(), {el: k for k, el in enumerate(R.ELEMENTS)}, 12, "e_above_hull")
    for k, t in _tensors(gb, roost).items():
        assert np.array_equal(t.cpu().numpy(), want[k]), k
    e_gb, e_roost = ds.collate([])
    assert e_gb.x.shape == (0, 200) and e_gb.edge_index.shape == (2, 0) and e_roost[0].shape == (0, 1)
    torch.manual_seed(0)
    net = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to("cuda:0")
    small, sroost = ds.collate(list(range(40)))
    out = net(small, sroost)
    assert out.shape == (40, 2) and torch.isfinite(out).all()


def test_packed_dataset_host_rules_cpu():
    """Pack-time parsing (data.py:62-80) without a GPU: three `comps` encodings give the same packed arrays as the
    oracle's per-crystal view; collating on the CPU is refused (no fallback)."""
    import cgat_amd as P
    from oracle import collate_oracle as O
    elem_id = {el: k for k, el in enumerate(R.ELEMENTS)}
    for comps_as, fmt in (("list", 0), ("tuple", 1), ("str", 0)):
        data = R.make_dataset(12, 5, fmt, comps_as)
        ds = P.PackedDataset.from_dict(data, EMB, max_neighbor_number=7, target="volume", device="cpu")
        assert len(ds) == 12
        for g in range(12):
            (x, ei, ea, y), (w, f, s, nb) = O.get_item(data, fmt, g, R.embedding_table(), elem_id, 7, "volume")
            a0, a1 = int(ds.t["atom_ptr"][g]), int(ds.t["atom_ptr"][g + 1])
            u0, u1 = int(ds.t["comp_ptr"][g]), int(ds.t["comp_ptr"][g + 1])
            assert a1 - a0 == x.shape[0] and u1 - u0 == f.shape[0]
            assert np.array_equal(ds.t["table"][ds.t["atom_elem"][a0:a1].long()].numpy(), x)
            assert np.array_equal(ds.t["comp_weight"][u0:u1].numpy(), w)
            assert np.array_equal(ds.t["shell"][a0:a1].numpy().astype(np.int64).flatten(), ea)
            assert np.array_equal(ds.t["nbr_idx"][a0:a1].numpy().astype(np.int64).flatten(), ei[1])
            assert float(ds.t["y_val"][g]) == float(y[0])
        with pytest.raises(RuntimeError):
            ds.collate([0, 1])
    bad = R.make_dataset(3, 1, 0, "list")
    bad["comps"][1][0] = "Xx"
    with pytest.raises(AssertionError):
        P.PackedDataset.from_dict(bad, EMB, device="cpu")


@pytest.mark.gpu
def test_training_step_matches_plain_torch_composition():
    """cgat_amd.DataParallelTrainer (BASELINE configs[3] on one rank: device collation -> CGAtNet -> RobustL1 ->
    backward -> fused AdamW) against the same step composed by hand from torch pieces (plain-python robust loss,
    torch.optim.AdamW) on ragged DCGAT-shaped crystals: identical parameters after two steps to 1e-5, with two
    micro-batches accumulated per step in a second run."""
    import copy
    import cgat_amd as P
    from cgat_amd.graph import synthetic_dataset_dict
    data, emb = synthetic_dataset_dict(60, (2, 40), 24, seed=3)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device="cuda:0")
    torch.manual_seed(0)
    net = P.CGAtNet(200, 64, 2, msg_heads=2, neighbor_number=12, update_edges=True).to("cuda:0")
    ref = copy.deepcopy(net)
    norm = P.Normalizer(0.3, 2.0)
    tr = P.DataParallelTrainer(net, ds, lr=1e-3, weight_decay=1e-2, normalizer=norm)
    opt = torch.optim.AdamW(ref.parameters(), lr=1e-3, weight_decay=1e-2)
    rs = np.random.RandomState(1)
    for _ in range(2):
        ids = rs.permutation(60)[:24]
        loss, edges = tr.step(ids)
        gb, roost = ds.collate(ids)
        assert edges == gb.edge_index.shape[1]
        opt.zero_grad()
        o, s = ref(gb, roost).chunk(2, dim=1)
        t = norm.norm(gb.y.view(-1, 1))
        want = (np.sqrt(2.0) * (o - t).abs() * torch.exp(-s) + s).mean()
        want.backward()
        opt.step()
        assert abs(float(loss) - float(want.detach())) <= 1e-5 * max(1.0, abs(float(want.detach())))
    # AdamW divides by sqrt(v): where a gradient element is ~0 the two runs' 1e-6-level gradient differences decide the
    # direction of an lr-sized update, so the bound is a fraction (0.2) of the largest possible update (lr per step)
    # (MH_A.fc_out.bias of scalar attention has an exactly-zero gradient by softmax shift invariance: what arrives is
    # rounding noise whose SIGN decides a full lr-sized AdamW step -- not comparable between two summation orders)
    for (n, p), q in zip(net.named_parameters(), ref.parameters()):
        if n.endswith("MH_A.fc_out.bias"):
            continue
        d = float((p.detach() - q.detach()).abs().max())
        assert d <= 0.2 * 1e-3 * 2, (n, d)
    # gradient accumulation over two micro-batches == one step on their union (same mean loss: equal sizes)
    tr2 = P.DataParallelTrainer(copy.deepcopy(ref), ds, lr=1e-3, weight_decay=1e-2, normalizer=norm, accumulate_grad_batches=2)
    tr1 = P.DataParallelTrainer(copy.deepcopy(ref), ds, lr=1e-3, weight_decay=1e-2, normalizer=norm)
    ids = rs.permutation(60)[:24]
    tr2.step([ids[:12], ids[12:]])
    tr1.step(ids)
    for (n, p), q in zip(tr2.model.named_parameters(), tr1.model.parameters()):
        if n.endswith("MH_A.fc_out.bias"):
            continue
        d = float((p.detach() - q.detach()).abs().max())
        assert d <= 0.2 * 1e-3, (n, d)


@pytest.mark.gpu
def test_training_step_never_synchronises_the_host():
    """after the warm-up steps (plan building, workspace growth) one whole training step -- device
    collation, forward, backward, fused AdamW -- issues no synchronising call: torch's sync-debug mode with the warning
    turned into an error.  (Round 4 still had one: the optimiser's pointer table was uploaded from pageable memory.)"""
    import warnings
    import cgat_amd as P
    from cgat_amd.graph import synthetic_dataset_dict
    data, emb = synthetic_dataset_dict(48, (2, 40), 24, seed=5)
    ds = P.PackedDataset.from_dict(data, emb, max_neighbor_number=12, device="cuda:0")
    from cgat_amd import ops
    torch.manual_seed(0)
    net = P.CGAtNet(200, 128, 2, msg_heads=3, neighbor_number=12, update_edges=True).to("cuda:0")
    tr = P.DataParallelTrainer(net, ds, lr=1e-3, weight_decay=1e-2)
    ids = np.random.RandomState(2).permutation(48)[:24]
    ops.set_validate_indices(False)         # the collation kernel is the only producer of the indices (as bench.py runs it);
    try:                                    # validating them is one deliberate .item() per plan
        for _ in range(2):
            tr.step(ids)
        torch.cuda.synchronize()
        torch.cuda.set_sync_debug_mode("warn")
        try:
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                tr.step(ids)
        finally:
            torch.cuda.set_sync_debug_mode("default")
    finally:
        ops.set_validate_indices(True)
    torch.cuda.synchronize()
    syncs = [str(x.message) for x in w if "synchroniz" in str(x.message).lower() and "prototype" not in str(x.message).lower()]
    assert not syncs, syncs
