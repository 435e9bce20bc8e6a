"""The small-batch path (VERDICT r01 missing #5): ``BucketedTrainer`` captures the whole Family A train step per size
bucket in a hipGraph and replays it; the same bucketed step run eagerly must give the SAME loss trajectory bit for bit,
and a bucket-padded batch must give the same predictions / gradients as the unpadded batch up to fp32 summation order."""
import numpy as np
import pytest
import torch

from helpers import g1_batch, g1_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _arena(g1, n=160, filler=2048, family_b=False):
    from blackwater.data.arena import GraphArena

    xs, eis = [], []
    for i in range(n):
        x, ei, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    host = g1_batch(g1, range(n))
    y, noisy = host["y"].numpy(), host["noisy"].numpy()
    if family_b:       # the reference's Family B takes exp values as [B, 1, k] (gnn.py:116)
        y, noisy = y.reshape(n, 1, -1), noisy.reshape(n, 1, -1)
    return GraphArena.from_arrays(xs, eis, y, noisy, host["depth"].numpy(), host["observable"].numpy(), device=DEV,
                                  filler_nodes=filler)


def test_padded_batch_equals_plain_batch(g1):
    from blackwater.nn import ExpValCircuitGraphModelA

    arena = _arena(g1)
    assert len(arena) == 160 and arena.filler_nodes == 2048
    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV).eval()
    ids = np.arange(10, 42)
    plain = arena.batch(ids)
    n, e = plain.structure.num_nodes, plain.structure.num_edges
    padded = arena.batch(ids, bucket=(-(-n // 256) * 256 + 256, e + 1000))
    assert padded.num_real == 32 and padded.num_graphs == 33 and padded.structure.num_nodes % 256 == 0
    # structure of the real part is identical, the filler rows are isolated
    assert torch.equal(padded.structure.in_ptr[:n + 1], plain.structure.in_ptr[:n + 1])
    assert torch.equal(padded.structure.in_src[:e], plain.structure.in_src[:e])
    assert (padded.structure.in_ptr[n:] == e).all() and (padded.structure.out_ptr[n:] == e).all()
    outs = []
    for b in (plain, padded):
        model.zero_grad()
        out = model(*b.model_args())[:32]
        torch.nn.functional.mse_loss(out, b.y[:32]).backward()
        outs.append((out.detach().clone(), [p.grad.clone() for p in model.parameters()]))
    assert (outs[0][0] - outs[1][0]).abs().max().item() < 1e-6
    for a, c in zip(outs[0][1], outs[1][1]):
        assert (a - c).abs().max().item() <= 1e-5 * (a.abs().max().item() + 1e-9)


def test_graph_replay_equals_eager_bucketed_steps_bit_for_bit(g1):
    """Same seeds, same selections: BucketedTrainer(graphs=True) and (graphs=False) -- dropout ON, Adam, 30 steps over
    several buckets (some revisited) -- must produce identical losses and identical final parameters."""
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import BucketedTrainer

    arena = _arena(g1)
    rng = np.random.RandomState(3)
    plans = [rng.choice(160, size=32, replace=False) for _ in range(30)]
    finals = []
    for graphs in (True, False):
        torch.manual_seed(0)
        model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV)
        tr = BucketedTrainer(model, arena, lr=1e-3, graphs=graphs, node_quantum=256, edge_quantum=512)
        torch.manual_seed(77)                    # torch's generator drives the observable MLP's dropout
        losses = [tr.step_ids(ids).item() for ids in plans]
        finals.append((losses, tr.flat_param.detach().clone(), len(tr._entries)))
    assert finals[0][2] >= 2                     # more than one bucket was captured
    assert finals[0][0] == finals[1][0]
    assert torch.equal(finals[0][1], finals[1][1])
    assert finals[0][0][-1] < finals[0][0][0]    # and it trains


def test_two_graph_form_equals_the_single_graph_bit_for_bit(g1):
    """Under data parallelism the step is captured as two graphs -- (assembly, forward, backward) and (Adam) -- with the
    gradient all-reduce enqueued between them.  ``split_update=True`` forces that form without a process group: same
    losses, same final parameters as the single captured graph."""
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import BucketedTrainer

    arena = _arena(g1)
    rng = np.random.RandomState(5)
    plans = [rng.choice(160, size=32, replace=False) for _ in range(20)]
    finals = []
    for split in (False, True):
        torch.manual_seed(0)
        model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV)
        tr = BucketedTrainer(model, arena, lr=1e-3, graphs=True, node_quantum=256, edge_quantum=512, split_update=split)
        torch.manual_seed(77)
        losses = [tr.step_ids(ids).item() for ids in plans]
        finals.append((losses, tr.flat_param.detach().clone()))
    assert finals[0][0] == finals[1][0]
    assert torch.equal(finals[0][1], finals[1][1])


def test_size_stratified_batches_share_one_bucket(g1):
    """StratifiedBatches: every batch holds the same number of graphs of every size, so all of them land in one bucket of
    the BucketedTrainer (one capture) and an epoch still visits every graph of a class before repeating one."""
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import BucketedTrainer, StratifiedBatches

    arena = _arena(g1)
    n = len(arena)
    sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], 32, seed=4)
    assert int(sampler.quota.sum()) == 32
    torch.manual_seed(0)
    tr = BucketedTrainer(ExpValCircuitGraphModelA(5, 22, 10).to(DEV), arena, lr=1e-3, graphs=True, node_quantum=256,
                         edge_quantum=512)
    buckets, first, last = set(), None, None
    for k in range(12):
        ids = sampler.draw()
        assert len(ids) == 32 and int(arena.node_counts[ids].sum()) == sampler.nodes_per_batch
        buckets.add(tr.bucket_of(ids))
        last = tr.step_ids(ids).item()
        first = last if first is None else first
    assert len(buckets) == 1 and len(tr._entries) == 1
    assert np.isfinite(last)


def test_family_b_step_replayed_from_a_graph_equals_the_eager_step(g1):
    """The reference's own model (TransformerConv / ASAPooling x2) through the captured step: ASAPooling's output sizes follow
    the per-graph node counts, so a capture is keyed by the batch's size pattern, which size-stratified batches repeat.
    graphs=True and graphs=False: same losses bit for bit (attention dropout on: its mask follows the device-resident step
    counter), one capture, and the loss moves."""
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import BucketedTrainer, StratifiedBatches

    arena = _arena(g1, family_b=True)
    n = len(arena)
    k = int(arena.y.shape[-1])
    finals = []
    for graphs in (True, False):
        sampler = StratifiedBatches(arena.node_counts[:n], arena.edge_counts[:n], 32, seed=9)
        torch.manual_seed(0)
        model = ExpValCircuitGraphModel(22, 15, k).to(DEV)
        tr = BucketedTrainer(model, arena, lr=1e-3, graphs=graphs, node_quantum=256, edge_quantum=512)
        torch.manual_seed(78)
        losses = [tr.step_ids(sampler.draw()).item() for _ in range(14)]
        finals.append((losses, tr.flat_param.detach().clone(), len(tr._entries)))
    assert finals[0][2] == 1
    assert finals[0][0] == finals[1][0]
    assert torch.equal(finals[0][1], finals[1][1])
    assert len(set(finals[0][0])) > 10           # fresh batches and fresh dropout masks every replay


def test_bucketed_trainer_follows_the_plain_trainer(g1):
    """Against the ordinary Trainer (no padding, host-counted dropout keys): dropout off so that the masks cannot differ,
    losses agree to fp32 rounding over 20 steps."""
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.native import ops
    from blackwater.train import BucketedTrainer, Trainer

    arena = _arena(g1)
    rng = np.random.RandomState(5)
    plans = [rng.choice(160, size=32, replace=False) for _ in range(20)]

    def no_dropout(model):
        model.eval()
        model.train = lambda *a, **k: model
        return model

    torch.manual_seed(0)
    m1 = no_dropout(ExpValCircuitGraphModelA(5, 22, 10).to(DEV))
    t1 = BucketedTrainer(m1, arena, graphs=True, node_quantum=256, edge_quantum=512)
    l1 = [t1.step_ids(ids).item() for ids in plans]
    ops.set_seed_counter(None)
    torch.manual_seed(0)
    m2 = no_dropout(ExpValCircuitGraphModelA(5, 22, 10).to(DEV))
    t2 = Trainer(m2)
    l2 = [t2.step(arena.batch(ids)).item() for ids in plans]
    assert np.allclose(l1, l2, rtol=2e-5, atol=1e-7)


def _cfg2_arena(n_j=12, filler=2048):
    """4-qubit TFIM-Trotter circuits of 15 different sizes (Trotter steps 0-14): the reference's training set shape (cfg2)."""
    from blackwater.data.arena import GraphArena
    from blackwater.data.synthetic import TfimCorpus

    h = TfimCorpus(4, list(range(15)), n_j, seed=3, two_q="cx", exp_value_size=4).host_graphs()
    return GraphArena.from_arrays(h["x"], h["edge_index"], h["y"][:, None, :], h["noisy"][:, None, :], h["depth"], h["observable"],
                                  device=DEV, filler_nodes=filler)


def test_size_stable_batch_pools_to_the_buckets_totals_and_leaves_the_circuits_alone():
    """A size-stable batch (train.stable_padding): the fillers bring the node totals after BOTH poolings to the bucket's values, the
    pooled boundaries computed on the device equal the host formula, and the circuits' predictions equal the unpadded batch's."""
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import BucketedTrainer

    arena = _cfg2_arena()
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel(22, 15, 4).to(DEV)
    tr = BucketedTrainer(model, arena, graphs=False, node_quantum=1024, edge_quantum=4096)
    assert tr.stable
    rng = np.random.RandomState(1)
    model.eval()
    for _ in range(3):
        ids = rng.permutation(len(arena))[:32]
        bucket, fill, plan, cap = tr._stable_bucket(ids)
        assert len(fill) == 16 and bucket[2] == 48 and bucket[0] % 1024 == 0
        sel, nptr, eptr, nb, eb, real = arena.selection(ids, bucket[:2], filler_sizes=fill)
        packed = torch.from_numpy(np.concatenate([sel, nptr, eptr]).astype(np.int32)).to(DEV)
        batch = arena.assemble(packed, len(sel), nb, eb, None, None, real, coarse_capacity=cap, pool_plan=plan)
        s0 = batch.structure
        sizes = nptr[1:] - nptr[:-1]
        keep1 = np.ceil(sizes.astype(np.float32) * np.float32(0.5)).astype(np.int64)
        keep2 = np.ceil(keep1.astype(np.float32) * np.float32(0.5)).astype(np.int64)
        assert plan[0][0] == keep1.sum() and plan[1][0] == keep2.sum()
        assert sizes.max() <= plan[0][1] and keep1.max() <= plan[0][2] and keep2.max() <= plan[1][2]
        got_ptr = ops.pool_keep_ptr(s0.graph_ptr, s0.num_graphs, 0.5).cpu().numpy()
        assert np.array_equal(got_ptr, np.concatenate([[0], np.cumsum(keep1)]))
        with torch.no_grad():
            g = model.transformer1(batch.nodes, s0)
            g, s1, _ = model.pooling1(g, s0)
            assert s1.num_nodes == plan[0][0] and int(s1.graph_ptr[-1].item()) == plan[0][0]
            g = model.transformer2(g, s1)
            g, s2, _ = model.pooling2(g, s1)
            assert s2.num_nodes == plan[1][0] and int(s2.graph_ptr[-1].item()) == plan[1][0]
            out_padded = model(*batch.model_args())[:32]
            out_plain = model(*arena.batch(ids).model_args())
        assert (out_padded - out_plain).abs().max().item() < 1e-6
    ops.set_seed_counter(None)


def test_family_b_shuffled_batches_replay_from_size_stable_captures():
    """The reference's loader (DataLoader(batch_size=32, shuffle=True), docs/tutorials/__ml_models.py:105): uniformly shuffled
    batches never repeat a size sequence, but they do repeat size-stable BUCKETS.  Captured (graphs=True) and eager (graphs=False)
    bucketed steps -- attention and head dropout ON, Adam, 30 steps -- give the same losses and parameters bit for bit; the captured
    run replays (fewer captures than steps), stays inside its capture budget and never runs a step eagerly."""
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import BucketedTrainer

    arena = _cfg2_arena()
    n = len(arena)
    finals = []
    for graphs in (True, False):
        rng = np.random.RandomState(11)
        torch.manual_seed(0)
        model = ExpValCircuitGraphModel(22, 15, 4).to(DEV)
        tr = BucketedTrainer(model, arena, lr=1e-3, graphs=graphs, node_quantum=1024, edge_quantum=4096)
        assert tr.stable
        torch.manual_seed(5)
        losses, keys = [], []
        for epoch in range(6):
            order = rng.permutation(n)
            for i in range(0, 32 * 5, 32):
                ids = order[i:i + 32]
                keys.append(tr.bucket_of(ids))
                losses.append(tr.step_ids(ids).item())
        finals.append((losses, tr.flat_param.detach().clone(), len(tr._entries), len(set(keys))))
        ops.check_overflow_flags()
        ops.set_seed_counter(None)
    assert len(finals[0][0]) == 30 and all(np.isfinite(finals[0][0]))
    assert finals[0][0] == finals[1][0]
    assert torch.equal(finals[0][1], finals[1][1])
    captures, buckets = finals[0][2], finals[0][3]
    assert captures == buckets < 30          # every bucket captured at first sight, and buckets come back
    assert captures < int(__import__("os").environ.get("MLQEM_MAX_PATTERN_CAPTURES", "64"))
    assert len(set(finals[0][0])) > 25


def test_bn_head_does_not_see_filler_rows():
    """ExpValCircuitGraphModel_2 (MLP2 head: BatchNorm over the batch): a bucket-padded batch must give the circuits the outputs of
    the unpadded batch in TRAIN mode too -- the filler graphs' rows are cut off before the head (dropout off for the comparison)."""
    from blackwater.native import ops
    from blackwater.nn.family_b import ExpValCircuitGraphModel_2
    from blackwater.train import BucketedTrainer

    arena = _cfg2_arena()
    torch.manual_seed(0)
    model = ExpValCircuitGraphModel_2(22, 15, 4, dropout=0.0).to(DEV).train()
    model.transformer1.dropout = model.transformer2.dropout = 0.0
    tr = BucketedTrainer(model, arena, graphs=False, node_quantum=1024, edge_quantum=4096)
    ids = np.arange(0, 160, 5)
    bucket, fill, plan, cap = tr._stable_bucket(ids)
    sel, nptr, eptr, nb, eb, real = arena.selection(ids, bucket[:2], filler_sizes=fill)
    packed = torch.from_numpy(np.concatenate([sel, nptr, eptr]).astype(np.int32)).to(DEV)
    batch = arena.assemble(packed, len(sel), nb, eb, None, None, real, coarse_capacity=cap, pool_plan=plan)
    with torch.no_grad():
        out_padded = model(*batch.model_args())
        out_plain = model(*arena.batch(ids).model_args())
    assert out_padded.shape == out_plain.shape == (32, 4)
    assert (out_padded - out_plain).abs().max().item() < 1e-5
    ops.set_seed_counter(None)


def test_fit_with_the_references_loader_replays_captured_steps():
    """``BucketedTrainer.fit`` -- the reference's loop (shuffled epochs of 32, summed validation loss into ReduceLROnPlateau,
    docs/tutorials/__ml_models.py:100-187) -- on Family B: every training step is the replay of a size-stable bucket's capture; the
    loss curves equal the eagerly enqueued bucketed fit's bit for bit (dropout on), and a scheduler step that lowers the rate
    reaches the captured Adam (the rate is a device tensor the graphs read)."""
    from blackwater.native import ops
    from blackwater.nn import ExpValCircuitGraphModel
    from blackwater.train import BucketedTrainer

    arena = _cfg2_arena(n_j=14)
    n = len(arena)
    train_ids, val_ids = np.arange(0, n - 30), np.arange(n - 30, n)
    curves = []
    for graphs in (True, False):
        torch.manual_seed(0)
        model = ExpValCircuitGraphModel(22, 15, 4).to(DEV)
        tr = BucketedTrainer(model, arena, lr=1e-3, graphs=graphs, node_quantum=1024, edge_quantum=4096)
        torch.manual_seed(7)
        hist = tr.fit(arena, train_ids, val_ids, epochs=4, batch_size=32, seed=3)
        curves.append((hist["train_losses"], hist["val_losses"], tr.flat_param.detach().clone(), len(tr._entries)))
        if graphs:
            steps = 4 * (-(-len(train_ids) // 32))
            assert 0 < len(tr._entries) < steps            # buckets come back: replays, not one capture per step
            lr = tr.optimizer.param_groups[0]["lr"]
            before = tr.flat_param.detach().clone()
            lr.fill_(0.0)                                   # what ReduceLROnPlateau does, in place -- taken to the extreme
            tr.step_ids(train_ids[:32])
            assert torch.equal(tr.flat_param.detach(), before)      # the replayed Adam read the new rate: no movement
            lr.fill_(1e-3)
        ops.set_seed_counter(None)
    assert len(curves[0][0]) == 3 and all(np.isfinite(curves[0][0])) and all(np.isfinite(curves[0][1]))
    assert curves[0][0] == curves[1][0] and curves[0][1] == curves[1][1]
    assert curves[0][0][-1] < curves[0][0][0]
