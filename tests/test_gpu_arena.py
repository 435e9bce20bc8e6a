"""Device-resident dataset: on-device batch assembly equals the host collate, and a few optimisation steps on the
GPU follow the CPU oracle's loss trajectory."""
import numpy as np
import pytest
import torch

from helpers import g1_batch, g1_graph

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _arena(g1, count, with_host=False):
    from blackwater.data.arena import GraphArena

    xs, eis = [], []
    for i in range(count):
        x, ei, _ = g1_graph(g1, i)
        n = x.shape[0]
        loops = np.arange(n)
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))  # AddSelfLoops, as the training path
    host = g1_batch(g1, range(count))
    arena = GraphArena.from_arrays(xs, eis, host["y"].numpy(), host["noisy"].numpy(), host["depth"].numpy(),
                                   host["observable"].numpy(), device=DEV)
    return (arena, host) if with_host else arena


def test_batch_assembly_matches_host_collate(g1):
    from blackwater.native import ops

    arena, all_host = _arena(g1, 120, with_host=True)
    sel = [5, 119, 0, 5, 64, 33]  # arbitrary order, one repeat
    b = arena.batch(sel)
    host = g1_batch(g1, sel)
    assert torch.equal(b.x.cpu(), host["x"])
    assert torch.equal(b.noisy_0.cpu(), host["noisy"]) and torch.equal(b.y.cpu(), host["y"])
    assert torch.equal(b.circuit_depth.cpu(), host["depth"])
    assert torch.equal(b.observable.cpu(), all_host["observable"][sel])
    ref = ops.csr_build(host["edge_index"].to(DEV), host["x"].shape[0])
    s = b.structure
    m = int(ref[0][-1].item())
    assert s.num_edges == m
    for got, want, k in ((s.in_ptr, ref[0], None), (s.in_src, ref[1], m), (s.out_ptr, ref[2], None),
                         (s.out_dst, ref[3], m), (s.loops, ref[4], None)):
        assert torch.equal(got[:k] if k else got[: want.numel()], want[:k] if k else want)
    n_b = host["x"].shape[0]
    assert torch.equal(s.in_ell[:n_b], ops.ell_from_csr(ref[0], ref[1], n_b))      # rebased from the arena's side tables
    assert torch.equal(s.out_ell[:n_b], ops.ell_from_csr(ref[2], ref[3], n_b))
    gcn, sage, cheb = ops.graph_norms(ref[0], ref[2], ref[4], host["x"].shape[0])
    assert torch.equal(s.gcn_dinv, gcn) and torch.equal(s.sage_rinv, sage) and torch.equal(s.cheb_dinv, cheb)
    assert s.graph_ptr.cpu().tolist() == np.concatenate([[0], np.cumsum(arena.node_counts[sel])]).tolist()


def test_training_steps_follow_the_oracle(g1):
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import Trainer
    from oracle.models import FamilyA

    arena = _arena(g1, 96)
    torch.manual_seed(3)
    model = ExpValCircuitGraphModelA(5, 22, 10)
    ref = FamilyA(5, 22, 10)
    ref.load_state_dict(model.state_dict())
    # dropout off on both sides so that the two trajectories are comparable step by step
    ref.p_gcn = ref.p_other = 0.0
    ref.obs_seq[1].p = 0.0
    model = model.to(DEV)
    trainer = Trainer(model, lr=1e-3)
    model.eval()
    model.train = lambda *a, **k: model  # keep eval() through Trainer.step: the product applies dropout in train mode
    opt = torch.optim.Adam(ref.parameters(), lr=1e-3)
    args = ("noisy", "observable", "depth", "x", "edge_index", "batch")
    for step in range(5):
        sel = list(range(step * 16, step * 16 + 32))
        batch = arena.batch(sel)
        loss = trainer.step(batch).item()
        host = g1_batch(g1, sel)
        host["observable"] = batch.observable.cpu()  # the arena holds the observables drawn for the full list
        opt.zero_grad()
        ref_loss = torch.nn.functional.mse_loss(ref(*[host[k] for k in args]), host["y"])
        ref_loss.backward()
        opt.step()
        assert abs(loss - ref_loss.item()) < 2e-4 * max(1.0, abs(ref_loss.item())), (step, loss, ref_loss.item())
    for (name, p), (_, q) in zip(model.named_parameters(), ref.named_parameters()):
        assert torch.allclose(p.detach().cpu(), q.detach(), rtol=2e-3, atol=2e-4), name


def test_fit_epoch_loop_and_scheduler(g1):
    """Trainer.fit: the reference's epoch loop (docs/tutorials/__ml_models.py:145-187): per-epoch shuffling, summed
    validation loss into ReduceLROnPlateau, history without epoch 0; the loss goes down on the golden graphs."""
    from blackwater.nn import ExpValCircuitGraphModelA
    from blackwater.train import Trainer

    arena = _arena(g1, 160)
    torch.manual_seed(0)
    model = ExpValCircuitGraphModelA(5, 22, 10).to(DEV)
    trainer = Trainer(model, lr=1e-3)
    seen = []
    hist = trainer.fit(arena, train_ids=np.arange(0, 128), val_ids=np.arange(128, 160), epochs=6, batch_size=32,
                       log=lambda epoch, h: seen.append(epoch))
    assert seen == list(range(6))
    assert len(hist["train_losses"]) == 5 and len(hist["val_losses"]) == 5  # epoch 0 is dropped, as in the reference
    assert hist["train_losses"][-1] < hist["train_losses"][0]
    assert trainer.optimizer.param_groups[0]["lr"] == 1e-3  # no plateau yet (patience 15)
    sd = model.state_dict()  # parameters are views of the flat buffer but save/load like any module
    clone = ExpValCircuitGraphModelA(5, 22, 10)
    clone.load_state_dict({k: v.cpu() for k, v in sd.items()}, strict=True)


def test_arena_from_shards_equals_arena_from_arrays(g1, tmp_path):
    """Binary shards (data/shards.py) -> device arena: same batches as the per-graph-list path, for one rank and for
    the round-robin split over two ranks."""
    from blackwater.data.arena import GraphArena
    from blackwater.data.shards import pack_graphs, write_shard

    count = 90
    xs, eis = [], []
    for i in range(count):
        x, ei, _ = g1_graph(g1, i)
        loops = np.arange(x.shape[0])
        xs.append(x.astype(np.float32))
        eis.append(np.concatenate([ei, np.stack([loops, loops])], axis=1))
    host = g1_batch(g1, range(count))
    lab = [host[k].numpy() for k in ("y", "noisy", "depth", "observable")]
    want = GraphArena.from_arrays(xs, eis, *lab, device=DEV)
    paths = []
    for k, (lo, hi) in enumerate(((0, 50), (50, 90))):   # two shards of different sizes
        paths.append(str(tmp_path / f"part{k}.mlqs"))
        write_shard(paths[-1], pack_graphs(xs[lo:hi], eis[lo:hi], *[a[lo:hi] for a in lab]))
    got = GraphArena.from_shards(paths, device=DEV)
    assert np.array_equal(got.node_counts, want.node_counts) and np.array_equal(got.edge_counts, want.edge_counts)
    for name in ("x", "gptr", "in_ptr", "in_src", "out_ptr", "out_dst", "loops", "out_eid", "nscal", "y", "noisy",
                 "depth", "observable"):
        assert torch.equal(getattr(got, name), getattr(want, name)), name
    sel = [3, 77, 51, 49, 50]
    a, b = got.batch(sel), want.batch(sel)
    assert torch.equal(a.x, b.x) and torch.equal(a.structure.in_src, b.structure.in_src)
    # rank r of 2 keeps graphs r, r+2, ...
    for rank in (0, 1):
        part = GraphArena.from_shards(paths, device=DEV, rank=rank, world=2)
        ids = np.arange(rank, count, 2)
        assert np.array_equal(part.node_counts, want.node_counts[ids])
        pb, wb = part.batch(np.arange(len(ids))), want.batch(ids)
        assert torch.equal(pb.x, wb.x) and torch.equal(pb.y, wb.y)
        assert torch.equal(pb.structure.in_ptr, wb.structure.in_ptr) and torch.equal(pb.structure.in_src, wb.structure.in_src)
        assert torch.equal(pb.structure.out_dst, wb.structure.out_dst)


def test_end_to_end_example_runs(tmp_path):
    """examples/train_family_a.py: shards -> arena -> Trainer.fit -> mitigation_report, as a user would run it."""
    import importlib.util
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    spec = importlib.util.spec_from_file_location("example_train_family_a", os.path.join(root, "examples", "train_family_a.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ["train_family_a.py", "--qubits", "6", "--steps", "3", "--n-j", "12", "--epochs", "3", "--batch", "8",
                "--shard-dir", str(tmp_path)]
    try:
        hist, rep = mod.main()
    finally:
        sys.argv = argv
    assert len(hist["train_losses"]) == 2 and np.isfinite(hist["val_losses"]).all()
    assert set(("RMSE_noisy", "RMSE_mitigated", "MAE_noisy", "MAE_mitigated")) <= set(rep)
